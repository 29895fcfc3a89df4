#!/bin/bash
# Dev tool (GPU box): hardware counters of k_map on the bench's own 50 k-read E. coli batch, one rocprofv3 pass per group
# (--pmc with --kernel-trace only).  Each pass maps the batch ONCE with the given library (tools/dev/ab_libs.py, one run);
# the passes cf / cw run the known-byte calibration kernels (tools/dev/pmc_calib.py) under FETCH_SIZE / WRITE_SIZE.
#   bash tools/dev/pmc_sq.sh <outdir> [lib.so] [reads] [groups...]
# Summaries (written BEFORE the large kernel traces are deleted, durations taken from each pass's own trace):
#   <outdir>/summary.json         SQ counters + derived shares           (-> profiles/rNN_pmc_sq_summary.json)
#   <outdir>/pmc_k_map.json       calibrated FETCH_SIZE + WRITE_SIZE     (-> profiles/rNN_pmc_k_map.json; bench.py reads it)
OUT=${1:-gpurun_out/pmc_sq}; LIB=${2:-uncalled_amd/libuncalled_hip.so}; READS=${3:-50000}; shift 3
# <reads> may carry the workload of tools/dev/ab_libs.py: 250000:grch38, 200000:chr20 (default ecoli)
NREADS=${READS%%:*}; WORKLOAD=ecoli; case $READS in *:*) WORKLOAD=${READS#*:};; esac
GROUPS_=${@:-a b d f w cf cw}
ROOT=$(pwd); mkdir -p $ROOT/$OUT; cd /tmp; export TMPDIR=/tmp
run() { name=$1; shift
  rm -rf $ROOT/$OUT/$name
  AB_NOPROF=1 AB_RUNS=1 timeout 600 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $ROOT/$OUT/$name -o pmc -- \
        python $ROOT/tools/dev/ab_libs.py $READS $ROOT/$LIB > $ROOT/$OUT/$name.log 2>&1 || echo "pass $name failed"; tail -1 $ROOT/$OUT/$name.log; }
calib() { name=$1; shift
  rm -rf $ROOT/$OUT/$name
  timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $ROOT/$OUT/$name -o calib -- \
        python $ROOT/tools/dev/pmc_calib.py > $ROOT/$OUT/$name.log 2>&1 || echo "pass $name failed"; tail -1 $ROOT/$OUT/$name.log; }
for g in $GROUPS_; do
case $g in
a) run a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_VMEM ;;   # (round 5: SQ_ACTIVE_INST_VMEM reads 0 on this build; the VMEM level is the witness of waiting on memory)
b) run b SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_INSTS_BRANCH ;;
c) run c SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VALU SQ_INST_CYCLES_SALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES ;;
d) run d SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS ;;
e) run e SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_SMEM SQ_IFETCH SQ_IFETCH_LEVEL SQ_LEVEL_WAVES SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE ;;
f) run f FETCH_SIZE ;;
w) run w WRITE_SIZE ;;
cf) calib cf FETCH_SIZE ;;
cw) calib cw WRITE_SIZE ;;
esac
done
cd $ROOT
python tools/dev/summarise_sq.py $OUT $NREADS $OUT/summary.json $WORKLOAD
if [ -d $OUT/f ] && [ -d $OUT/w ]; then python tools/dev/summarise_pmc.py $OUT $OUT/pmc_k_map.json $NREADS $WORKLOAD f w cf cw | tail -12; fi
# (after the summaries, which take counters and durations from these files:) what goes home is k_map's and the calibration kernels'
# rows only -- the complete CSVs of seven passes are more than gpurun merges back
for f in $(find $OUT -name "*_counter_collection.csv" -o -name "*_kernel_trace.csv"); do
  (head -1 $f; grep -E "k_map|k_calib" $f) > ${f%.csv}_k_map_rows.csv; rm -f $f
done
find $OUT -type f -size +4M -delete
