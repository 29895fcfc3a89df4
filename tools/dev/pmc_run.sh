#!/bin/bash
# Dev tool (GPU box): hardware counters of k_map, one rocprofv3 pass per group (no tracing domains besides kernel-trace).
#   bash tools/dev/pmc_run.sh <outdir> [reads]
OUT=${1:-gpurun_out/pmc}; READS=${2:-12000}
ROOT=$(pwd); mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
run() { name=$1; shift; timeout 400 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $ROOT/$OUT/$name -o pmc -- \
        python $ROOT/bench.py --reads $READS --steps 1 --warmup 0 --no-cpu-baseline --no-profile-pass --secondary "" > $ROOT/$OUT/$name.log 2>&1 || echo "pass $name failed"; }
run sq SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM SQ_IFETCH_LEVEL
run sq2 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_FLAT SQ_WAVES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU
# (round 1: the TA_* / TCP_* groups abort on this rocprofv3 build and sit until the timeout -- left out)
#run ta TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_TOTAL_WAVEFRONTS_sum GRBM_GUI_ACTIVE
#run tcp TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCP_TA_TCP_STATE_READ_sum TCP_PENDING_STALL_CYCLES_sum
run sqc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_DCACHE_REQ SQC_DCACHE_MISSES SQ_INSTS_SMEM SQ_IFETCH
#run tcp2 TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum
cd $ROOT
python - <<PY
import csv, glob, collections
tot = collections.defaultdict(float)
for f in glob.glob("$OUT/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_map" in r.get("Kernel_Name", ""):
            tot[r["Counter_Name"]] += float(r["Counter_Value"])
for k in sorted(tot): print("%-45s %.6g" % (k, tot[k]))
PY
