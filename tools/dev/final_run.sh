#!/bin/bash
# Dev tool (GPU box): the measurements that go into profiles/ (round 6).  Usage: tools/dev/final_run.sh [stage ...]
#   tests    pytest -m gpu (everything, grch38 included)
#   smoke    __graft_entry__.smoke()
#   pmc / pmc_chr20 / pmc_grch38   counters of the SHIPPED k_map on the bench's own batch of that workload, one rocprofv3 pass per group:
#            SQ wave-cycle shares (a), instruction mix (b), SQ_INSTS (d), FETCH_SIZE / WRITE_SIZE (f, w) and their known-byte
#            calibration kernels (cf, cw) -> gpurun_out/final/pmc_<workload>/{summary,pmc_k_map}.json, copied to profiles/r06_*_<workload>.json so that
#            the bench stage below reads `traffic` and `issue` of THIS kernel
#   bench    the driver's command line (headline + cpu_baseline + secondary blocks)
#   stats    the E. coli headline under rocprofv3 --kernel-trace --stats (steps one after the other: --no-pipeline, so that a dispatch's duration is one launch's)
#   e2e      python -m uncalled_amd map on multi-fast5 files (end to end: HDF5 -> staging -> GPU -> PAF text)
ROOT=$(pwd); OUT=$ROOT/gpurun_out/final; mkdir -p $OUT
STAGES=${@:-tests smoke pmc bench stats}   # pmc before bench: the bench line reads `traffic` / `issue` from the summaries the pmc stage writes
for s in $STAGES; do
case $s in
tests)
  (timeout 1200 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_gpu.log); tail -3 $OUT/pytest_gpu.log ;;
smoke)
  timeout 300 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log ;;
pmc|pmc_chr20|pmc_grch38)
  # counters of the SHIPPED k_map on the bench's own batch of that workload -> profiles/r06_pmc_{k_map,sq_summary}_<workload>.json,
  # which the bench stage below reads `traffic` and `issue` from
  case $s in pmc) W=ecoli; R=50000;; pmc_chr20) W=chr20; R=200000;; pmc_grch38) W=grch38; R=250000;; esac
  bash tools/dev/pmc_sq.sh gpurun_out/final/pmc_$W uncalled_amd/libuncalled_hip.so $R:$W a b c d f w cf cw > $OUT/pmc_$W.log 2>&1; tail -5 $OUT/pmc_$W.log
  cp $OUT/pmc_$W/summary.json profiles/r06_pmc_sq_summary_$W.json; cp $OUT/pmc_$W/pmc_k_map.json profiles/r06_pmc_k_map_$W.json
  mkdir -p $OUT/profiles_out; cp profiles/r06_pmc_sq_summary_$W.json profiles/r06_pmc_k_map_$W.json $OUT/profiles_out/ ;;
bench)
  UNC_BENCH_DETAIL=$OUT/bench_detail.json timeout 1700 python bench.py --steps 20 --warmup 5 > $OUT/bench_default.json 2> $OUT/bench_default.err; tail -c 400 $OUT/bench_default.json; echo ;;
stats)
  cd /tmp; export TMPDIR=/tmp
  UNC_BENCH_DETAIL=$OUT/bench_detail_rocprof.json timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o r06 -- python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-profile-pass --no-pipeline --secondary "" > $OUT/bench_rocprof.json 2> $OUT/bench_rocprof.err
  cd $ROOT; ls $OUT/stats/*/ 2>/dev/null | head ;;
e2e)
  timeout 900 python tools/dev/e2e_map.py 200000 > $OUT/e2e_map.json 2> $OUT/e2e_map.err; tail -c 600 $OUT/e2e_map.json; echo ;;
esac
done
find $OUT -name "*_kernel_trace.csv" -size +3M -delete   # keep the merge-back small: the stats / counter files carry what is needed
find $ROOT/gpurun_out -type f -size +6M -delete          # gpurun merges back at most 64 MiB and drops EVERYTHING beyond that
du -sh $ROOT/gpurun_out; ls -la $OUT | head -40
