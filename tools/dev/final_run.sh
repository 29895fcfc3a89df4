#!/bin/bash
# Dev tool (GPU box): the measurements that go into profiles/ (round 2).  Usage: tools/dev/final_run.sh [stage ...]
#   tests    pytest -m gpu (everything, grch38 included)
#   bench    the driver's command line (headline + cpu_baseline + secondary blocks)
#   stats    the E. coli headline under rocprofv3 --kernel-trace --stats
#   pmc      HBM traffic of k_map: FETCH_SIZE / WRITE_SIZE, one pass each, on the bench's own batch (50 k reads), plus the
#            known-byte calibration kernels in the same access shape
#   rt       realtime workload (512 channels), E. coli and chr20 thresholds
#   e2e      python -m uncalled_amd map on multi-fast5 files (end to end: HDF5 -> staging -> GPU -> PAF text)
ROOT=$(pwd); OUT=$ROOT/gpurun_out/final; mkdir -p $OUT
STAGES=${@:-tests pmc bench stats rt e2e}   # pmc before bench: the bench line reads `traffic` from the summary the pmc stage writes
for s in $STAGES; do
case $s in
tests)
  (timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_gpu.log); tail -3 $OUT/pytest_gpu.log ;;
bench)
  timeout 1500 python bench.py --steps 20 --warmup 5 > $OUT/bench_default.json 2> $OUT/bench_default.err; tail -c 400 $OUT/bench_default.json; echo ;;
stats)
  cd /tmp; export TMPDIR=/tmp
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o r02 -- python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --secondary "" > $OUT/bench_rocprof.json 2> $OUT/bench_rocprof.err
  cd $ROOT; ls $OUT/stats/*/ 2>/dev/null | head ;;
pmc)
  cd /tmp; export TMPDIR=/tmp
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 500 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_$c -o pmc -- python $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-profile-pass --secondary "" > $OUT/pmc_$c.json 2> $OUT/pmc_$c.err || echo "pmc $c failed"
    timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/calib_$c -o calib -- python $ROOT/tools/dev/pmc_calib.py > $OUT/calib_$c.log 2>&1 || echo "calib $c failed"
  done
  cd $ROOT; python tools/dev/summarise_pmc.py $OUT profiles/r02_pmc_k_map.json 50000 ecoli | tail -30 ;;
rt)
  timeout 300 python bench.py --workload realtime --steps 20 --warmup 3 > $OUT/bench_realtime.json 2> $OUT/bench_realtime.err; tail -c 600 $OUT/bench_realtime.json; echo
  timeout 400 python bench.py --workload realtime --rt-ref chr20 --steps 20 --warmup 3 > $OUT/bench_realtime_chr20.json 2> $OUT/bench_realtime_chr20.err; tail -c 600 $OUT/bench_realtime_chr20.json; echo ;;
e2e)
  timeout 600 python tools/dev/e2e_map.py 20000 > $OUT/e2e_map.json 2> $OUT/e2e_map.err; tail -c 600 $OUT/e2e_map.json; echo ;;
esac
done
find $OUT -name "*_kernel_trace.csv" -size +3M -delete   # keep the merge-back small: the stats / counter files carry what is needed
ls -la $OUT | head -40
