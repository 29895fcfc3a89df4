#!/bin/bash
# Dev tool (GPU box): the measurements that go into profiles/ -- gpu tests, default bench line, rocprofv3 kernel stats of the
# same command, HBM traffic counters (one pass each), realtime and chr20-sized workloads.
ROOT=$(pwd); OUT=$ROOT/gpurun_out/final; mkdir -p $OUT
(timeout 700 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_gpu.log)
tail -3 $OUT/pytest_gpu.log
timeout 500 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; tail -c 600 $OUT/bench_default.json
cd /tmp; export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o r01 -- python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_rocprof.json 2> $OUT/bench_rocprof.err
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_$c -o pmc -- python $ROOT/bench.py --reads 12000 --steps 1 --warmup 0 --no-cpu-baseline --no-profile-pass > $OUT/pmc_$c.json 2> $OUT/pmc_$c.err || echo "pmc $c failed"
done
cd $ROOT
timeout 300 python bench.py --workload realtime > $OUT/bench_realtime.json 2> $OUT/bench_realtime.err; tail -c 300 $OUT/bench_realtime.json
timeout 500 python bench.py --workload chr20 --steps 1 --warmup 1 --no-cpu-baseline > $OUT/bench_chr20.json 2> $OUT/bench_chr20.err; tail -c 300 $OUT/bench_chr20.json
find $OUT -name "*_kernel_trace.csv" -size +3M -delete   # keep the merge-back small: the stats files carry what is needed
ls -la $OUT $OUT/stats 2>/dev/null | head -40
