#!/bin/bash
# round 6, call 10: the torch-free index build (tests + chr20-sized timing), then the chr20 bench with the 24-chunks-per-slot pool rule
ROOT=$(pwd); OUT=$ROOT/gpurun_out/c10; mkdir -p $OUT
timeout 1200 python -m pytest tests/test_gpu_index_build.py tests/test_gpu_host.py -m gpu -x -q > $OUT/pytest_index_build.log 2>&1; echo "pytest exit $?"; tail -5 $OUT/pytest_index_build.log
timeout 600 python tools/dev/time_index_build.py > $OUT/time_index_build_chr20.log 2>&1; cat $OUT/time_index_build_chr20.log | tail -6
timeout 900 python bench.py --workload chr20 --steps 4 --warmup 1 > $OUT/bench_chr20.json 2> $OUT/bench_chr20.err; tail -3 $OUT/bench_chr20.err; head -c 1500 $OUT/bench_chr20.json
