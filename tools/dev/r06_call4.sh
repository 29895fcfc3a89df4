#!/bin/bash
# round 6, call 4: the headline with batches overlapped (two mappers, unc_map_batch_begin / _end) and without; the chunked path before /
# after the walk's countable loads (teams of 8); GPU parity subset
ROOT=$(pwd); OUT=$ROOT/gpurun_out/c4; mkdir -p $OUT
UNC_BENCH_DETAIL=$OUT/detail_pipe.json timeout 600 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-profile-pass --secondary "" > $OUT/bench_pipe.json 2> $OUT/bench_pipe.err; tail -c 700 $OUT/bench_pipe.json; echo
UNC_BENCH_DETAIL=$OUT/detail_nopipe.json timeout 600 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-profile-pass --secondary "" --no-pipeline > $OUT/bench_nopipe.json 2> $OUT/bench_nopipe.err; tail -c 700 $OUT/bench_nopipe.json; echo
V=uncalled_amd/variants
timeout 600 python tools/dev/ab_rt.py 30 $V/libunc_base.so $V/libunc_head.so $V/libunc_base.so $V/libunc_head.so > $OUT/ab_rt.log 2>&1; tail -6 $OUT/ab_rt.log
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "two_halves or same_row or mid_reference or synthetic_batch or example_read or chunked or team" > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
