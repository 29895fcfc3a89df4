#!/bin/bash
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r05; mkdir -p $OUT
UNC_BENCH_DETAIL=$OUT/bench_detail_headline.json timeout 500 python bench.py --steps 20 --warmup 5 --secondary "" > $OUT/bench_headline.json 2> $OUT/bench_headline.err; tail -c 300 $OUT/bench_headline.json; echo
timeout 420 python tools/dev/check_sorter.py grch38 > $OUT/check_sorter_grch38.log 2>&1; tail -4 $OUT/check_sorter_grch38.log
