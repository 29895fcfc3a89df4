#!/bin/bash
# round 6, call 15: slice length with the new rings (fewer parks = fewer L2 write-backs / invalidates against a longer tail), and the bench lines
ROOT=$(pwd); OUT=$ROOT/gpurun_out/c15; mkdir -p $OUT
L=uncalled_amd/libuncalled_hip.so
AB_NOPROF=1 timeout 600 python tools/dev/ab_libs.py 50000 $L $L@0@2048 $L@0@4096 $L $L@0@2048 $L@0@512 > $OUT/ab_slice_ecoli.log 2>&1; grep "k_map ms" $OUT/ab_slice_ecoli.log | cut -c1-200
AB_NOPROF=1 timeout 1200 python tools/dev/ab_libs.py 250000:grch38 $L $L@0@2048 $L@0@4096 > $OUT/ab_slice_grch38.log 2>&1; grep "k_map ms" $OUT/ab_slice_grch38.log | cut -c1-200
timeout 900 python bench.py --steps 8 --warmup 2 --secondary "" > $OUT/bench_headline.json 2> $OUT/bench_headline.err; head -c 900 $OUT/bench_headline.json; echo
