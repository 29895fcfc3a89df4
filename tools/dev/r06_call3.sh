#!/bin/bash
# round 6, call 3: the wide instantiation (GRCh38, 50 k reads) -- round 5's library against the round-6 tree (branch-free tile staging, run
# offsets out of scratch memory, field-wise merge, k-mer ranges a pass ahead in the walk), alternating, three launches each
ROOT=$(pwd); OUT=$ROOT/gpurun_out/c3; mkdir -p $OUT
V=uncalled_amd/variants
AB_NOPROF=1 AB_RUNS=3 timeout 1500 python tools/dev/ab_libs.py 50000:grch38 $V/libunc_base.so $V/libunc_wide1.so $V/libunc_base.so $V/libunc_wide1.so > $OUT/ab_grch38.log 2>&1; grep -v "^{" $OUT/ab_grch38.log | tail -5
