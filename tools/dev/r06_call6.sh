#!/bin/bash
# round 6, call 6: GRCh38 at the bench's size (250 000 reads): round 5's library against the round-6 tree, three launches each, twice
ROOT=$(pwd); OUT=$ROOT/gpurun_out/c6; mkdir -p $OUT
V=uncalled_amd/variants
AB_NOPROF=1 AB_RUNS=3 timeout 2400 python tools/dev/ab_libs.py 250000:grch38 $V/libunc_base.so uncalled_amd/libuncalled_hip.so $V/libunc_base.so uncalled_amd/libuncalled_hip.so > $OUT/ab_grch38_250k.log 2>&1; grep -v "^{" $OUT/ab_grch38_250k.log | tail -5
