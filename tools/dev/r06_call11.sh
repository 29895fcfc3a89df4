#!/bin/bash
# round 6, call 11: GRCh38 slow / fast launches per read and per XCD; the pool at 1.27 x its high-water; longer slices
ROOT=$(pwd); OUT=$ROOT/gpurun_out/c11; mkdir -p $OUT
timeout 1700 python tools/dev/grch38_phase_spread.py 250000 6 auto,pool=460000,slice=4096 > $OUT/grch38_phase_spread_variants.log 2>&1; grep -v "build_index_big" $OUT/grch38_phase_spread_variants.log | cut -c1-420 | tail -60
