#!/bin/bash
# round 6, call 13: what add_seeds does in the reads that the slow XCDs decide (rounds, nodes walked, cycles inside the pool's ring): dev build -DUNC_DBG_SEED=1
ROOT=$(pwd); OUT=$ROOT/gpurun_out/c13; mkdir -p $OUT
export SPREAD_LIB=$ROOT/uncalled_amd/variants/libunc_dbgseed.so
SPREAD_WORKLOAD=ecoli timeout 600 python tools/dev/grch38_phase_spread.py 50000 1 auto > $OUT/dbgseed_ecoli.log 2>&1; grep -v "build_index" $OUT/dbgseed_ecoli.log | cut -c1-1500 | tail -8
timeout 1500 python tools/dev/grch38_phase_spread.py 250000 3 auto > $OUT/dbgseed_grch38.log 2>&1; grep -v "build_index" $OUT/dbgseed_grch38.log | cut -c1-2500 | tail -40
