#!/bin/bash
# round 6, call 14: the rings without acquire loads in polling loops and the pool's ring without a compare-and-swap loop, against the tree before
ROOT=$(pwd); OUT=$ROOT/gpurun_out/c14; mkdir -p $OUT
V=$ROOT/uncalled_amd/variants
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "pool or scale or slic or big_forest or example" > $OUT/pytest_subset.log 2>&1; echo "pytest exit $?"; tail -3 $OUT/pytest_subset.log
AB_NOPROF=1 timeout 600 python tools/dev/ab_libs.py 50000 $V/libunc_oldring.so uncalled_amd/libuncalled_hip.so $V/libunc_oldring.so uncalled_amd/libuncalled_hip.so > $OUT/ab_rings_ecoli.log 2>&1; grep "k_map ms" $OUT/ab_rings_ecoli.log | cut -c1-200
AB_NOPROF=1 timeout 900 python tools/dev/ab_libs.py 200000:chr20 $V/libunc_oldring.so uncalled_amd/libuncalled_hip.so > $OUT/ab_rings_chr20.log 2>&1; grep "k_map ms" $OUT/ab_rings_chr20.log | cut -c1-200
AB_RUNS=3 AB_NOPROF=1 timeout 1500 python tools/dev/ab_libs.py 250000:grch38 $V/libunc_oldring.so uncalled_amd/libuncalled_hip.so > $OUT/ab_rings_grch38.log 2>&1; grep "k_map ms" $OUT/ab_rings_grch38.log | cut -c1-200
SPREAD_LIB=$V/libunc_dbgseed.so timeout 1200 python tools/dev/grch38_phase_spread.py 250000 3 auto > $OUT/dbgseed_grch38_new_rings.log 2>&1; grep -v "build_index" $OUT/dbgseed_grch38_new_rings.log | cut -c1-900 | tail -24
