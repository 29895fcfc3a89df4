#!/bin/bash
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r05; mkdir -p $OUT
V=uncalled_amd/variants
(timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu_2.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_gpu_2.log); tail -3 $OUT/pytest_gpu_2.log
timeout 400 python tools/dev/ab_libs.py 50000 $V/libunc_base.so uncalled_amd/libuncalled_hip.so $V/libunc_base.so uncalled_amd/libuncalled_hip.so > $OUT/ab_waits2.log 2>&1; cat $OUT/ab_waits2.log | tail -8
AB_NOPROF=1 timeout 900 python tools/dev/ab_libs.py 50000:grch38 $V/libunc_base.so uncalled_amd/libuncalled_hip.so $V/libunc_base.so uncalled_amd/libuncalled_hip.so > $OUT/ab_grch38.log 2>&1; grep -v "^{" $OUT/ab_grch38.log | tail -5
timeout 600 python tests/dev/dbg_read.py grch38 tests/golden/sweep_reads_r05.npz 61770 > $OUT/dbg_61770.log 2>&1; tail -12 $OUT/dbg_61770.log
SPREAD_LEGS=fixed,auto,fixed2 timeout 1200 python tools/dev/grch38_spread.py 250000 3 > $OUT/grch38_spread2.log 2> $OUT/grch38_spread2.err; grep "==" $OUT/grch38_spread2.log
