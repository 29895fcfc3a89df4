#!/bin/bash
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r05; mkdir -p $OUT
V=uncalled_amd/variants
AB_NOPROF=1 AB_RUNS=3 timeout 600 python tools/dev/ab_libs.py 50000 $V/libunc_base.so uncalled_amd/libuncalled_hip.so $V/libunc_base.so uncalled_amd/libuncalled_hip.so > $OUT/ab_ecoli_3.log 2>&1; grep -v "^{" $OUT/ab_ecoli_3.log | tail -5
AB_NOPROF=1 AB_RUNS=3 timeout 1500 python tools/dev/ab_libs.py 50000:grch38 $V/libunc_base.so uncalled_amd/libuncalled_hip.so $V/libunc_base.so uncalled_amd/libuncalled_hip.so > $OUT/ab_grch38_3.log 2>&1; grep -v "^{" $OUT/ab_grch38_3.log | tail -5
