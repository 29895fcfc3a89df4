#!/bin/bash
# round 6, call 1: lane utilisation of the shipped k_map (group c) beside the wave-cycle shares and the instruction mix,
# phase shares of the profiling instantiation, and the E. coli index taken home (data/ecoli_p.* saves its build in later calls)
ROOT=$(pwd); OUT=$ROOT/gpurun_out/c1; mkdir -p $OUT
timeout 900 python tools/dev/ab_libs.py 50000 uncalled_amd/libuncalled_hip.so > $OUT/ab_base.log 2>&1; tail -3 $OUT/ab_base.log
mkdir -p $ROOT/gpurun_out/data
for f in /tmp/ub/ecoli_syn.*; do cp $f $ROOT/gpurun_out/data/ecoli_p.${f##*.}; done
ls -la $ROOT/gpurun_out/data
bash tools/dev/pmc_sq.sh gpurun_out/c1/pmc_ecoli uncalled_amd/libuncalled_hip.so 50000 a b c d > $OUT/pmc.log 2>&1; tail -30 $OUT/pmc.log
find $ROOT/gpurun_out -type f -size +8M -delete
