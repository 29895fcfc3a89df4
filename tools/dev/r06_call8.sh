#!/bin/bash
# round 6, call 8: what made GRCh38's launch times come in two levels?  Round 5's library, the round-6 tree with ONLY the wide merge's
# run offsets and staging put back as they were (scratch-resident adj, a scratch load and a full wait before every key load), and the
# round-6 tree: three launches of the 250 000-read batch each; then the round-6 tree's phase shares on the same batch
ROOT=$(pwd); OUT=$ROOT/gpurun_out/c8; mkdir -p $OUT
V=uncalled_amd/variants
AB_NOPROF=1 AB_RUNS=3 timeout 2400 python tools/dev/ab_libs.py 250000:grch38 $V/libunc_base.so $V/libunc_oldkaw.so uncalled_amd/libuncalled_hip.so $V/libunc_oldkaw.so > $OUT/ab_grch38_cause.log 2>&1; grep -v "^{" $OUT/ab_grch38_cause.log | tail -5
AB_RUNS=1 timeout 900 python tools/dev/ab_libs.py 250000:grch38 uncalled_amd/libuncalled_hip.so > $OUT/grch38_phases.log 2>&1; tail -2 $OUT/grch38_phases.log
