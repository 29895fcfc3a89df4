#!/bin/bash
# round 5, final call: final_run.sh stages, then the GRCh38 A/B against the round-4 kernel and the 10 240-read parity sweeps on the index the bench left in /tmp
ROOT=$(pwd); OUT=$ROOT/gpurun_out/final; mkdir -p $OUT
bash tools/dev/final_run.sh tests smoke pmc bench stats
V=uncalled_amd/variants
AB_NOPROF=1 AB_RUNS=3 timeout 600 python tools/dev/ab_libs.py 50000:grch38 $V/libunc_base.so uncalled_amd/libuncalled_hip.so $V/libunc_base.so uncalled_amd/libuncalled_hip.so > $OUT/ab_grch38_final.log 2>&1; grep -v "^{" $OUT/ab_grch38_final.log | tail -5
timeout 900 python tests/dev/parity_sweep.py grch38 10240 64 > $OUT/parity_sweep_grch38.log 2> $OUT/parity_sweep_grch38.err; tail -c 900 $OUT/parity_sweep_grch38.log
timeout 400 python tests/dev/parity_sweep.py chr20 10240 64 > $OUT/parity_sweep_chr20.log 2> $OUT/parity_sweep_chr20.err; tail -c 900 $OUT/parity_sweep_chr20.log
find $ROOT/gpurun_out -type f -size +6M -delete
