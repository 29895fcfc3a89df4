#!/bin/bash
# Parity at bench scale against the reference's object code (tests/dev/parity_sweep.py; GPU box, untimed), any of:
#   ecoli_all   all 50 000 reads of the E. coli bench batch, every reported field          (reference: ~10 min on 128 threads)
#   chr20       10 240 reads drawn across the 200 000-read chr20 batch
#   grch38      10 240 reads drawn across the 250 000-read GRCh38 batch                     (reference: ~7 min)
#   rt          the chunked path, 512 channels x 12 reads, teams of 8 and 1, E. coli and chr20 thresholds (4 x 6 144 reads)
# usage: bash tools/dev/parity_sweeps.sh <out dir under gpurun_out> [ecoli_all] [chr20] [grch38] [rt]
ROOT=$(pwd); OUT=$ROOT/gpurun_out/$1; shift; mkdir -p $OUT
for s in "$@"; do
case $s in
ecoli_all) timeout 1500 python tests/dev/parity_sweep.py ecoli 50000 128 > $OUT/parity_sweep_ecoli_all.log 2> $OUT/parity_sweep_ecoli_all.err; tail -c 700 $OUT/parity_sweep_ecoli_all.log; echo ;;
chr20)     timeout 1200 python tests/dev/parity_sweep.py chr20 10240 128 > $OUT/parity_sweep_chr20.log 2> $OUT/parity_sweep_chr20.err; tail -c 700 $OUT/parity_sweep_chr20.log; echo ;;
grch38)    timeout 1800 python tests/dev/parity_sweep.py grch38 10240 128 > $OUT/parity_sweep_grch38.log 2> $OUT/parity_sweep_grch38.err; tail -c 700 $OUT/parity_sweep_grch38.log; echo ;;
rt)
  for ref in ecoli chr20; do for team in 8 1; do
    T=""; [ $team = 1 ] && T="UNC_RT_TEAM=1"
    env $T timeout 900 python tests/dev/parity_sweep.py rt:$ref 12 100 > $OUT/parity_sweep_rt_${ref}_team$team.log 2> $OUT/parity_sweep_rt_${ref}_team$team.err; tail -c 420 $OUT/parity_sweep_rt_${ref}_team$team.log; echo
  done; done ;;
esac
done
find $ROOT/gpurun_out -type f -size +6M -delete
