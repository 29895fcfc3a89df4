#!/bin/bash
# round 6, call 9: the phases of slow and fast GRCh38 launches (cycle-counting instantiation, one mapper, eight launches of the 250 000-read batch)
ROOT=$(pwd); OUT=$ROOT/gpurun_out/c9; mkdir -p $OUT
timeout 1500 python tools/dev/grch38_phase_spread.py 250000 8 > $OUT/grch38_phase_spread.log 2>&1; tail -12 $OUT/grch38_phase_spread.log
