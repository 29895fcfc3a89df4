#!/bin/bash
# round 6, call 12: do E. coli and chr20 launches show the XCDs that decide few reads as well?  which reads are those?
ROOT=$(pwd); OUT=$ROOT/gpurun_out/c12; mkdir -p $OUT
SPREAD_WORKLOAD=ecoli timeout 600 python tools/dev/grch38_phase_spread.py 50000 3 auto > $OUT/phase_spread_ecoli.log 2>&1; grep -v "build_index" $OUT/phase_spread_ecoli.log | cut -c1-900 | tail -30
SPREAD_WORKLOAD=chr20 timeout 900 python tools/dev/grch38_phase_spread.py 200000 2 auto > $OUT/phase_spread_chr20.log 2>&1; grep -v "build_index" $OUT/phase_spread_chr20.log | cut -c1-900 | tail -24
