#!/bin/bash
# round 6, call 16: one pair of scheduler rings per XCD (no L2 write-back / invalidate at park and resume) against one pair for the device, same library
ROOT=$(pwd); OUT=$ROOT/gpurun_out/c16; mkdir -p $OUT
L=uncalled_amd/libuncalled_hip.so
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > $OUT/pytest_gpu_parity.log 2>&1; echo "pytest exit $?"; tail -3 $OUT/pytest_gpu_parity.log
AB_NOPROF=1 timeout 600 python tools/dev/ab_libs.py 50000 $L@0@0@0@0@1 $L $L@0@0@0@0@1 $L > $OUT/ab_rings_per_xcd_ecoli.log 2>&1; grep "k_map ms" $OUT/ab_rings_per_xcd_ecoli.log | cut -c1-220
AB_NOPROF=1 timeout 900 python tools/dev/ab_libs.py 200000:chr20 $L@0@0@0@0@1 $L > $OUT/ab_rings_per_xcd_chr20.log 2>&1; grep "k_map ms" $OUT/ab_rings_per_xcd_chr20.log | cut -c1-220
AB_NOPROF=1 timeout 1500 python tools/dev/ab_libs.py 250000:grch38 $L@0@0@0@0@1 $L > $OUT/ab_rings_per_xcd_grch38.log 2>&1; grep "k_map ms" $OUT/ab_rings_per_xcd_grch38.log | cut -c1-220
