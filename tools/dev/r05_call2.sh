#!/bin/bash
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r05; mkdir -p $OUT
V=uncalled_amd/variants
AB_NOPROF=1 timeout 400 python tools/dev/ab_libs.py 50000 $V/libunc_base.so uncalled_amd/libuncalled_hip.so $V/libunc_fm2.so $V/libunc_base.so uncalled_amd/libuncalled_hip.so $V/libunc_fm2.so > $OUT/ab_waits.log 2>&1; grep -v "^{" $OUT/ab_waits.log | tail -8
timeout 300 python tests/dev/dump_reads.py chr20 $OUT/dump_chr20.npz 37043 > $OUT/dump_chr20.log 2>&1; tail -3 $OUT/dump_chr20.log
timeout 400 python tests/dev/dump_reads.py grch38 $OUT/dump_grch38.npz 66882 61770 > $OUT/dump_grch38.log 2>&1; tail -3 $OUT/dump_grch38.log
(timeout 600 python -m pytest tests/test_gpu_index_build.py -m gpu -x -q > $OUT/pytest_gpu_index.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_gpu_index.log); tail -3 $OUT/pytest_gpu_index.log
