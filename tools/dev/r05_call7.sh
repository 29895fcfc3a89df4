#!/bin/bash
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r05; mkdir -p $OUT
(timeout 1500 python -m pytest tests -m gpu -q > $OUT/pytest_gpu_final.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_gpu_final.log); tail -4 $OUT/pytest_gpu_final.log
timeout 400 python tools/dev/check_sorter.py chr20 > $OUT/check_sorter_chr20.log 2>&1; tail -5 $OUT/check_sorter_chr20.log
