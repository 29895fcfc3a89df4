#!/bin/bash
# round 6, call 7: the GPU suite (scale cases now 4 096 / 1 024 reads), then parity at bench scale: ALL 50 000 E. coli reads against the
# reference's object code, and the chunked path on 512 channels x 12 reads (teams of 8 and 1, both threshold sets) against its chunk path
ROOT=$(pwd); OUT=$ROOT/gpurun_out/c7; mkdir -p $OUT
(timeout 1800 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_gpu.log); tail -3 $OUT/pytest_gpu.log
timeout 1500 python tests/dev/parity_sweep.py ecoli 50000 128 > $OUT/parity_sweep_ecoli_all.log 2> $OUT/parity_sweep_ecoli_all.err; tail -c 600 $OUT/parity_sweep_ecoli_all.log; echo
timeout 900 python tests/dev/parity_sweep.py rt:ecoli 12 100 > $OUT/parity_sweep_rt_ecoli_team8.log 2> $OUT/parity_sweep_rt_ecoli_team8.err; tail -c 500 $OUT/parity_sweep_rt_ecoli_team8.log; echo
UNC_RT_TEAM=1 timeout 900 python tests/dev/parity_sweep.py rt:ecoli 12 100 > $OUT/parity_sweep_rt_ecoli_team1.log 2> $OUT/parity_sweep_rt_ecoli_team1.err; tail -c 500 $OUT/parity_sweep_rt_ecoli_team1.log; echo
timeout 900 python tests/dev/parity_sweep.py rt:chr20 12 100 > $OUT/parity_sweep_rt_chr20_team8.log 2> $OUT/parity_sweep_rt_chr20_team8.err; tail -c 500 $OUT/parity_sweep_rt_chr20_team8.log; echo
UNC_RT_TEAM=1 timeout 900 python tests/dev/parity_sweep.py rt:chr20 12 100 > $OUT/parity_sweep_rt_chr20_team1.log 2> $OUT/parity_sweep_rt_chr20_team1.err; tail -c 500 $OUT/parity_sweep_rt_chr20_team1.log; echo
find $ROOT/gpurun_out -type f -size +6M -delete
