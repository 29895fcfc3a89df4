#!/bin/bash
# round 5, first GPU call: GPU suite on the new tree, E. coli A/B baseline, GRCh38 launch spread (pool by rule vs by need), bench-scale parity sweeps
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r05; mkdir -p $OUT
(timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_gpu.log); tail -3 $OUT/pytest_gpu.log
AB_NOPROF=1 timeout 300 python tools/dev/ab_libs.py 50000 uncalled_amd/libuncalled_hip.so > $OUT/ab_base.log 2>&1; tail -2 $OUT/ab_base.log
(rocprofv3 -L 2>/dev/null | grep -i -E "utcl|tlb|xnack|translat" | head -60) > $OUT/counters_translation.txt 2>&1; wc -l $OUT/counters_translation.txt
timeout 1200 python tools/dev/grch38_spread.py 250000 4 > $OUT/grch38_spread.log 2> $OUT/grch38_spread.err; grep "==" $OUT/grch38_spread.log
timeout 1200 python tests/dev/parity_sweep.py grch38 10240 64 > $OUT/parity_sweep_grch38.log 2> $OUT/parity_sweep_grch38.err; tail -c 600 $OUT/parity_sweep_grch38.log
timeout 600 python tests/dev/parity_sweep.py chr20 10240 64 > $OUT/parity_sweep_chr20.log 2> $OUT/parity_sweep_chr20.err; tail -c 600 $OUT/parity_sweep_chr20.log
du -sh $ROOT/gpurun_out
