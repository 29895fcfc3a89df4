#!/bin/bash
# Dev tool (GPU box): k_map time of ONE library under different widths of the seed-cluster bucket grid (UNC_BUCKET_SHIFT is read
# when the index is loaded).  Usage: tools/dev/ab_bucket_shift.sh <lib.so> <n_reads[:workload]> <shift> [<shift> ...]
LIB=$1; W=$2; shift 2
echo "== default shift, $W"; AB_NOPROF=1 python tools/dev/ab_libs.py $W $LIB 2>&1 | grep k_map
for s in "$@"; do
  echo "== UNC_BUCKET_SHIFT=$s, $W"; UNC_BUCKET_SHIFT=$s AB_NOPROF=1 python tools/dev/ab_libs.py $W $LIB 2>&1 | grep k_map
done
