#!/bin/bash
# round 6, call 2: is k_map bound by a per-wavefront resource (issue, latency) or by a shared one (memory system)?
# k_map time against the number of resident wavefronts, reads in flight = 4 x wavefronts; then the instruction cache counters.
ROOT=$(pwd); OUT=$ROOT/gpurun_out/c2; mkdir -p $OUT
L=uncalled_amd/libuncalled_hip.so
AB_NOPROF=1 AB_RUNS=2 timeout 900 python tools/dev/ab_libs.py 50000 $L $L@0@0@0@1024 $L@0@0@0@2048 $L@0@0@0@3072 $L@0@0@0@4096 $L@8192@0@0@2048 $L@16384@0@0@2048 > $OUT/ab_waves.log 2>&1; grep -v "^{" $OUT/ab_waves.log | tail -8
cd /tmp; export TMPDIR=/tmp
AB_NOPROF=1 AB_RUNS=1 timeout 600 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_DCACHE_REQ SQC_DCACHE_MISSES SQ_IFETCH SQ_IFETCH_LEVEL --kernel-trace --output-format csv -d $OUT/sqc -o pmc -- python $ROOT/tools/dev/ab_libs.py 50000 $ROOT/$L > $OUT/sqc.log 2>&1; tail -1 $OUT/sqc.log
cd $ROOT
python - <<PY
import csv, glob, collections
tot = collections.defaultdict(float)
for f in glob.glob("$OUT/sqc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_map" in r.get("Kernel_Name", ""):
            tot[r["Counter_Name"]] += float(r["Counter_Value"])
for k in sorted(tot): print("%-45s %.6g" % (k, tot[k]))
PY
find $ROOT/gpurun_out -type f -size +4M -delete
