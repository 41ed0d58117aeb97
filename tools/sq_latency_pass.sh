#!/bin/bash
# Where a kernel's waits go: average VMEM / LDS / instruction-fetch latencies from the SQ "level" counters (accumulated in-flight
# counts / issued counts) and the instruction cache's hit rate. tools/sq_latency_pass.sh <tag> <leaf> <instances> <frames> <path> [mem_cap]
set -e -o pipefail
TAG=$1; LEAF=$2; N=$3; FR=$4; PATHSEL=${5:-auto}; MC=${6:-0}
R=$(pwd); OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() { rocprofv3 --pmc $2 --output-format csv -d $OUT/$1 -- python3 $R/tools/leaf_scaling.py $LEAF $N --frames $FR --path $PATHSEL --mem-cap $MC > $OUT/$1.log 2>&1 || echo "(pass $1 failed: see $1.log)"; }
run a "SQ_WAVE_CYCLES SQ_INSTS_VMEM SQ_INST_LEVEL_VMEM SQ_INSTS_LDS SQ_INST_LEVEL_LDS SQ_WAIT_ANY"
run b "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAIT_INST_ANY SQ_INSTS_SMEM SQ_INST_LEVEL_SMEM SQ_WAVE_CYCLES"
run c "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES"
run d "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCC_HIT_sum TCC_MISS_sum"
cd $R
python3 - "$OUT" "$LEAF" <<'PY'
import csv, glob, sys, collections
out, leaf = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
for f in glob.glob(out + "/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if leaf.lower() not in k.lower(): continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); calls[(k, r["Counter_Name"])] += 1
with open(out + "/summary.txt", "w") as fo:
    for k, c in agg.items():
        print(k, file=fo)
        v = {n: x / max(1, calls[(k, n)]) for n, x in c.items()}
        for n, x in sorted(v.items()):
            print(f"  {n:32s} {x:16.1f} per launch", file=fo)
        for a, b, what in (("SQ_INST_LEVEL_VMEM", "SQ_INSTS_VMEM", "cycles a VMEM instruction is in flight"),
                           ("SQ_INST_LEVEL_LDS", "SQ_INSTS_LDS", "cycles an LDS instruction is in flight"),
                           ("SQ_IFETCH_LEVEL", "SQ_IFETCH", "cycles an instruction fetch is in flight"),
                           ("SQ_INST_LEVEL_SMEM", "SQ_INSTS_SMEM", "cycles a scalar load is in flight")):
            if v.get(a) and v.get(b):
                print(f"  -> {v[a] / v[b]:10.1f} {what}", file=fo)
print(open(out + "/summary.txt").read())
PY
