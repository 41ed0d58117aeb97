#!/bin/bash
# SQ counter pass of one leaf's kernel (own run, no trace domains mixed in): tools/sq_pass.sh <tag> <leaf> <instances> <frames> <path> [mem_cap]
set -e -o pipefail
TAG=$1; LEAF=$2; N=$3; FR=$4; PATHSEL=${5:-auto}; MC=${6:-0}; KSUB=${7:-$2}     # (KSUB: substring of the kernel names to keep; Faust leaves: zf_cbg ...)
R=$(pwd); OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d $OUT/a -- python3 $R/tools/leaf_scaling.py $LEAF $N --frames $FR --path $PATHSEL --mem-cap $MC > $OUT/a.log 2>&1
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/b -- python3 $R/tools/leaf_scaling.py $LEAF $N --frames $FR --path $PATHSEL --mem-cap $MC > $OUT/b.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_INSTS_FLAT SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_SMEM --output-format csv -d $OUT/c -- python3 $R/tools/leaf_scaling.py $LEAF $N --frames $FR --path $PATHSEL --mem-cap $MC > $OUT/c.log 2>&1 || echo "(pass c failed: see c.log)"
cd $R
python3 - "$OUT" "$KSUB" <<'PY'
import csv, glob, sys, collections
out, leaf = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter(); meta = {}
for f in glob.glob(out + "/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if leaf.lower() not in k.lower(): continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); calls[(k, r["Counter_Name"])] += 1
        meta[k] = (r["VGPR_Count"], r["Accum_VGPR_Count"], r["SGPR_Count"], r["Grid_Size"], r["Workgroup_Size"])
with open(out + "/summary.txt", "w") as fo:
    for k, c in agg.items():
        print(k, "VGPR/AGPR/SGPR/grid/wg", meta[k], file=fo)
        for n, v in sorted(c.items()):
            print(f"  {n:24s} {v / max(1, calls[(k, n)]):16.1f} per launch ({calls[(k, n)]} launches)", file=fo)
print(open(out + "/summary.txt").read())
PY
