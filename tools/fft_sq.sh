#!/bin/bash
# SQ counter passes of the FFT harness' kernel: tools/fft_sq.sh <tag>
set -e -o pipefail
TAG=${1:-rXX}; R=$(pwd); OUT=$R/gpurun_out/${TAG}_fft_sq; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d $OUT/a -- python3 $R/tools/fft_one.py 2048 4096 4 1 > $OUT/a.log 2>&1
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/b -- python3 $R/tools/fft_one.py 2048 4096 4 1 > $OUT/b.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_INSTS_FLAT SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_SMEM --output-format csv -d $OUT/c -- python3 $R/tools/fft_one.py 2048 4096 4 1 > $OUT/c.log 2>&1
cd $R
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter(); meta = {}
for f in glob.glob(out + "/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "fftbench" not in k or "tail" in k or "prepare" in k: continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); calls[(k, r["Counter_Name"])] += 1
        meta[k] = (r["VGPR_Count"], r["Accum_VGPR_Count"], r["SGPR_Count"], r["LDS_Block_Size"], r["Grid_Size"], r["Workgroup_Size"])
with open(out + "/summary.txt", "w") as fo:
    for k, c in agg.items():
        print(k, "VGPR/AGPR/SGPR/LDS/grid/wg", meta[k], file=fo)
        for n, v in sorted(c.items()):
            print(f"  {n:24s} {v / max(1, calls[(k, n)]):16.1f} per launch ({calls[(k, n)]} launches)", file=fo)
print(open(out + "/summary.txt").read())
PY
