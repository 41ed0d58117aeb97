#!/bin/bash
# HBM traffic of the FFT harness' kernel (config C3 ii), separate --pmc passes as for the headline kernel: tools/fft_traffic.sh <tag>
set -e -o pipefail
TAG=${1:-rXX}; R=$(pwd); OUT=$R/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_fft_pmc_fetch -- python3 $R/tools/fft_one.py 2048 4096 4 1 > $OUT/${TAG}_fft_pmc.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_fft_pmc_write -- python3 $R/tools/fft_one.py 2048 4096 4 1 >> $OUT/${TAG}_fft_pmc.log 2>&1
cd $R
python3 - $OUT $TAG <<'PY'
import csv, glob, json, sys
out, tag = sys.argv[1], sys.argv[2]
def mean(d, name):
    v = [float(r["Counter_Value"]) for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True) for r in csv.DictReader(open(f))
         if r["Counter_Name"] == name and "fftbench" in r["Kernel_Name"] and ("tpar" in r["Kernel_Name"] or "process" in r["Kernel_Name"]) and "tail" not in r["Kernel_Name"]]
    return (sum(v) / len(v), len(v)) if v else (None, 0)
f, nf = mean(f"{out}/{tag}_fft_pmc_fetch", "FETCH_SIZE"); w, nw = mean(f"{out}/{tag}_fft_pmc_write", "WRITE_SIZE")
alg = 2048 * 4096 * 16 * 2 * 2 * 4          # buffers x points x 16 B x (read + write) x 2 fused operations x K = 4 round trips per launch
res = {"case": "fx_fftbench 2048 x 4096 points, K = 4 fused round trips per launch", "FETCH_SIZE_KB_mean": f, "WRITE_SIZE_KB_mean": w, "launches_sampled": [nf, nw],
       "correction": "FETCH_SIZE x2 (gfx950), WRITE_SIZE as is; KB = 1024 B", "hbm_bytes_per_launch": (2 * f + w) * 1024 if f and w else None,
       "algorithmic_bytes_per_launch": alg}
if res["hbm_bytes_per_launch"]:
    res["ratio"] = res["hbm_bytes_per_launch"] / alg
json.dump(res, open(f"{out}/{tag}_fft_pmc_traffic.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
