#!/bin/bash
# round 4, GPU session 16: map loops load a group ahead of the previous group's stores; only whole-number constants are literals.
O=gpurun_out; mkdir -p $O
timeout -k 10 1100 python -m pytest tests -m gpu -q --maxfail=30 -p no:cacheprovider > $O/s16_gpu_tests.log 2>&1; echo "gpu tests rc=$?" | tee $O/s16_summary.txt
tail -5 $O/s16_gpu_tests.log
timeout -k 10 300 python tools/stft_parts.py 1024 > $O/s16_stft_parts.txt 2>&1; cat $O/s16_stft_parts.txt
timeout -k 10 900 python tools/catalog_sweep.py --cpu-seconds 0 --out $O/s16_sweep.json > $O/s16_sweep.log 2>&1; echo "sweep rc=$?" | tee -a $O/s16_summary.txt
python - <<'PY'
import json
new = {r["leaf"]: r for r in json.load(open("gpurun_out/s16_sweep.json"))}
old = {r["leaf"]: r for r in json.load(open("profiles/r04_catalog_sweep.json"))}
for k, r in sorted(new.items()):
    o = old.get(k, {})
    print(f"{k:22s} {r.get('kernel_ms')!s:>10.9} ms (r04 {o.get('kernel_ms')!s:>10.9}) {r.get('kernel','')} {r.get('status','')[:60]}")
PY

