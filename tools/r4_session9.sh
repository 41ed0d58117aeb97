#!/bin/bash
# round 4, ninth GPU session: where Sample's waits go (SQ level counters), the headline with the shorter meter window, and the
# catalog sweep after named constants became literals.
O=gpurun_out; mkdir -p $O
timeout -k 10 400 bash tools/sq_latency_pass.sh s9_sample_lat Sample 1 2000 generic 524288 > $O/s9_lat.log 2>&1; echo "latency pass rc=$?" | tee $O/s9_summary.txt
tail -45 $O/s9_lat.log
timeout -k 10 300 python bench.py > $O/s9_bench.json 2> $O/s9_bench.err; echo "bench rc=$?" | tee -a $O/s9_summary.txt
cut -c1-900 $O/s9_bench.json
timeout -k 10 900 python tools/catalog_sweep.py --cpu-seconds 0 --out $O/s9_sweep.json > $O/s9_sweep.log 2>&1; echo "sweep rc=$?" | tee -a $O/s9_summary.txt
python - <<'PY'
import json
new = {r["leaf"]: r for r in json.load(open("gpurun_out/s9_sweep.json"))["rows"]} if isinstance(json.load(open("gpurun_out/s9_sweep.json")), dict) else {r["leaf"]: r for r in json.load(open("gpurun_out/s9_sweep.json"))}
old_raw = json.load(open("profiles/r04_catalog_sweep.json"))
old = {r["leaf"]: r for r in (old_raw["rows"] if isinstance(old_raw, dict) else old_raw)}
for k, r in sorted(new.items()):
    o = old.get(k, {})
    print(f"{k:22s} {r.get('kernel_ms')!s:>10} ms (r04 {o.get('kernel_ms')!s:>10}) {r.get('kernel','')} {r.get('status','')[:60]}")
PY
