#!/bin/bash
# HBM traffic of the dominant kernel of ONE bench.py configuration: two rocprofv3 passes (FETCH_SIZE, WRITE_SIZE -- separate runs,
# no trace domains mixed in), reduced to gpurun_out/pmc_<kernel>_<instances>x<frames>.json (copy it into profiles/: bench.py picks
# it up by kernel name and batch shape and reports it as roofline.traffic).
#     bash tools/pmc_pass.sh <tag> [bench.py arguments, e.g. --leaf ClickBeGoneSG --instances-total 1024 --frames 48000]
set -e -o pipefail
TAG=${1:?tag}; shift
R=$(pwd); OUT=$R/gpurun_out; mkdir -p $OUT
python3 bench.py --no-cpu-baseline --steps 3 --warmup 1 "$@" > $OUT/${TAG}_bench.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_pmc_fetch -- python3 $R/bench.py --no-cpu-baseline --steps 2 --warmup 1 "$@" > /dev/null
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_pmc_write -- python3 $R/bench.py --no-cpu-baseline --steps 2 --warmup 1 "$@" > /dev/null
cd $R
python3 - "$OUT" "$TAG" <<'PY'
import json, sys
from pathlib import Path
sys.path.insert(0, "tools")
from pmc_summarise import counter_mean
out, tag = Path(sys.argv[1]), sys.argv[2]
bench = json.loads((out / f"{tag}_bench.json").read_text().strip().splitlines()[-1])
kern = bench["roofline"]["kernel"]
fetch, nf = counter_mean(out / f"{tag}_pmc_fetch", "FETCH_SIZE", kern)
write, nw = counter_mean(out / f"{tag}_pmc_write", "WRITE_SIZE", kern)
rec = {"kernel": kern, "leaf": bench["config"].get("leaf", "DDT"), "instances": bench["config"]["instances_rank0"],
       "frames": bench["config"]["frames_per_step"], "fast": True, "FETCH_SIZE_KB_mean": fetch, "WRITE_SIZE_KB_mean": write,
       "launches_sampled": [nf, nw], "kernel_ms": bench["roofline"]["kernel_ms"],
       "correction": "FETCH_SIZE x2 (gfx950 reports half of wide coalesced reads), WRITE_SIZE as is; KB = 1024 B"}
if fetch is not None and write is not None:
    rec["hbm_bytes_per_launch"] = (2.0 * fetch + write) * 1024.0
    rec["algorithmic_bytes_per_launch"] = bench["roofline"]["algorithmic_bytes_per_launch"]
    rec["ratio"] = rec["hbm_bytes_per_launch"] / rec["algorithmic_bytes_per_launch"]
name = f"pmc_{kern}_{rec['instances']}x{rec['frames']}.json"
(out / name).write_text(json.dumps(rec, indent=1))
print(name, json.dumps(rec))
PY
