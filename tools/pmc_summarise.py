"""Reduce the rocprofv3 outputs of tools/profile_round.sh to the files kept under profiles/.

    python tools/pmc_summarise.py gpurun_out r01
writes  <dir>/<tag>_bench_kernel_stats.csv  (copy of the --stats kernel summary)
        <dir>/<tag>_pmc_traffic.json        (FETCH_SIZE / WRITE_SIZE per launch of the dominant kernel, corrected as
                                             MI355X_MICROARCH.md prescribes: FETCH_SIZE x2 on gfx950, KB = 1024 B)
"""
import csv
import json
import shutil
import sys
from pathlib import Path


def counter_mean(d: Path, counter: str, kernel_sub: str):
    vals = []
    for f in d.rglob("*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter and kernel_sub in r["Kernel_Name"]:
                vals.append(float(r["Counter_Value"]))
    return (sum(vals) / len(vals), len(vals)) if vals else (None, 0)


def main():
    out, tag = Path(sys.argv[1]), sys.argv[2]
    bench = json.loads((out / f"{tag}_bench.json").read_text().strip().splitlines()[-1])
    kern = bench["roofline"]["kernel"]
    for f in (out / f"{tag}_trace").rglob("*kernel_stats.csv"):
        shutil.copy(f, out / f"{tag}_bench_kernel_stats.csv")
    fetch, nf = counter_mean(out / f"{tag}_pmc_fetch", "FETCH_SIZE", kern)
    write, nw = counter_mean(out / f"{tag}_pmc_write", "WRITE_SIZE", kern)
    rec = {"kernel": kern, "leaf": bench["config"].get("leaf", "DDT"), "instances": bench["config"]["instances_rank0"], "frames": bench["config"]["frames_per_step"],
           "fast": True, "FETCH_SIZE_KB_mean": fetch, "WRITE_SIZE_KB_mean": write, "launches_sampled": [nf, nw],
           "correction": "FETCH_SIZE x2 (gfx950 reports half of wide coalesced reads), WRITE_SIZE as is; KB = 1024 B"}
    if fetch is not None and write is not None:
        rec["hbm_bytes_per_launch"] = (2.0 * fetch + write) * 1024.0
        rec["algorithmic_bytes_per_launch"] = bench["roofline"]["algorithmic_bytes_per_launch"]
        rec["ratio"] = rec["hbm_bytes_per_launch"] / rec["algorithmic_bytes_per_launch"]
    (out / f"{tag}_pmc_traffic.json").write_text(json.dumps(rec, indent=1))
    print(json.dumps(rec))


if __name__ == "__main__":
    main()
