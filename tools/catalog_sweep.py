"""Catalog sweep: every discovered leaf through the engine, one line per leaf.

    python tools/catalog_sweep.py [--instances 1024] [--frames 48000] [--out profiles/r01_catalog_sweep.json]

Leaves whose script reaches a host-only builtin (buffer messages, peer names) are reported as "host-assisted": the engine
refuses them loudly (ZAB_E_UNSUPPORTED) instead of running them with stubbed host calls. Timing is the engine's own HIP
events around the launches of one zab_process call (inputs resident in HBM), warm-up call discarded.
"""
from __future__ import annotations

import argparse
import json
import re
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT / "zorakaudio-experimental-plugins_amd"), str(ROOT)]


def main() -> int:
    import zabatch
    ap = argparse.ArgumentParser()
    ap.add_argument("--instances", type=int, default=1024)
    ap.add_argument("--frames", type=int, default=48000)
    ap.add_argument("--block", type=int, default=512)
    ap.add_argument("--out", default="")
    ap.add_argument("--only", default="", help="comma-separated leaf names (default: every built leaf)")
    args = ap.parse_args()
    lib = ROOT / "zorakaudio-experimental-plugins_amd" / "lib"
    leaves = sorted(p.stem for p in lib.glob("*.json") if not p.stem.startswith("fx_"))
    if args.only:
        leaves = [x for x in leaves if x in set(args.only.split(","))]
    rows = []
    for leaf in leaves:
        meta = zabatch.leaf_meta(leaf)
        nch = int(meta["nch"])
        row = {"leaf": leaf, "kind": meta.get("kind", "jsfx"), "nch": nch, "instances": args.instances, "frames": args.frames}
        if "msg" in meta.get("features", []):  # one engine = one message bus = one session's worth of plugins
            row["instances"] = min(args.instances, 256)
        run_one(zabatch, args, leaf, meta, nch, row, 0)
        m = re.search(r"needed >= (\d+)", row.get("status", ""))
        if m:                                  # fixed arena too small for this leaf: size it from the device's report
            cap = 1 << (int(m.group(1)) + 64).bit_length()
            row["instances"] = max(1, min(args.instances, (48 << 30) // (cap * 8)))
            row["mem_cap"] = cap
            run_one(zabatch, args, leaf, meta, nch, row, cap)
        rows.append(row)
        print(json.dumps(row), flush=True)
    base = ROOT / "profiles" / "r01_catalog_sweep.json"          # previous round's sweep: the ratio per leaf at the same batch size
    if base.exists():
        old = {r["leaf"]: r for r in json.loads(base.read_text())}
        for r in rows:
            o = old.get(r["leaf"])
            if o and r.get("kernel_ms") and o.get("kernel_ms") and o.get("instances") == r.get("instances") and o.get("frames") == r.get("frames"):
                r["speedup_vs_r01"] = round(o["kernel_ms"] / r["kernel_ms"], 2)
    if args.out:
        Path(args.out).write_text(json.dumps(rows, indent=1))
    return 0


def run_one(zabatch, args, leaf, meta, nch, row, mem_cap):
        n = row["instances"]
        try:
            with zabatch.Engine(leaf, n, max_block=args.block, mem_cap=mem_cap) as e:
                e.set_sliders(meta["default_sliders"])
                e.prepare()
                if nch == 0:
                    row["status"] = "no audio channels (MIDI-only leaf)"
                    return
                nbytes = n * nch * args.frames * 4
                d_in, d_out = e.device_alloc(nbytes), e.device_alloc(nbytes)
                e.device_noise(d_in, args.frames)
                e.process_device(d_in, d_out, args.frames, block=args.block); e.sync()       # warm-up
                e.process_device(d_in, d_out, args.frames, block=args.block); e.sync()
                ms, launches = e.last_timing()
                row.update(status="ok", kernel_ms=ms, launches=launches, fast_path=bool(e.used_fast_path()), kernel=e.last_kernel_name(),
                           msamples_per_s=n * nch * args.frames / (ms * 1e-3) / 1e6,
                           realtime_factor=args.frames / 48000.0 / (ms * 1e-3))
                if not row["fast_path"]:         # generic path: instances per wavefront and the LDS window over mem[] it ran with
                    row["instances_per_wave"], row["lds_mem_words"] = (int(x) for x in e.launch_shape())
                    row["mem_high"] = int(max(e.mem_high()))
        except zabatch.ZabError as ex:
            row["status"] = "host-assisted (refused)" if ex.code == -5 else f"error {ex.code}: {ex}"


if __name__ == "__main__":
    sys.exit(main())
