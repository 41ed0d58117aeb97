"""Catalog sweep: every discovered leaf through the engine, one line per leaf.

    python tools/catalog_sweep.py [--instances 1024] [--frames 48000] [--out profiles/r03_catalog_sweep.json] [--cpu-seconds 2]

Per row, beside the device numbers: `cpu_port_msamples_16c` -- the leaf's CPU checker (oracle/port.py, or oracle/faust_ref.c for
the Faust leaves) on this box's host cores, one instance per thread for --cpu-seconds; `vs_cpu` = device rate / that;
(a measurement harness like bench.py's `cpu_baseline` leg: the checker is timed beside the device, never in its place; with
--cpu-seconds 0 nothing under oracle/ is touched) `hbm_frac` = the audio a launch must move (4 B in + 4 B out per channel and frame) / kernel time / 8 TB/s; `speedup_vs_r02` =
the previous round's kernel time for the same batch / this one.

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
    ap.add_argument("--cpu-seconds", type=float, default=2.0, help="budget of the CPU checker per leaf (0: skip)")
    ap.add_argument("--arena-gb", type=int, default=192, help="HBM the mem[] arenas of one leaf's batch may take (rounds 1-3: 48)")
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
            # (arenas of tens of millions of cells: as many instances as two thirds of the card's 288 GB hold -- a wavefront's run
            #  time does not depend on how many others run until the chip is full, so a small batch only understates the kernel)
            row["instances"] = max(1, min(args.instances, (args.arena_gb << 30) // (cap * 8)))
            row["mem_cap"] = cap
            run_one(zabatch, args, leaf, meta, nch, row, cap)
        if row.get("kernel_ms"):
            row["hbm_frac"] = round(row["instances"] * nch * args.frames * 8 / (row["kernel_ms"] * 1e-3) / 8.0e12, 6)
            if args.cpu_seconds > 0:
                cpu = cpu_rate(leaf, meta, nch, args.cpu_seconds, args.block)
                if cpu:
                    row["cpu_port_msamples_16c"], row["cpu_cores"] = round(cpu[0], 2), cpu[1]
                    row["vs_cpu"] = round(row["msamples_per_s"] / cpu[0], 1)
        rows.append(row)
        print(json.dumps(row), flush=True)
    base = ROOT / "profiles" / "r02_catalog_sweep.json"          # previous round's sweep: the ratio per leaf at the same batch size
    if base.exists():
        old = {r["leaf"]: r for r in json.loads(base.read_text())}
        for r in rows:
            o = old.get(r["leaf"])
            if o and r.get("kernel_ms") and o.get("kernel_ms") and o.get("instances") == r.get("instances") and o.get("frames") == r.get("frames"):
                r["speedup_vs_r02"] = round(o["kernel_ms"] / r["kernel_ms"], 2)
                r["r02_kernel_ms"] = round(o["kernel_ms"], 3)
    if args.out:
        Path(args.out).write_text(json.dumps(rows, indent=1))
    return 0


def cpu_rate(leaf, meta, nch, seconds, block):
    """(Msamples/s over all threads, threads) of the leaf's CPU checker: one instance per thread, 24 000-frame passes of noise."""
    import os
    import time
    from concurrent.futures import ThreadPoolExecutor
    from zajit import noise
    try:
        if meta.get("kind") == "faust":
            from oracle import faust_ref
            fresh = lambda: faust_ref.FaustRef(leaf, 48000.0)
            run = lambda o, x: o.compute(x, meta["default_sliders"][:8], block=block)
        else:
            from oracle import port
            if not port.port_path(leaf).exists():
                return None

            def fresh():
                p = port.Port(leaf, 48000.0, mem_cap=1 << 25)
                p.set_sliders(meta["default_sliders"]); p.prepare()
                return p
            run = lambda o, x: o.process(x, block)
        cores = max(1, min(os.cpu_count() or 1, 16))
        frames = 24000
        x = noise.white_noise(range(cores), frames, channels=max(nch, 2))[:, :nch]
        x = np.ascontiguousarray(x)

        def work(i):
            o = fresh()
            t, reps = time.perf_counter(), 0
            while True:
                run(o, x[i]); reps += 1
                if time.perf_counter() - t > seconds:
                    return reps
        t0 = time.perf_counter()
        with ThreadPoolExecutor(cores) as ex:
            reps = list(ex.map(work, range(cores)))
        return sum(reps) * frames * nch / (time.perf_counter() - t0) / 1e6, cores
    except Exception as ex:        # noqa: BLE001  (a checker that cannot run this leaf here: the row simply has no CPU figure)
        print(f"# cpu checker of {leaf}: {ex}", file=sys.stderr)
        return None


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
                if row["kind"] == "jsfx" and row["fast_path"]:     # a generated time-parallel kernel: how much of the launch it handed to the serial tail
                    row["handed_back_instances"], row["handed_back_frames"] = (int(x) for x in e.handback())
                if not row["fast_path"]:         # generic path: instances per wavefront and the LDS window over mem[] it ran with
                    row["instances_per_wave"], row["lds_mem_words"] = (int(x) for x in e.launch_shape())
                    row["mem_high"] = int(max(e.mem_high()))
        except zabatch.ZabError as ex:
            row["status"] = "host-assisted (refused)" if ex.code == -5 else f"error {ex.code}: {ex}"


if __name__ == "__main__":
    sys.exit(main())
