"""BASELINE config C4 in its literal shape: EVERY leaf of the catalog resident at once, one engine (own stream) per leaf with
K instances each, all launched together for the same stretch of audio; wall time from the first launch to the last completion.

    python tools/catalog_mixed.py [--per-leaf 1,32] [--seconds 10] [--out profiles/r03_catalog_mixed.json]
"""
from __future__ import annotations

import argparse
import json
import re
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT / "zorakaudio-experimental-plugins_amd"), str(ROOT)]


def main() -> int:
    import zabatch
    ap = argparse.ArgumentParser()
    ap.add_argument("--per-leaf", default="1,32")
    ap.add_argument("--seconds", type=float, default=10.0)
    ap.add_argument("--block", type=int, default=512)
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    ks = [int(x) for x in args.per_leaf.split(",")]
    if len(ks) > 1:
        # one process per batch size: a queue keeps the scratch of the largest kernel it ever ran (Sample: 36 KB per lane), and the
        # second pass's 32 new queues on top of the first pass's ran the device out of scratch (HSA_STATUS_ERROR_OUT_OF_RESOURCES).
        # This process touches no GPU itself.
        import subprocess
        import tempfile
        rows = []
        for k in ks:
            with tempfile.NamedTemporaryFile(suffix=".json") as tf:
                subprocess.run([sys.executable, __file__, "--per-leaf", str(k), "--seconds", str(args.seconds), "--block", str(args.block),
                                "--out", tf.name], check=True)
                rows += json.loads(Path(tf.name).read_text())
        if args.out:
            Path(args.out).write_text(json.dumps(rows, indent=1))
        return 0
    lib = ROOT / "zorakaudio-experimental-plugins_amd" / "lib"
    leaves = sorted(p.stem for p in lib.glob("*.json") if not p.stem.startswith("fx_"))
    frames = int(args.seconds * 48000)
    out = []
    for k in ks:
        engines, skipped = [], []
        for leaf in leaves:
            meta = zabatch.leaf_meta(leaf)
            nch = int(meta["nch"])
            if nch == 0:
                skipped.append(leaf)
                continue
            cap = 0
            for _try in range(2):
                try:
                    e = zabatch.Engine(leaf, k, max_block=args.block, mem_cap=cap)
                    e.set_sliders(meta["default_sliders"]); e.prepare()
                    nbytes = k * nch * frames * 4
                    d_in, d_out = e.device_alloc(nbytes), e.device_alloc(nbytes)
                    e.device_noise(d_in, frames)
                    e.process_device(d_in, d_out, 4096, block=args.block); e.sync()        # warm-up (also sizes the arena)
                    engines.append((leaf, e, d_in, d_out, nch))
                    break
                except zabatch.ZabError as ex:
                    m = re.search(r"needed >= (\d+)", str(ex))
                    if m and not cap:
                        cap = 1 << (int(m.group(1)) + 64).bit_length()
                        continue
                    skipped.append(f"{leaf}: {str(ex)[:60]}")
                    break
        t0 = time.perf_counter()
        for leaf, e, d_in, d_out, nch in engines:
            e.process_device(d_in, d_out, frames, block=args.block)
        for leaf, e, *_ in engines:
            e.sync()
        wall = time.perf_counter() - t0
        per = {}
        for leaf, e, d_in, d_out, nch in engines:
            ms, launches = e.last_timing()
            per[leaf] = {"kernel_ms": round(ms, 3), "fast_path": bool(e.used_fast_path()), "kernel": e.last_kernel_name()}
        samples = sum(k * nch * frames for _, _, _, _, nch in engines)
        row = {"instances_per_leaf": k, "leaves": len(engines), "skipped": skipped, "audio_seconds": args.seconds, "wall_s": round(wall, 4),
               "realtime_factor": round(args.seconds / wall, 2), "msamples_per_s": round(samples / wall / 1e6, 1), "per_leaf": per}
        print(json.dumps({kk: v for kk, v in row.items() if kk != "per_leaf"}), flush=True)
        out.append(row)
        for _, e, *_ in engines:
            e.close()
    if args.out:
        Path(args.out).write_text(json.dumps(out, indent=1))
    return 0


if __name__ == "__main__":
    sys.exit(main())
