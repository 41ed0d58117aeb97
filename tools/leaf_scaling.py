"""Instance-count scaling of one leaf (default: whichever path the engine picks; --path generic / fast pins it):

    python tools/leaf_scaling.py ERBTilt 1024 4096 16384 [--frames 12000] [--ipw K] [--path generic]

Prints kernel time and aggregate real-time factor (instance-seconds of audio per second) per instance count.
"""
import argparse, json, os, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT / "zorakaudio-experimental-plugins_amd"), str(ROOT)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("leaf"); ap.add_argument("counts", nargs="+", type=int)
    ap.add_argument("--frames", type=int, default=12000); ap.add_argument("--block", type=int, default=512)
    ap.add_argument("--mem-cap", type=int, default=0); ap.add_argument("--ipw", type=int, default=0)
    ap.add_argument("--path", choices=["auto", "generic", "fast"], default="auto")
    a = ap.parse_args()
    if a.ipw:
        os.environ["ZAB_IPW"] = str(a.ipw)
    import zabatch
    meta = zabatch.leaf_meta(a.leaf); nch = int(meta["nch"])
    for n in a.counts:
        path = {"auto": zabatch.ZAB_PATH_AUTO, "generic": zabatch.ZAB_PATH_GENERIC, "fast": zabatch.ZAB_PATH_FAST}[a.path]
        with zabatch.Engine(a.leaf, n, max_block=a.block, mem_cap=a.mem_cap, path=path) as e:
            e.set_sliders(meta["default_sliders"]); e.prepare()
            nbytes = n * nch * a.frames * 4
            d_in, d_out = e.device_alloc(nbytes), e.device_alloc(nbytes)
            e.device_noise(d_in, a.frames)
            e.process_device(d_in, d_out, a.frames, block=a.block); e.sync()
            e.process_device(d_in, d_out, a.frames, block=a.block); e.sync()
            ms, _ = e.last_timing()
            ipw, lds_words = e.launch_shape()
            print(json.dumps({"leaf": a.leaf, "instances": n, "frames": a.frames, "ipw": ipw, "lds_mem_words": lds_words, "kernel": e.last_kernel_name(), "kernel_ms": round(ms, 3),
                              "instance_seconds_per_s": round(n * a.frames / 48000.0 / (ms * 1e-3), 1)}), flush=True)


if __name__ == "__main__":
    main()
