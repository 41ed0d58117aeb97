"""PCIe-inclusive rate of the host-buffer path (ZAB_BUF_HOST): DDT x N instances, host numpy buffers in and out.

    python tools/pcie_rate.py [--instances 1024] [--frames 96000]

Never the bench `value` (that one has its inputs resident in HBM); DESIGN.md section 5 quotes this beside it.
"""
import argparse, json, sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT / "zorakaudio-experimental-plugins_amd"), str(ROOT)]


def main():
    import zabatch
    ap = argparse.ArgumentParser()
    ap.add_argument("--instances", type=int, default=1024); ap.add_argument("--frames", type=int, default=96000)
    a = ap.parse_args()
    meta = zabatch.leaf_meta("DDT")
    x = (np.random.default_rng(0).random((a.instances, 2, a.frames), dtype=np.float32) - 0.5)
    with zabatch.Engine("DDT", a.instances) as e:
        e.set_sliders(meta["default_sliders"]); e.prepare()
        n = a.instances * 2 * a.frames
        y = np.empty_like(x)
        with zabatch.PinnedArray(x.shape) as pi, zabatch.PinnedArray(x.shape) as po:
            pi.array[...] = x
            for label, xi, yo in (("pageable", x, y), ("pinned", pi.array, po.array)):
                e.process_host(xi, block=512, out=yo)                       # warm-up (staging buffers, page faults)
                t = time.perf_counter(); e.process_host(xi, block=512, out=yo); dt = time.perf_counter() - t
                ms, launches = e.last_timing()
                print(json.dumps({"host_memory": label, "instances": a.instances, "frames": a.frames, "wall_ms": round(dt * 1e3, 1),
                                  "launches": launches, "pcie_inclusive_msamples_per_s": round(n / dt / 1e6, 1),
                                  "host_gb_per_s_each_way": round(x.nbytes / dt / 1e9, 2)}), flush=True)


if __name__ == "__main__":
    main()
