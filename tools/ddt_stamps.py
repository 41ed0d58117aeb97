"""Phase attribution of zab_ddt_fast: python tools/ddt_stamps.py N FRAMES   (module built with ZA_EXTRA_HIP_FLAGS=-DDDT_STAMPS)

Prints the share of per-wave cycles between the in-kernel stamps (s_memtime), summed over all waves of one launch."""
import ctypes as C
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "zorakaudio-experimental-plugins_amd"))
import numpy as np
import zabatch

n, frames = int(sys.argv[1]), int(sys.argv[2])
meta = zabatch.leaf_meta("DDT")
with zabatch.Engine("DDT", n) as e:
    e.set_sliders(meta["default_sliders"]); e.prepare()
    nbytes = n * 2 * frames * 4
    d_in, d_out = e.device_alloc(nbytes), e.device_alloc(nbytes)
    e.device_noise(d_in, frames)
    L = C.CDLL(str(zabatch.module_path("DDT")))
    out = (C.c_ulonglong * 16)()
    e.process_device(d_in, d_out, frames); e.sync()
    L.zab_ddt_stamps(out, 1)
    e.process_device(d_in, d_out, frames); e.sync()
    ms, _ = e.last_timing()
    assert L.zab_ddt_stamps(out, 1) == 0
import os
if os.environ.get("ZAB_DDT_KERNEL") == "wide":
    names = ["A: audio->ring", "barrier 1", "taps", "transposes", "direct+poles+scans", "barrier 2", "carry+C: fixup/mix/store"]
else:
    names = ["load + local + scans", "barrier 1 + carry chain", "fix-up + ring / mix-part writes", "barrier 2", "issue of the next chunk's loads", "taps (incl. K, a^n K)", "mix + store (+ meters)"]
v = np.array([out[i] for i in range(7)], dtype=np.float64)
print(f"N={n} frames={frames} kernel {ms:.3f} ms, waves {out[15]}")
for k, x in zip(names, v):
    print(f"  {k:28s} {100 * x / v.sum():5.1f} %   {x / out[15] / ((frames + 255) // 256):8.1f} ticks per wave-chunk")
