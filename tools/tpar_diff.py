"""Where a leaf's time-parallel kernel first departs from its generic kernel: first frame, then variables and arena cells.
usage: python tools/tpar_diff.py LEAF [frames] [block] [mem_cap_log2]"""
import sys
sys.path[:0] = ['zorakaudio-experimental-plugins_amd', '.']
import numpy as np, zabatch
from zajit import noise

leaf = sys.argv[1]
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 1500
block = int(sys.argv[3]) if len(sys.argv) > 3 else 500
cap = 1 << (int(sys.argv[4]) if len(sys.argv) > 4 else 16)
meta = zabatch.leaf_meta(leaf)
x = noise.white_noise([3], frames, channels=int(meta["nch"]))
res = {}
for label, path in (("fast", zabatch.ZAB_PATH_FAST), ("generic", zabatch.ZAB_PATH_GENERIC)):
    with zabatch.Engine(leaf, 1, path=path, mem_cap=cap) as e:
        e.set_sliders(meta["default_sliders"]); e.prepare()
        y = e.process_host(x, block=block)
        res[label] = (y, e.read_vars()[0], e.read_mem(0, min(cap, max(1, int(e.mem_high()[0])))), e.var_names(), e.used_fast_path() if label == "fast" else None)
yf, yg = res["fast"][0][0], res["generic"][0][0]
print(leaf, "fast path used:", res["fast"][4])
d = np.abs(yf.astype(np.float64) - yg).max(axis=0)
bad = np.nonzero(d > 1e-6)[0]
print("frames that differ:", len(bad), "first:", bad[:20])
names = res["fast"][3]
vf, vg = res["fast"][1], res["generic"][1]
for k in np.nonzero(np.abs(vf - vg) > 1e-9)[0][:40]:
    print(f"  var {names[k]}: fast {vf[k]!r} generic {vg[k]!r}")
mf, mg = res["fast"][2], res["generic"][2]
n = min(len(mf), len(mg))
cells = np.nonzero(np.abs(mf[:n] - mg[:n]) > 1e-9)[0]
print("mem cells that differ:", len(cells), cells[:40], "high", len(mf), len(mg))
