"""PsychoConvolver with a 0.5 s impulse response against the lanes per instance: python tools/psycho_ipw.py  (GPU box)"""
import os, subprocess, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
code = '''
import sys; sys.path[:0] = [r"%s", r"%s"]
import zabatch, numpy as np
from zajit import noise
ir = (noise.white_noise([321], 24000)[0].T * np.exp(-np.arange(24000) / 4000.0)[:, None]).reshape(-1).astype(np.float64)
n, frames = 256, 16384
with zabatch.Engine("PsychoConvolver", n, mem_cap=1 << 22) as e:
    e.file_slot_set(0, ir, channels=2, sample_rate=48000.0)
    e.set_sliders(zabatch.leaf_meta("PsychoConvolver")["default_sliders"]); e.prepare()
    nb = n * 2 * frames * 4
    di, do = e.device_alloc(nb), e.device_alloc(nb)
    e.device_noise(di, frames)
    for _ in range(2): e.process_device(di, do, frames); e.sync()
    print("ipw", e.launch_shape()[0], "ms", round(e.last_timing()[0], 1))
''' % (ROOT / "zorakaudio-experimental-plugins_amd", ROOT)
for ipw in ("1", "2", "4", "8", "16"):
    env = dict(os.environ, ZAB_IPW=ipw)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
    print(r.stdout.strip() or r.stderr[-300:], flush=True)
