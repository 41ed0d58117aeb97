"""PsychoConvolver with an impulse response loaded (0.5 s stereo: 12 partitions of 2048), one timed launch: python tools/psycho_ir.py [instances] [frames]
(under rocprofv3 --kernel-trace --stats this shows how the launch splits between the time-parallel kernel and its serial tail)"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT / "zorakaudio-experimental-plugins_amd"), str(ROOT)]
import numpy as np
import zabatch
from zajit import noise

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
ir = (noise.white_noise([321], 24000)[0].T * np.exp(-np.arange(24000) / 4000.0)[:, None]).reshape(-1).astype(np.float64)
with zabatch.Engine("PsychoConvolver", n, mem_cap=1 << 22) as e:
    e.file_slot_set(0, ir, channels=2, sample_rate=48000.0)
    e.set_sliders(zabatch.leaf_meta("PsychoConvolver")["default_sliders"]); e.prepare()
    nb = n * 2 * frames * 4
    di, do = e.device_alloc(nb), e.device_alloc(nb)
    e.device_noise(di, frames)
    for _ in range(2):
        e.process_device(di, do, frames); e.sync()
    ms, _ = e.last_timing()
    print(f"PsychoConvolver+IR N={n} frames={frames}: {ms:.2f} ms, kernel {e.last_kernel_name()}")
