"""One case of the FFT harness (tools/fft_bench.py), for profiler passes: python tools/fft_one.py [buffers] [points] [K] [fused 0|1]"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT / "zorakaudio-experimental-plugins_amd"), str(ROOT)]
import numpy as np
import zabatch

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
size = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
K = int(sys.argv[3]) if len(sys.argv) > 3 else 4
fused = bool(int(sys.argv[4])) if len(sys.argv) > 4 else True
with zabatch.Engine("fx_fftbench", n, mem_cap=1 << 17) as e:
    row = np.zeros(64); row[0] = size; row[1] = K; row[2] = 31 if fused else 15
    e.set_sliders(row); e.prepare()
    frames = 64
    nb = n * e.nch * frames * 4
    di, do = e.device_alloc(nb), e.device_alloc(nb)
    e.device_noise(di, frames)
    for _ in range(3):
        e.process_device(di, do, frames, block=64); e.sync()
    ms, _ = e.last_timing()
    print(f"fx_fftbench buffers={n} points={size} K={K} fused={fused}: {ms:.3f} ms = {ms / K * 1e3:.1f} us per round trip, kernel {e.last_kernel_name()}")
