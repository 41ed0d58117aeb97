"""Phases of one in-LDS transform (zart_fft.h, -DZA_FFT_STAMPS): build tests/fixtures/fftbench.jsfx as fx_fftbenchclk with
ZA_EXTRA_HIP_FLAGS=-DZA_FFT_STAMPS, then  python tools/probes/fft_phase_clock.py [buffers]  (GPU box). The kernel prints the
cycle counts of four transforms of workgroup 7; the harness' time for the launch is printed beside them."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path[:0] = [str(ROOT / "zorakaudio-experimental-plugins_amd"), str(ROOT)]
import numpy as np
import zabatch
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
for size in (1024, 256):
    with zabatch.Engine("fx_fftbenchclk", n, mem_cap=1 << 17) as e:
        row = np.zeros(64); row[0] = size; row[1] = 4; row[2] = 31
        e.set_sliders(row); e.prepare()
        frames = 64
        nb = n * e.nch * frames * 4
        di, do = e.device_alloc(nb), e.device_alloc(nb)
        e.device_noise(di, frames)
        e.process_device(di, do, frames, block=64); e.sync()
        ms, _ = e.last_timing()
        print(f"buffers={n} points={size}: {ms / 4 * 1e3:.1f} us per fused round trip, kernel {e.last_kernel_name()}", flush=True)
