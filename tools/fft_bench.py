"""FFT builtin throughput (BASELINE config C3 ii) and the STFT / FFT-using leaves: python tools/fft_bench.py  (GPU box)"""
import sys; from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT / "zorakaudio-experimental-plugins_amd"), str(ROOT)]
import zabatch, numpy as np
import os
CASES = [("fx_fftbench", 2048, 4096, 4), ("fx_fftbench", 2048, 2048, 4), ("fx_fftbench", 2048, 1024, 4), ("fx_fftbench", 256, 4096, 4),
         ("fx_fftbench+fused", 2048, 4096, 4), ("fx_fftbench+fused", 2048, 1024, 4),
         ("fx_fftbench_full", 2048, 4096, 4), ("fx_fftbench_full", 2048, 1024, 4), ("fx_fftbench_full", 256, 4096, 4)]
for leaf, n, size, K in CASES:
    fused = leaf.endswith("+fused")      # the four builtins as adjacent calls: fft;permute and ipermute;ifft run as one transform each
    label, leaf = leaf, leaf.split("+")[0]
    if not zabatch.module_path(leaf).exists():
        continue
    with zabatch.Engine(leaf, n, mem_cap=1<<17) as e:
        row = np.zeros(64); row[0]=size; row[1]=K; row[2]=31 if fused else 15
        e.set_sliders(row); e.prepare()
        nch = e.nch; frames = 64
        nb = n*nch*frames*4
        di, do = e.device_alloc(nb), e.device_alloc(nb)
        e.device_noise(di, frames)
        for _ in range(2): e.process_device(di, do, frames, block=64); e.sync()
        ms,_ = e.last_timing()
        flops = 2*5*size*np.log2(size)
        print(f"{label} size={size} buffers={n} K={K}: {ms:.2f} ms -> {ms/K*1e3:.1f} us per round trip of the batch, {n*K*flops/ms/1e6:.1f} GFLOP/s, {n*K*4*2*size*16/ms/1e6:.1f} GB/s (4 ops x r+w)", flush=True)
if os.environ.get("FFT_BENCH_ONLY_KERNELS"):
    sys.exit(0)
from zajit import noise
ir = (noise.white_noise([321], 24000)[0].T * np.exp(-np.arange(24000) / 4000.0)[:, None]).reshape(-1).astype(np.float64)   # 0.5 s stereo
for leaf, n, frames in (("fx_stft4k", 1024, 16384), ("fx_stft", 1024, 16384), ("DOT", 1024, 16384), ("PsychoConvolver", 1024, 16384),
                        ("PsychoConvolver+IR", 256, 16384)):
    loaded = leaf.endswith("+IR")
    leaf = leaf.split("+")[0]
    with zabatch.Engine(leaf, n, mem_cap=(1<<22) if leaf=="PsychoConvolver" else 0) as e:
        if loaded:
            e.file_slot_set(0, ir, channels=2, sample_rate=48000.0)      # a 0.5 s impulse response: 12 partitions of 2048
        e.set_sliders(zabatch.leaf_meta(leaf)["default_sliders"]); e.prepare()
        nb = n*2*frames*4
        di, do = e.device_alloc(nb), e.device_alloc(nb)
        e.device_noise(di, frames)
        for _ in range(2): e.process_device(di, do, frames); e.sync()
        ms,_ = e.last_timing()
        print(f"{leaf}{'+IR' if loaded else ''} N={n} frames={frames}: {ms:.1f} ms  {n*2*frames/ms/1e3:.1f} Msamples/s  rt x{frames/48000/(ms/1e3):.2f}", flush=True)
