"""Where an STFT hop's time goes (tests/fixtures/stftparts.jsfx): python tools/stft_parts.py [instances]"""
import sys; from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT / "zorakaudio-experimental-plugins_amd"), str(ROOT)]
import zabatch, numpy as np
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
frames = 16384
base = None
for mask, nm in ((0, "per-sample ring i/o only"), (1, "+ window loop"), (2, "+ fft;permute (fused)"), (4, "+ gain loop"), (8, "+ ipermute;ifft (fused)"),
                 (16, "+ overlap-add loop"), (32, "+ memcpy x2, memset"), (63, "all")):
    with zabatch.Engine("fx_stftparts", n) as e:
        row = np.array(zabatch.leaf_meta("fx_stftparts")["default_sliders"], dtype=np.float64); row[1] = mask
        e.set_sliders(row); e.prepare()
        nb = n * 2 * frames * 4
        di, do = e.device_alloc(nb), e.device_alloc(nb)
        e.device_noise(di, frames)
        for _ in range(2): e.process_device(di, do, frames); e.sync()
        ms, _ = e.last_timing()
    if mask == 0: base = ms
    hops = frames // 256
    print(f"{nm:28s} {ms:7.2f} ms   {(ms - base) / hops / 2 * 1e3:7.1f} us per channel and hop (both instances of a wavefront)", flush=True)
