"""Per-operation cost of the replica-lane builtins and loops (tests/fixtures/opbench.jsfx): python tools/op_bench.py [instances]"""
import sys; from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT / "zorakaudio-experimental-plugins_amd"), str(ROOT)]
import zabatch, numpy as np
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
names = ["none", "memcpy 2048", "memset 2048", "addbuf while 2048", "convolve_c 1024", "memcpy + fft_real 2048 + permute", "memcpy + ipermute + ifft_real 2048",
         "scale loop 2048", "overlap while 1024"]
reps = 64
base = None
for op, nm in enumerate(names):
    with zabatch.Engine("fx_opbench", n, mem_cap=1 << 15) as e:
        row = np.zeros(64); row[0] = op; row[1] = reps
        e.set_sliders(row); e.prepare()
        frames = 64
        nb = n * e.nch * frames * 4
        di, do = e.device_alloc(nb), e.device_alloc(nb)
        e.device_noise(di, frames)
        for _ in range(3): e.process_device(di, do, frames, block=64); e.sync()
        ms, _ = e.last_timing()
        ipw = e.launch_shape()[0]
    if op == 0: base = ms
    print(f"{nm:38s} {ms:8.3f} ms per block of {reps}  -> {(ms - base) / reps * 1e3:8.2f} us per operation   (instances {n}, {ipw} per wavefront)", flush=True)
