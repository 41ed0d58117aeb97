"""Per-sample arena access cost in a replica-lane leaf (tests/fixtures/ringio.jsfx): python tools/ring_io.py [instances]"""
import sys; from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT / "zorakaudio-experimental-plugins_amd"), str(ROOT)]
import zabatch, numpy as np
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
frames = 16384
for mask, nm in ((0, "counters only"), (1, "two stores"), (2, "two loads"), (4, "modulo counter"), (3, "stores + loads"), (7, "stores + loads + modulo")):
    with zabatch.Engine("fx_ringio", n) as e:
        row = np.zeros(64); row[0] = mask
        e.set_sliders(row); e.prepare()
        nb = n * 2 * frames * 4
        di, do = e.device_alloc(nb), e.device_alloc(nb)
        e.device_noise(di, frames)
        for _ in range(2): e.process_device(di, do, frames); e.sync()
        ms, _ = e.last_timing()
        print(f"{nm:26s} {ms:7.2f} ms  {ms / frames * 1e3:6.3f} us per frame  (ipw {e.launch_shape()[0]})", flush=True)
