"""In-kernel phase clock of a generated time-parallel kernel: python tools/tpar_stamps.py LEAF [instances] [frames] [mem_cap] [ir]
(build the leaf with ZA_TPAR_STAMPS=1 first, e.g. as <LEAF>_stamps: tools/build_stamps.py). Phases: 0 @block / @slider between
blocks, 1 block prologue (invariants, address passes, staging of per-trip cells), 2 lane-parallel nodes and scans of the chunks,
3 serial recurrences, 4 switched recurrences, 5 stores / carries / output, 6 the frames events and guards fall on (section code),
7 loop overhead of the block loop, 8.. uniform loops."""
import ctypes
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT / "zorakaudio-experimental-plugins_amd"), str(ROOT)]


def main():
    import zabatch
    leaf = sys.argv[1]
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
    frames = int(sys.argv[3]) if len(sys.argv) > 3 else 12000
    meta = zabatch.leaf_meta(leaf)
    nch = int(meta["nch"])
    with zabatch.Engine(leaf, n, max_block=512, mem_cap=int(sys.argv[4]) if len(sys.argv) > 4 else 0) as e:
        if len(sys.argv) > 5 and sys.argv[5] == "ir":       # an impulse response in file slot 0 (PsychoConvolver): 0.5 s of decaying stereo noise
            from zajit import noise
            ir = (noise.white_noise([321], 24000)[0].T * np.exp(-np.arange(24000) / 4000.0)[:, None]).reshape(-1).astype(np.float64)
            e.file_slot_set(0, ir, channels=2, sample_rate=48000.0)
        e.set_sliders(meta["default_sliders"]); e.prepare()
        nbytes = n * nch * frames * 4
        d_in, d_out = e.device_alloc(nbytes), e.device_alloc(nbytes)
        e.device_noise(d_in, frames)
        e.process_device(d_in, d_out, frames, block=512); e.sync()
        mod = ctypes.CDLL(str(zabatch.module_path(leaf)))
        buf = (ctypes.c_ulonglong * 64)()
        mod.zab_tpar_stamps(None, 1)
        e.process_device(d_in, d_out, frames, block=512); e.sync()
        ms, _ = e.last_timing()
        mod.zab_tpar_stamps(buf, 0)
    v = np.array(list(buf), dtype=np.float64)
    tot = v.sum()
    print(f"{leaf} x{n} x{frames}: {ms:.2f} ms, kernel {e.last_kernel_name() if False else ''}")
    names = {0: "@block/@slider", 1: "block prologue", 2: "nodes + scans", 3: "serial recurrences", 4: "switched recurrences",
             5: "stores/carries/output", 6: "event / guard frames", 7: "block loop"}
    for k in range(64):
        if v[k]:
            print(f"  phase {k:2d} {names.get(k, 'uniform loop %d' % (2 * (k - 8))):24s} {100 * v[k] / tot:6.2f} %   {v[k] / n / 100e6 * 1e3:9.3f} ms at 100 MHz")


if __name__ == "__main__":
    main()
