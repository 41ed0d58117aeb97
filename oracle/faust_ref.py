"""ctypes front of oracle/faust_ref.c -- the CPU restatement of the Faust leaves' mydsp::compute(). TEST INFRASTRUCTURE ONLY.

    ref = FaustRef("ClickBeGoneSG", 48000); y = ref.compute(x[2, n], zones, block=512); ref.state()

PARITY UNPINNED (no Faust compiler in this image, no reference fixture): see the header of faust_ref.c.
"""
from __future__ import annotations

import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
OUT = HERE / "_port"
LEAVES = {"ClickBeGoneSG": 0, "ModTilt": 1, "GTS": 2, "VAR": 3, "RED": 4}


def lib_path() -> Path:
    return OUT / "libfaust_ref.so"


def build(force=False) -> Path:
    src, so = HERE / "faust_ref.c", lib_path()
    OUT.mkdir(exist_ok=True)
    if force or not so.exists() or so.stat().st_mtime < src.stat().st_mtime:
        subprocess.run(["gcc", "-O2", "-ffp-contract=off", "-fno-fast-math", "-shared", "-fPIC", "-o", str(so), str(src), "-lm"],
                       check=True)
    return so


_lib = None


def _load():
    global _lib
    if _lib is None:
        if not lib_path().exists():
            build()
        L = C.CDLL(str(lib_path()))
        L.fref_state_bytes.restype = C.c_int
        L.fref_state.restype = C.c_int
        _lib = L
    return _lib


class FaustRef:
    def __init__(self, leaf: str, srate: float):
        self.L = _load()
        self.leaf = LEAVES[leaf]
        self.nch = self.L.fref_channels(self.leaf)
        self.st = C.create_string_buffer(self.L.fref_state_bytes(self.leaf))
        self.L.fref_init(self.leaf, self.st, int(srate))          # prepareToPlay: dsp->init((int) sampleRate)

    def compute(self, x: np.ndarray, zones, block: int = 512) -> np.ndarray:
        """processBlock per `block` frames: zones pushed, then compute(count, in, out) in place."""
        x = np.ascontiguousarray(x, dtype=np.float32)
        y = np.zeros_like(x)
        z = np.ascontiguousarray(zones, dtype=np.float32)
        n = x.shape[1]
        PF = C.POINTER(C.c_float)
        for pos in range(0, n, block):
            cnt = min(block, n - pos)
            ins = (PF * self.nch)(*[x[c, pos:].ctypes.data_as(PF) for c in range(self.nch)])
            outs = (PF * self.nch)(*[y[c, pos:].ctypes.data_as(PF) for c in range(self.nch)])
            self.L.fref_compute(self.leaf, self.st, z.ctypes.data_as(PF), cnt, ins, outs)
        return y

    def state(self) -> np.ndarray:
        o = np.zeros(1024, dtype=np.float32)
        n = self.L.fref_state(self.leaf, self.st, o.ctypes.data_as(C.POINTER(C.c_float)))
        return o[:n].copy()
