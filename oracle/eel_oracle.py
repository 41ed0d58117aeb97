"""TEST INFRASTRUCTURE ONLY -- ctypes driver for oracle/_ref/libeel_oracle.so.

The library is the reference's own WDL/EEL2 portable VM (built by oracle/Makefile from
/root/reference/src/WDL) behind this repo's JUCE-free host (oracle/eel_host.cpp), which replays the
reference's shadow-runtime sequence (src/JSFXCorrectnessCheck.h:732-750, src/JSFXJuceProcessor.cpp:3589-3657).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
LIB_PATH = _HERE / "_ref" / "libeel_oracle.so"

_lib = None


def available() -> bool:
    return LIB_PATH.exists()


def lib():
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            raise RuntimeError(f"{LIB_PATH} missing: run `make -C oracle ref` where /root/reference exists")
        L = C.CDLL(str(LIB_PATH))
        vp, d, i64, i32 = C.c_void_p, C.c_double, C.c_int64, C.c_int
        dp, fp = C.POINTER(C.c_double), C.POINTER(C.c_float)
        L.eelo_create.restype = vp; L.eelo_create.argtypes = [C.c_char_p]
        L.eelo_destroy.argtypes = [vp]
        L.eelo_error.restype = C.c_char_p; L.eelo_error.argtypes = [vp]
        L.eelo_bind_alias.argtypes = [vp, i32, C.c_char_p]
        L.eelo_set_sliders.argtypes = [vp, dp, i32]
        L.eelo_get_sliders.argtypes = [vp, dp, i32]
        L.eelo_prepare.argtypes = [vp, d, i64]
        L.eelo_run_slider.argtypes = [vp]
        L.eelo_run_block.argtypes = [vp]
        L.eelo_run_sample.argtypes = [vp]
        L.eelo_process.argtypes = [vp, fp, fp, i32, i64, i32, d]
        L.eelo_get_var.restype = i32; L.eelo_get_var.argtypes = [vp, C.c_char_p, dp]
        L.eelo_set_var.argtypes = [vp, C.c_char_p, d]
        L.eelo_get_spl.restype = d; L.eelo_get_spl.argtypes = [vp, i32]
        L.eelo_mem_read.restype = i64; L.eelo_mem_read.argtypes = [vp, i64, i64, dp]
        L.eelo_mem_write.restype = i64; L.eelo_mem_write.argtypes = [vp, i64, i64, dp]
        L.eelo_mem_high.restype = i64; L.eelo_mem_high.argtypes = [vp]
        L.eelo_set_write_trace.argtypes = [vp, i32]
        L.eelo_pending_masks.argtypes = [vp, C.POINTER(C.c_uint64)]
        L.eelo_wdl_fft.argtypes = [dp, i32, i32]
        L.eelo_wdl_real_fft.argtypes = [dp, i32, i32]
        L.eelo_wdl_fft_permute.restype = i32; L.eelo_wdl_fft_permute.argtypes = [i32, i32]
        _lib = L
    return _lib


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


class EelOracle:
    """One reference VM instance running one JSFX script."""

    def __init__(self, jsfx_text: str, slider_aliases: dict[int, str] | None = None):
        self.L = lib()
        self.h = self.L.eelo_create(jsfx_text.encode("utf-8", errors="replace"))
        err = self.L.eelo_error(self.h).decode()
        if err:
            self.close()
            raise RuntimeError(f"EEL2 oracle compile error: {err}")
        for idx0, name in (slider_aliases or {}).items():
            self.L.eelo_bind_alias(self.h, int(idx0), name.encode())

    def close(self):
        if getattr(self, "h", None):
            self.L.eelo_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_sliders(self, values):
        v = np.zeros(64, dtype=np.float64)
        vals = np.asarray(values, dtype=np.float64)
        v[: len(vals)] = vals
        self.L.eelo_set_sliders(self.h, _dp(v), 64)

    def sliders(self):
        v = np.zeros(64, dtype=np.float64)
        self.L.eelo_get_sliders(self.h, _dp(v), 64)
        return v

    def prepare(self, srate: float, mem_hint: int = 0):
        self.srate = float(srate)
        self.L.eelo_prepare(self.h, float(srate), int(mem_hint))

    def run_slider(self):
        self.L.eelo_run_slider(self.h)

    def process(self, x: np.ndarray, block: int = 512) -> np.ndarray:
        """x: float32 [nCh, frames] planar. Returns same-shape float32 output."""
        x = np.ascontiguousarray(x, dtype=np.float32)
        nch, frames = x.shape
        y = np.zeros_like(x)
        self.L.eelo_process(self.h, _fp(x), _fp(y), nch, frames, int(block), self.srate)
        return y

    def var(self, name: str):
        out = C.c_double(0.0)
        ok = self.L.eelo_get_var(self.h, name.encode(), C.byref(out))
        return out.value if ok else None

    def set_var(self, name: str, v: float):
        self.L.eelo_set_var(self.h, name.encode(), float(v))

    def spl(self, ch: int) -> float:
        return self.L.eelo_get_spl(self.h, ch)

    def mem(self, start: int, count: int) -> np.ndarray:
        out = np.zeros(count, dtype=np.float64)
        self.L.eelo_mem_read(self.h, int(start), int(count), _dp(out))
        return out

    def mem_write(self, start: int, values):
        v = np.ascontiguousarray(values, dtype=np.float64)
        self.L.eelo_mem_write(self.h, int(start), len(v), _dp(v))

    @property
    def mem_high(self) -> int:
        return int(self.L.eelo_mem_high(self.h))

    def set_write_trace(self, on: bool):
        """Shadow-runtime store instrumentation (mem_high tracking); off for timing runs."""
        self.L.eelo_set_write_trace(self.h, 1 if on else 0)

    def pending_masks(self):
        m = (C.c_uint64 * 3)()
        self.L.eelo_pending_masks(self.h, m)
        return tuple(int(v) for v in m)


def wdl_fft(buf: np.ndarray, n: int, inverse: bool = False) -> np.ndarray:
    """In-place reference WDL_fft on interleaved complex f64 (returns the array)."""
    b = np.ascontiguousarray(buf, dtype=np.float64).copy()
    lib().eelo_wdl_fft(_dp(b), int(n), 1 if inverse else 0)
    return b


def wdl_real_fft(buf: np.ndarray, n: int, inverse: bool = False) -> np.ndarray:
    b = np.ascontiguousarray(buf, dtype=np.float64).copy()
    lib().eelo_wdl_real_fft(_dp(b), int(n), 1 if inverse else 0)
    return b


def wdl_fft_permute(n: int) -> np.ndarray:
    L = lib()
    return np.array([L.eelo_wdl_fft_permute(int(n), i) for i in range(n)], dtype=np.int64)
