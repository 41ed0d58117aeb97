"""ctypes binding of include/zabatch.h + a host-side mirror of the reference's processor surface.

`JsfxBatchProcessor` keeps the call sequence of JSFXJuceProcessor (src/JSFXJuceProcessor.cpp:3239-3342 prepareToPlay,
:3435-3772 processBlock, :9286-9357 pushParamsToStateSliders) with the same names and argument meaning, except that
every call acts on N instances of the leaf living on one GPU. This module never computes DSP on the CPU: if the HIP
library or the leaf's module is missing it raises.
"""
from __future__ import annotations

import ctypes as C
import json
from pathlib import Path
from typing import Optional

import numpy as np

PKG = Path(__file__).resolve().parent
LIB_DIR = PKG / "lib"

ZAB_PATH_AUTO, ZAB_PATH_GENERIC, ZAB_PATH_FAST = 0, 1, 2
ZAB_BUF_DEVICE, ZAB_BUF_HOST = 0, 1


class ZabError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"zabatch error {code}: {msg}")
        self.code = code


class zab_config(C.Structure):
    _fields_ = [("n_instances", C.c_int32), ("device", C.c_int32), ("srate", C.c_double), ("max_block", C.c_int32),
                ("path", C.c_int32), ("mem_cap", C.c_int64), ("first_instance_id", C.c_uint64)]


class zab_host_state(C.Structure):      # include/zabatch.h
    _fields_ = [("struct_size", C.c_uint64),
                ("spl", C.POINTER(C.c_double)), ("sliders", C.POINTER(C.c_double)), ("vars", C.POINTER(C.c_double)),
                ("mem", C.POINTER(C.c_double)), ("mem_n", C.c_int64), ("pending_masks", C.POINTER(C.c_int64)),
                ("rand_mt", C.POINTER(C.c_uint32)), ("rand_index", C.POINTER(C.c_uint32)),
                ("slider_visible_mask", C.POINTER(C.c_int64)), ("slider_visibility_init", C.POINTER(C.c_int32)),
                ("mem_high", C.POINTER(C.c_int64)), ("flags", C.POINTER(C.c_uint32)), ("slider_changes", C.POINTER(C.c_uint64))]


class zab_wav_info(C.Structure):
    _fields_ = [("channels", C.c_int32), ("sample_rate", C.c_int32), ("bits", C.c_int32), ("is_float", C.c_int32), ("frames", C.c_int64)]


class zab_group_stats(C.Structure):
    _fields_ = [("max_kernel_ms", C.c_double), ("sum_kernel_ms", C.c_double), ("units", C.c_double), ("max_value", C.c_double),
                ("n_shards", C.c_int32), ("used_rccl", C.c_int32)]


class zab_pool_entry(C.Structure):
    _fields_ = [("offset_items", C.c_int64), ("frames", C.c_int32), ("sample_rate", C.c_int32), ("channels", C.c_int32),
                ("peak", C.c_float), ("rms", C.c_float)]


class zab_info(C.Structure):
    _fields_ = [("name", C.c_char * 64), ("nvars", C.c_int32), ("n_channels", C.c_int32), ("n_inputs", C.c_int32),
                ("n_outputs", C.c_int32), ("has_init", C.c_int32), ("has_slider", C.c_int32), ("has_block", C.c_int32),
                ("has_sample", C.c_int32), ("has_fast_path", C.c_int32), ("mem_cap", C.c_int64),
                ("n_instances", C.c_int32), ("layout_instance_major", C.c_int32)]


# every symbol include/zabatch.h declares (tests check the library exports all of them)
ABI_SYMBOLS = [
    "zab_last_error", "zab_abi_version", "zab_host_abi_version", "zab_create", "zab_destroy", "zab_get_info", "zab_var_count", "zab_var_name",
    "zab_var_index", "zab_set_sliders", "zab_get_sliders", "zab_consume_slider_changes", "zab_prepare", "zab_process", "zab_sync", "zab_read_vars",
    "zab_read_mem", "zab_write_mem", "zab_read_mem_high", "zab_device_alloc", "zab_device_free", "zab_device_upload",
    "zab_device_download", "zab_device_noise", "zab_last_timing", "zab_timing_history", "zab_stream",
    "zab_used_fast_path", "zab_last_kernel_name", "zab_launch_shape", "zab_handback_stats", "zab_host_alloc", "zab_host_free", "zab_state_upload", "zab_state_download", "zab_run_section", "zab_gmem_read", "zab_gmem_write", "zab_gmem_seq", "zab_pool_upload", "zab_file_slot_set",
    "zab_wav_read", "zab_wav_free", "zab_file_slot_load_wav", "zab_pool_upload_wav",
    "zab_group_create", "zab_group_destroy", "zab_group_size", "zab_group_shard", "zab_group_set_sliders", "zab_group_prepare",
    "zab_group_process", "zab_group_sync", "zab_group_reduce",
]

_lib = None


def wav_read(path):
    """(float32 array [frames, channels], sample_rate, bits, is_float) through the C library's own RIFF/WAVE reader."""
    L = load_runtime()
    wi, p = zab_wav_info(), C.POINTER(C.c_float)()
    rc = L.zab_wav_read(str(path).encode(), C.byref(wi), C.byref(p))
    if rc != 0:
        raise ZabError(rc, L.zab_last_error().decode())
    try:
        a = np.ctypeslib.as_array(p, shape=(max(1, wi.frames * wi.channels),))[:wi.frames * wi.channels].copy()
    finally:
        L.zab_wav_free(p)
    return a.reshape(wi.frames, wi.channels), int(wi.sample_rate), int(wi.bits), bool(wi.is_float)


def runtime_path() -> Path:
    return LIB_DIR / "libzabatch.so"


def load_runtime():
    """Load libzabatch.so; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    p = runtime_path()
    if not p.exists():
        raise ZabError(-2, f"{p} not built: run __graft_entry__.build() / python -m zajit.build")
    L = C.CDLL(str(p))
    vp, i32, i64, d = C.c_void_p, C.c_int32, C.c_int64, C.c_double
    L.zab_last_error.restype = C.c_char_p
    L.zab_create.argtypes = [C.c_char_p, C.POINTER(zab_config), C.POINTER(vp)]
    L.zab_destroy.argtypes = [vp]
    L.zab_get_info.argtypes = [vp, C.POINTER(zab_info)]
    L.zab_var_count.argtypes = [vp]
    L.zab_var_name.restype = C.c_char_p; L.zab_var_name.argtypes = [vp, C.c_int]
    L.zab_var_index.argtypes = [vp, C.c_char_p]
    L.zab_set_sliders.argtypes = [vp, i32, i32, C.POINTER(d)]
    L.zab_get_sliders.argtypes = [vp, i32, i32, C.POINTER(d)]
    L.zab_consume_slider_changes.argtypes = [vp, i32, i32, C.POINTER(C.c_uint64), C.POINTER(d)]
    L.zab_prepare.argtypes = [vp]
    L.zab_process.argtypes = [vp, vp, vp, i64, i64, i32, i32]
    L.zab_sync.argtypes = [vp]
    L.zab_read_vars.argtypes = [vp, i32, i32, C.POINTER(d)]
    L.zab_read_mem.argtypes = [vp, i32, i32, i64, i64, C.POINTER(d)]
    L.zab_write_mem.argtypes = [vp, i32, i32, i64, i64, C.POINTER(d)]
    L.zab_read_mem_high.argtypes = [vp, i32, i32, C.POINTER(i64)]
    L.zab_device_alloc.argtypes = [vp, i64, C.POINTER(vp)]
    L.zab_device_free.argtypes = [vp, vp]
    L.zab_device_upload.argtypes = [vp, vp, vp, i64]
    L.zab_device_download.argtypes = [vp, vp, vp, i64]
    L.zab_device_noise.argtypes = [vp, vp, i64, i64, C.c_uint64]
    L.zab_last_timing.argtypes = [vp, C.POINTER(d), C.POINTER(i32)]
    L.zab_timing_history.argtypes = [vp, C.POINTER(d), i32]
    L.zab_stream.restype = vp; L.zab_stream.argtypes = [vp]
    L.zab_used_fast_path.argtypes = [vp]
    L.zab_last_kernel_name.argtypes = [vp]
    L.zab_last_kernel_name.restype = C.c_char_p
    L.zab_launch_shape.argtypes = [vp, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    L.zab_handback_stats.argtypes = [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.zab_host_alloc.argtypes = [C.c_size_t, C.POINTER(C.c_void_p)]
    L.zab_host_free.argtypes = [vp]
    L.zab_state_upload.argtypes = [vp, i32, C.POINTER(zab_host_state)]
    L.zab_state_download.argtypes = [vp, i32, C.POINTER(zab_host_state)]
    L.zab_run_section.argtypes = [vp, i32, i32]
    L.zab_gmem_read.argtypes = [vp, i64, i64, C.POINTER(d)]
    L.zab_gmem_write.argtypes = [vp, i64, i64, C.POINTER(d)]
    L.zab_gmem_seq.argtypes = [vp, i64, C.POINTER(C.c_uint64)]
    L.zab_pool_upload.argtypes = [vp, i32, C.POINTER(zab_pool_entry), C.POINTER(C.c_float), i64]
    L.zab_file_slot_set.argtypes = [vp, i32, i32, C.c_double, C.POINTER(C.c_double), i64]
    L.zab_wav_read.argtypes = [C.c_char_p, C.POINTER(zab_wav_info), C.POINTER(C.POINTER(C.c_float))]
    L.zab_wav_free.argtypes = [C.POINTER(C.c_float)]
    L.zab_wav_free.restype = None
    L.zab_file_slot_load_wav.argtypes = [vp, i32, C.c_char_p, C.POINTER(zab_wav_info)]
    L.zab_pool_upload_wav.argtypes = [vp, i32, C.POINTER(C.c_char_p)]
    L.zab_group_create.argtypes = [C.c_char_p, C.POINTER(zab_config), C.POINTER(i32), i32, C.POINTER(vp)]
    L.zab_group_destroy.argtypes = [vp]
    L.zab_group_size.argtypes = [vp]
    L.zab_group_shard.argtypes = [vp, i32, C.POINTER(vp), C.POINTER(i32), C.POINTER(i32)]
    L.zab_group_set_sliders.argtypes = [vp, i32, i32, C.POINTER(d)]
    L.zab_group_prepare.argtypes = [vp]
    L.zab_group_process.argtypes = [vp, C.POINTER(vp), C.POINTER(vp), i64, i64, i32, i32]
    L.zab_group_sync.argtypes = [vp]
    L.zab_group_reduce.argtypes = [vp, C.POINTER(d), C.POINTER(zab_group_stats)]
    _lib = L
    return L


class PinnedArray:
    """float32 numpy array over page-locked host memory (zab_host_alloc); .array is valid until close()."""

    def __init__(self, shape):
        self.L = load_runtime()
        n = int(np.prod(shape)) * 4
        p = C.c_void_p()
        rc = self.L.zab_host_alloc(n, C.byref(p))
        if rc:
            raise ZabError(rc, self.L.zab_last_error().decode(errors="replace"))
        self.ptr = p
        self.array = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_float)), shape=(int(np.prod(shape)),)).reshape(shape)

    def close(self):
        if self.ptr:
            self.array = None
            self.L.zab_host_free(self.ptr)
            self.ptr = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def module_path(leaf: str) -> Path:
    return LIB_DIR / f"libzab_{leaf}.so"


def leaf_meta(leaf: str) -> dict:
    p = LIB_DIR / f"{leaf}.json"
    if not p.exists():
        raise ZabError(-2, f"{p} missing: leaf {leaf} has not been built")
    return json.loads(p.read_text())


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


class Engine:
    """Thin RAII wrapper over zab_engine*."""

    def __init__(self, leaf: str, n_instances: int, srate: float = 48000.0, max_block: int = 512, device: int = 0,
                 path: int = ZAB_PATH_AUTO, mem_cap: int = 0, first_instance_id: int = 1):
        self.L = load_runtime()
        mp = module_path(leaf)
        if not mp.exists():
            raise ZabError(-2, f"{mp} missing: leaf {leaf} has not been built (no CPU fallback exists)")
        cfg = zab_config(int(n_instances), int(device), float(srate), int(max_block), int(path), int(mem_cap),
                         int(first_instance_id))
        h = C.c_void_p()
        self.h = None
        self._chk(self.L.zab_create(str(mp).encode(), C.byref(cfg), C.byref(h)))
        self.h = h
        self.leaf = leaf
        self.n = int(n_instances)
        self.srate = float(srate)
        info = zab_info()
        self._chk(self.L.zab_get_info(self.h, C.byref(info)))
        self.info = info
        self.nch = info.n_channels
        self.nvars = info.nvars
        self.mem_cap = int(info.mem_cap)
        self._owned = []

    def _chk(self, rc):
        if rc != 0:
            raise ZabError(rc, self.L.zab_last_error().decode(errors="replace"))

    def close(self):
        if getattr(self, "h", None):
            for p in list(self._owned):
                self.L.zab_device_free(self.h, p)
            self._owned.clear()
            self.L.zab_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # -- metadata
    def var_names(self):
        return [self.L.zab_var_name(self.h, i).decode() for i in range(self.nvars)]

    def var_index(self, name: str) -> int:
        return int(self.L.zab_var_index(self.h, name.encode()))

    # -- sliders / lifecycle
    def set_sliders(self, values, first: int = 0, count: Optional[int] = None):
        v = np.ascontiguousarray(values, dtype=np.float64)
        if v.ndim == 1:
            row = np.zeros(64)
            row[: len(v)] = v
            self._chk(self.L.zab_set_sliders(self.h, 0, 0, _dp(row)))
            return
        cnt = v.shape[0] if count is None else count
        full = np.zeros((cnt, 64))
        full[:, : v.shape[1]] = v[:cnt]
        self._chk(self.L.zab_set_sliders(self.h, int(first), int(cnt), _dp(full)))

    def get_sliders(self, first=0, count=None):
        cnt = self.n - first if count is None else count
        out = np.zeros((cnt, 64))
        self._chk(self.L.zab_get_sliders(self.h, int(first), int(cnt), _dp(out)))
        return out

    def consume_slider_changes(self, first=0, count=None):
        """(masks [count] uint64, sliders [count, 64]): slider masks the scripts raised since the last call (cleared), and the
        current slider values -- consumeDspSliderChanges() of the reference host."""
        cnt = self.n - first if count is None else count
        masks = np.zeros(cnt, dtype=np.uint64)
        rows = np.zeros((cnt, 64))
        self._chk(self.L.zab_consume_slider_changes(self.h, int(first), int(cnt), masks.ctypes.data_as(C.POINTER(C.c_uint64)), _dp(rows)))
        return masks, rows

    def prepare(self):
        self._chk(self.L.zab_prepare(self.h))

    # -- audio
    def process_host(self, x: np.ndarray, block: int = 512, out: np.ndarray = None) -> np.ndarray:
        """x: float32 [N, nch, frames] in host memory; staged over PCIe (long buffers: chunked three-stream pipeline);
        returns the output array (`out` if given, e.g. a PinnedArray's .array)."""
        x = np.ascontiguousarray(x, dtype=np.float32)
        assert x.shape[0] == self.n and x.shape[1] == self.nch, (x.shape, self.n, self.nch)
        frames = x.shape[2]
        y = np.empty_like(x) if out is None else out
        assert y.shape == x.shape and y.dtype == np.float32 and y.flags.c_contiguous
        self._chk(self.L.zab_process(self.h, x.ctypes.data, y.ctypes.data, frames, frames, int(block), ZAB_BUF_HOST))
        return y

    def device_alloc(self, nbytes: int) -> int:
        p = C.c_void_p()
        self._chk(self.L.zab_device_alloc(self.h, int(nbytes), C.byref(p)))
        self._owned.append(p.value)
        return p.value

    def device_free(self, p: int):
        self._chk(self.L.zab_device_free(self.h, p))
        if p in self._owned:
            self._owned.remove(p)

    def upload(self, dptr: int, arr: np.ndarray):
        a = np.ascontiguousarray(arr)
        self._chk(self.L.zab_device_upload(self.h, dptr, a.ctypes.data, a.nbytes))

    def download(self, dptr: int, shape, dtype=np.float32) -> np.ndarray:
        out = np.empty(shape, dtype=dtype)
        self._chk(self.L.zab_device_download(self.h, out.ctypes.data, dptr, out.nbytes))
        return out

    def device_noise(self, dptr: int, frames: int, stride: Optional[int] = None, id_offset: int = 0):
        self._chk(self.L.zab_device_noise(self.h, dptr, int(frames), int(stride or frames), int(id_offset)))

    def process_device(self, d_in: int, d_out: int, frames: int, stride: Optional[int] = None, block: int = 512):
        self._chk(self.L.zab_process(self.h, d_in, d_out, int(frames), int(stride or frames), int(block), ZAB_BUF_DEVICE))

    def sync(self):
        self._chk(self.L.zab_sync(self.h))

    def last_timing(self):
        ms, n = C.c_double(), C.c_int32()
        self._chk(self.L.zab_last_timing(self.h, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def timing_history(self, max_entries: int = 64) -> np.ndarray:
        out = np.zeros(max_entries)
        n = self.L.zab_timing_history(self.h, _dp(out), int(max_entries))
        if n < 0:
            self._chk(n)
        return out[:n]

    def used_fast_path(self) -> bool:
        return bool(self.L.zab_used_fast_path(self.h))

    def last_kernel_name(self) -> str:
        return self.L.zab_last_kernel_name(self.h).decode()

    def handback(self):
        """(instance-launches the time-parallel kernel handed to the serial section code, frames run there) in the last process call."""
        n, f = C.c_uint64(0), C.c_uint64(0)
        self._chk(self.L.zab_handback_stats(self.h, C.byref(n), C.byref(f)))
        return n.value, f.value

    def launch_shape(self):
        """(instances per wavefront, mem[] words per instance held in LDS) of the lane-per-instance kernels."""
        ipw, k = C.c_int32(0), C.c_int32(0)
        self._chk(self.L.zab_launch_shape(self.h, C.byref(ipw), C.byref(k)))
        return ipw.value, k.value

    # -- state
    def read_vars(self, first=0, count=None) -> np.ndarray:
        cnt = self.n - first if count is None else count
        out = np.zeros((cnt, self.nvars))
        self._chk(self.L.zab_read_vars(self.h, int(first), int(cnt), _dp(out)))
        return out

    def read_mem(self, start: int, n: int, first=0, count=None) -> np.ndarray:
        cnt = self.n - first if count is None else count
        out = np.zeros((cnt, n))
        self._chk(self.L.zab_read_mem(self.h, int(first), int(cnt), int(start), int(n), _dp(out)))
        return out

    def write_mem(self, start: int, values: np.ndarray, first=0):
        v = np.ascontiguousarray(values, dtype=np.float64)
        if v.ndim == 1:
            v = v[None, :]
        self._chk(self.L.zab_write_mem(self.h, int(first), v.shape[0], int(start), v.shape[1], _dp(v)))

    def gmem_read(self, start: int, n: int) -> np.ndarray:
        out = np.zeros(n)
        self._chk(self.L.zab_gmem_read(self.h, int(start), int(n), _dp(out)))
        return out

    def gmem_write(self, start: int, values):
        v = np.ascontiguousarray(values, dtype=np.float64)
        self._chk(self.L.zab_gmem_write(self.h, int(start), len(v), _dp(v)))

    def gmem_seq(self, page: int = -1) -> int:
        out = C.c_uint64(0)
        self._chk(self.L.zab_gmem_seq(self.h, int(page), C.byref(out)))
        return int(out.value)

    def pool_upload(self, samples, sample_rates=None):
        """samples: list of float32 arrays [frames, channels] (or [frames]); packs them like a pool generation."""
        ents, chunks, off = [], [], 0
        for k, a in enumerate(samples):
            a = np.asarray(a, dtype=np.float32)
            if a.ndim == 1:
                a = a[:, None]
            sr = int(sample_rates[k]) if sample_rates is not None else int(self.srate)
            peak = float(np.abs(a).max()) if a.size else 0.0
            rms = float(np.sqrt(np.mean(a.astype(np.float64) ** 2))) if a.size else 0.0
            ents.append(zab_pool_entry(off, a.shape[0], sr, a.shape[1], peak, rms))
            chunks.append(np.ascontiguousarray(a).reshape(-1))
            off += a.size
        audio = np.concatenate(chunks) if chunks else np.zeros(0, np.float32)
        arr = (zab_pool_entry * max(1, len(ents)))(*ents)
        self._chk(self.L.zab_pool_upload(self.h, len(ents), arr, audio.ctypes.data_as(C.POINTER(C.c_float)), audio.size))

    def file_slot_load_wav(self, slot: int, path) -> dict:
        """Decode a RIFF/WAVE file into a file slot (zab_file_slot_load_wav: PCM 8/16/24/32, float 32/64, any channel count)."""
        wi = zab_wav_info()
        self._chk(self.L.zab_file_slot_load_wav(self.h, int(slot), str(path).encode(), C.byref(wi)))
        return {k: getattr(wi, k) for k, _ in zab_wav_info._fields_}

    def pool_upload_wav(self, paths):
        """One sample-pool generation from RIFF/WAVE files, sample ids 1.. in argument order (zab_pool_upload_wav)."""
        arr = (C.c_char_p * max(1, len(paths)))(*[str(p).encode() for p in paths])
        self._chk(self.L.zab_pool_upload_wav(self.h, len(paths), arr))

    def file_slot_set(self, slot: int, items, channels: int = 1, sample_rate: float = 48000.0):
        """Assign decoded file data (interleaved doubles) to a file slot; items=None unassigns it."""
        if items is None:
            self._chk(self.L.zab_file_slot_set(self.h, int(slot), 0, C.c_double(0.0), None, 0))
            return
        a = np.ascontiguousarray(items, dtype=np.float64)
        self._chk(self.L.zab_file_slot_set(self.h, int(slot), int(channels), C.c_double(float(sample_rate)), _dp(a), a.size))

    # -- checkpoint / resume (the state exchange format doubles as the engine's checkpoint) -------------------
    def _host_state(self, arrs):
        h = zab_host_state()
        h.struct_size = C.sizeof(zab_host_state)
        P = C.POINTER
        h.spl = arrs["spl"].ctypes.data_as(P(C.c_double)); h.sliders = arrs["sliders"].ctypes.data_as(P(C.c_double))
        h.vars = arrs["vars"].ctypes.data_as(P(C.c_double))
        if arrs["mem"].size:
            h.mem = arrs["mem"].ctypes.data_as(P(C.c_double))
        h.mem_n = arrs["mem"].size
        h.pending_masks = arrs["masks"].ctypes.data_as(P(C.c_int64))
        h.rand_mt = arrs["mt"].ctypes.data_as(P(C.c_uint32)); h.rand_index = arrs["mti"].ctypes.data_as(P(C.c_uint32))
        h.slider_visible_mask = arrs["vis"].ctypes.data_as(P(C.c_int64))
        h.slider_visibility_init = arrs["visi"].ctypes.data_as(P(C.c_int32))
        h.mem_high = arrs["high"].ctypes.data_as(P(C.c_int64))
        h.flags = arrs["flags"].ctypes.data_as(P(C.c_uint32))
        if "changes" in arrs:
            h.slider_changes = arrs["changes"].ctypes.data_as(P(C.c_uint64))
        return h

    def checkpoint(self) -> dict:
        """Everything zab_process depends on, per instance, as numpy arrays: spl / sliders / vars, mem[] up to the write
        high-water mark (sparse: untouched tails are not stored) and the mark itself, pending slider masks, the engine's
        per-instance flags (sliders changed -> @slider pending), MT19937 state, visibility. The shared gmem segment, sample
        pool and file slots belong to the host and are not part of it."""
        n, nv = self.n, max(1, self.nvars)
        high = self.mem_high()
        out = {"leaf": np.array(self.leaf), "srate": np.array(self.srate), "spl": np.zeros((n, 64)), "sliders": np.zeros((n, 64)),
               "vars": np.zeros((n, nv)), "masks": np.zeros((n, 3), np.int64), "mt": np.zeros((n, 624), np.uint32),
               "mti": np.zeros(n, np.uint32), "vis": np.zeros(n, np.int64), "visi": np.zeros(n, np.int32),
               "mem_high": high.astype(np.int64), "flags": np.zeros(n, np.uint32), "mem_cap": np.array(self.mem_cap, np.int64),
               "changes": np.zeros(n, np.uint64),      # (slider masks raised but not yet consumed by the host)
               # the script this state belongs to: its variable table's hash (a script's named constants are literals in the
               # kernels, zajit/program.py -- an image of another script text would carry cells the kernels do not read)
               "vars_sha1": np.array(str(leaf_meta(self.leaf).get("vars_sha1", "")))}
        mems = []
        for i in range(n):
            a = {k: out[k][i:i + 1].reshape(-1) if out[k].ndim > 1 else out[k][i:i + 1] for k in ("spl", "sliders", "vars", "masks", "mt", "mti", "vis", "visi", "flags", "changes")}
            a["high"] = out["mem_high"][i:i + 1]
            a["mem"] = np.zeros(int(min(high[i], self.mem_cap)))
            h = self._host_state(a)
            self._chk(self.L.zab_state_download(self.h, i, C.byref(h)))
            mems.append(a["mem"])
        out["mem_offsets"] = np.concatenate([[0], np.cumsum([m.size for m in mems])]).astype(np.int64)
        out["mem_data"] = np.concatenate(mems) if mems else np.zeros(0)
        return out

    def restore(self, ck: dict):
        """Load a checkpoint() into this engine (same leaf, instance count, sample rate and arena capacity), whatever the
        engine ran before: every instance's image is replaced -- arena cells above the stored prefix read as zeros again, the
        write high-water marks and pending-@slider flags are the checkpoint's -- and the next zab_process continues from it."""
        if str(ck["leaf"]) != self.leaf or ck["vars"].shape[0] != self.n:
            raise ZabError(-1, "checkpoint does not match this engine (leaf / instance count)")
        if "vars_sha1" in ck and str(ck["vars_sha1"]) != str(leaf_meta(self.leaf).get("vars_sha1", "")):
            raise ZabError(-1, "checkpoint was taken from another text of this script (variable tables differ)")
        if float(ck["srate"]) != self.srate:
            raise ZabError(-1, f"checkpoint was taken at srate {float(ck['srate'])}, this engine runs at {self.srate}")
        if "mem_cap" in ck and int(ck["mem_cap"]) != self.mem_cap:
            raise ZabError(-1, f"checkpoint was taken with mem_cap {int(ck['mem_cap'])}, this engine has {self.mem_cap}")
        off = ck["mem_offsets"]
        flags = ck["flags"] if "flags" in ck else np.zeros(self.n, np.uint32)
        for i in range(self.n):
            a = {k: np.ascontiguousarray(ck[k][i:i + 1].reshape(-1) if ck[k].ndim > 1 else ck[k][i:i + 1]) for k in ("spl", "sliders", "vars", "masks", "mt", "mti", "vis", "visi")}
            a["high"] = np.ascontiguousarray(ck["mem_high"][i:i + 1], dtype=np.int64)
            a["flags"] = np.ascontiguousarray(flags[i:i + 1], dtype=np.uint32)
            a["changes"] = np.ascontiguousarray(ck["changes"][i:i + 1] if "changes" in ck else np.zeros(1), dtype=np.uint64)
            a["mem"] = np.ascontiguousarray(ck["mem_data"][off[i]:off[i + 1]])
            h = self._host_state(a)
            self._chk(self.L.zab_state_upload(self.h, i, C.byref(h)))

    def mem_high(self, first=0, count=None) -> np.ndarray:
        cnt = self.n - first if count is None else count
        out = np.zeros(cnt, dtype=np.int64)
        self._chk(self.L.zab_read_mem_high(self.h, int(first), int(cnt), out.ctypes.data_as(C.POINTER(C.c_int64))))
        return out


class Group:
    """One job sharded by instance over several GPUs of a node (zab_group_*, include/zabatch.h): contiguous instance ranges,
    one engine + stream + host thread per shard, no collective on the data path, RCCL for the end-of-run statistics."""

    def __init__(self, leaf: str, n_instances: int, devices, srate: float = 48000.0, max_block: int = 512,
                 path: int = ZAB_PATH_AUTO, mem_cap: int = 0, first_instance_id: int = 1):
        self.L = load_runtime()
        mp = module_path(leaf)
        if not mp.exists():
            raise ZabError(-2, f"{mp} missing: leaf {leaf} has not been built (no CPU fallback exists)")
        cfg = zab_config(int(n_instances), 0, float(srate), int(max_block), int(path), int(mem_cap), int(first_instance_id))
        devs = (C.c_int32 * len(devices))(*[int(x) for x in devices])
        h = C.c_void_p()
        self.h = None
        self._chk(self.L.zab_group_create(str(mp).encode(), C.byref(cfg), devs, len(devices), C.byref(h)))
        self.h = h
        self.n = int(n_instances)
        self.shards = []                       # (first, count, Engine view) per shard
        for k in range(self.L.zab_group_size(self.h)):
            eh, first, count = C.c_void_p(), C.c_int32(), C.c_int32()
            self._chk(self.L.zab_group_shard(self.h, k, C.byref(eh), C.byref(first), C.byref(count)))
            self.shards.append((first.value, count.value, _EngineView(self.L, eh, leaf, count.value, srate)))
        self.nch = self.shards[0][2].nch

    def _chk(self, rc):
        if rc != 0:
            raise ZabError(rc, self.L.zab_last_error().decode(errors="replace"))

    def close(self):
        if getattr(self, "h", None):
            self.L.zab_group_destroy(self.h)
            self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def set_sliders(self, values, first: int = 0, count: Optional[int] = None):
        v = np.ascontiguousarray(values, dtype=np.float64)
        if v.ndim == 1:
            row = np.zeros(64)
            row[: len(v)] = v
            self._chk(self.L.zab_group_set_sliders(self.h, 0, 0, _dp(row)))
            return
        cnt = v.shape[0] if count is None else count
        full = np.zeros((cnt, 64))
        full[:, : v.shape[1]] = v[:cnt]
        self._chk(self.L.zab_group_set_sliders(self.h, int(first), int(cnt), _dp(full)))

    def prepare(self):
        self._chk(self.L.zab_group_prepare(self.h))

    def process_host(self, x: np.ndarray, block: int = 512) -> np.ndarray:
        """x: float32 [N, nch, frames] for the whole job; every shard stages its own slice (one host thread per shard)."""
        x = np.ascontiguousarray(x, dtype=np.float32)
        assert x.shape[0] == self.n and x.shape[1] == self.nch
        y = np.empty_like(x)
        frames = x.shape[2]
        ins = (C.c_void_p * len(self.shards))(*[x[f:f + c].ctypes.data for f, c, _ in self.shards])
        outs = (C.c_void_p * len(self.shards))(*[y[f:f + c].ctypes.data for f, c, _ in self.shards])
        self._chk(self.L.zab_group_process(self.h, ins, outs, frames, frames, int(block), ZAB_BUF_HOST))
        return y

    def process_device(self, d_in, d_out, frames: int, stride: Optional[int] = None, block: int = 512):
        ins = (C.c_void_p * len(self.shards))(*d_in)
        outs = (C.c_void_p * len(self.shards))(*d_out)
        self._chk(self.L.zab_group_process(self.h, ins, outs, int(frames), int(stride or frames), int(block), ZAB_BUF_DEVICE))

    def sync(self):
        self._chk(self.L.zab_group_sync(self.h))

    def reduce(self, shard_values=None) -> dict:
        st = zab_group_stats()
        v = None if shard_values is None else np.ascontiguousarray(shard_values, dtype=np.float64)
        self._chk(self.L.zab_group_reduce(self.h, None if v is None else _dp(v), C.byref(st)))
        return {k: getattr(st, k) for k, _ in zab_group_stats._fields_}


class _EngineView(Engine):
    """An engine owned by a Group: the Engine methods over a borrowed handle (close() leaves it to the group)."""

    def __init__(self, L, handle, leaf, n, srate):
        self.L, self.h, self.leaf, self.n, self.srate = L, handle, leaf, int(n), float(srate)
        info = zab_info()
        self._chk(self.L.zab_get_info(self.h, C.byref(info)))
        self.info, self.nch, self.nvars, self.mem_cap = info, info.n_channels, info.nvars, int(info.mem_cap)
        self._owned = []

    def close(self):
        for p in list(self._owned):
            self.L.zab_device_free(self.h, p)
        self._owned.clear()
        self.h = None


class JsfxBatchProcessor:
    """N instances of one JSFX leaf behind the reference processor's method names.

    reference                                   here
    ------------------------------------------  ---------------------------------------------------------------
    JSFXJuceProcessor()  (parses slider decls)  JsfxBatchProcessor(leaf, n_instances)
    host parameter values (APVTS)               set_parameter(slider_index0, host_value[, instance])
    prepareToPlay(sampleRate, blockSize)        prepareToPlay(sampleRate, samplesPerBlockExpected)
    processBlock(AudioBuffer&, MidiBuffer&)     processBlock(buffer[N, nch, n])  (in-place semantics: returns output)
    """

    def __init__(self, leaf: str, n_instances: int, device: int = 0, path: int = ZAB_PATH_AUTO, mem_cap: int = 0):
        from zajit.sliders import SliderDecl
        self.leaf, self.n, self.device, self.path, self.mem_cap = leaf, int(n_instances), device, path, mem_cap
        self.meta = leaf_meta(leaf)
        self.decls = {}
        for k, sd in self.meta["sliders"].items():
            self.decls[int(k)] = SliderDecl(index0=int(k), default=sd["default"], vmin=sd["min"], vmax=sd["max"],
                                            step=sd["step"], var_name=sd["var"], is_choice=sd["is_choice"],
                                            is_string=sd["is_string"], label=sd["label"])
        self.host_params = np.zeros((self.n, 64))
        for i, dcl in self.decls.items():
            if dcl.is_string:
                continue
            host = dcl.default
            if dcl.is_choice:
                host = (dcl.default - dcl.vmin) / dcl.step if dcl.step > 0 else 0.0
            self.host_params[:, i] = host
        # script-originated slider values the host has not caught up with yet (internalSliderShadow / internalSliderPendingMask)
        self.shadow = np.zeros((self.n, 64))
        self.shadow_pending = np.zeros(self.n, dtype=np.uint64)
        self.engine: Optional[Engine] = None

    def set_parameter(self, slider_index0: int, host_value, instance=None):
        if instance is None:
            self.host_params[:, slider_index0] = host_value
        else:
            self.host_params[instance, slider_index0] = host_value

    def _slider_rows(self) -> np.ndarray:
        rows = np.zeros((self.n, 64))
        for i, dcl in self.decls.items():
            if dcl.is_string:
                continue
            col = self.host_params[:, i]
            uniq = {}
            for j, hv in enumerate(col):
                if hv not in uniq:
                    uniq[hv] = dcl.to_slider_value(float(hv))
                rows[j, i] = uniq[hv]
            bit = np.uint64(1) << np.uint64(i)
            for j in np.flatnonzero(self.shadow_pending & bit):      # pushParamsToStateSliders (:9333-9340)
                if self._equivalent(dcl, rows[j, i], self.shadow[j, i]):
                    self.shadow_pending[j] &= ~bit
                else:
                    rows[j, i] = self.shadow[j, i]
        return rows

    @staticmethod
    def _equivalent(dcl, a: float, b: float) -> bool:
        """sliderValuesEquivalent (src/JSFXJuceProcessor.cpp:5599-5606)."""
        from zajit.sliders import _llround
        if dcl.is_choice:
            return _llround(a) == _llround(b)
        return abs(a - b) <= max(1.0e-6, dcl.step * 0.25 if dcl.step > 0 else 1.0e-6)

    def _consume_slider_changes(self):
        """consumeDspSliderChanges (:5665-5739): sliders a script changed and announced (sliderchange / slider_automate)
        become the host's parameter values (through the parameter's own range and step, as setValueNotifyingHost does), and
        the raw script value stays in st.sliders[] until the host value is equivalent to it."""
        masks, rows = self.engine.consume_slider_changes()
        for j in np.flatnonzero(masks):
            for i, dcl in self.decls.items():
                if dcl.is_string or not (int(masks[j]) >> i) & 1:
                    continue
                v = float(rows[j, i])
                if dcl.is_choice:
                    step = dcl.step if dcl.step > 0 else 1.0
                    from zajit.sliders import _llround
                    nchoice = max(1, len(dcl.choices))
                    host = float(min(max(_llround((v - dcl.vmin) / step), 0), nchoice - 1))
                else:
                    host = float(np.float32(dcl.to_slider_value(v)))       # rawForParam is a float
                if not self._equivalent(dcl, dcl.to_slider_value(self.host_params[j, i]), v):
                    self.host_params[j, i] = host
                self.shadow[j, i] = v
                self.shadow_pending[j] |= np.uint64(1) << np.uint64(i)

    def prepareToPlay(self, sampleRate: float, samplesPerBlockExpected: int):
        if self.engine is not None:
            self.engine.close()
        self.engine = Engine(self.leaf, self.n, srate=sampleRate, max_block=max(1, samplesPerBlockExpected),
                             device=self.device, path=self.path, mem_cap=self.mem_cap)
        self.block = max(1, samplesPerBlockExpected)
        self.engine.set_sliders(self._slider_rows())      # sliders are valid inside @init (:3297-3302)
        self.engine.prepare()

    def processBlock(self, buffer: np.ndarray) -> np.ndarray:
        if self.engine is None:
            raise ZabError(-7, "processBlock before prepareToPlay")
        self.engine.set_sliders(self._slider_rows())      # pushParamsToStateSliders(); changed rows re-run @slider
        n = buffer.shape[2]
        out = self.engine.process_host(buffer, block=max(1, min(self.block, n)) if n else self.block)
        if self.meta.get("features") and "sliderchange" in self.meta["features"]:
            self._consume_slider_changes()
        return out

    def releaseResources(self):
        if self.engine is not None:
            self.engine.close()
            self.engine = None
