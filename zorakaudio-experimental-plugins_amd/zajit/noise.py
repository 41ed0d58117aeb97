"""Synthetic white-noise input shared by tests, bench.py and the device generator (csrc/zab_noise.hip).

SURVEY §8(d): xorshift64 (x^=x<<13; x^=x>>7; x^=x<<17), value ((x>>11)*2^-53*2-1)*0.5 cast to f32,
L then R per frame, seed 0x9E3779B97F4A7C15 ^ (instance_id * 0xD1B54A32D192ED03).
"""
from __future__ import annotations

import numpy as np

SEED0 = 0x9E3779B97F4A7C15
SEED_MUL = 0xD1B54A32D192ED03
_M64 = (1 << 64) - 1


def instance_seed(instance_id: int) -> int:
    s = (SEED0 ^ ((instance_id * SEED_MUL) & _M64)) & _M64
    return s if s != 0 else SEED0


def white_noise(instances, frames: int, channels: int = 2) -> np.ndarray:
    """float32 [len(instances), channels, frames]; vectorised over instances, serial in time."""
    ids = np.atleast_1d(np.asarray(instances, dtype=np.uint64))
    with np.errstate(over="ignore"):
        x = np.array([instance_seed(int(i)) for i in ids], dtype=np.uint64)
        out = np.empty((len(ids), channels, frames), dtype=np.float32)
        for t in range(frames):
            for c in range(channels):
                x ^= x << np.uint64(13)
                x ^= x >> np.uint64(7)
                x ^= x << np.uint64(17)
                u = (x >> np.uint64(11)).astype(np.float64) * (2.0 ** -53)
                out[:, c, t] = ((u * 2.0 - 1.0) * 0.5).astype(np.float32)
    return out
