"""Numeric model shared by the analysis (constant folding) and the numpy restatement: the value semantics of zart.h's
operators on arrays with one element per lane, the HOLD marker, the MT19937 stream of rand()."""
from __future__ import annotations

import math
import os
from typing import Dict, List, Optional, Tuple

import numpy as np

from .. import syntax as S
from ..emit import NOOP_CALLS, PURE_MATH1, PURE_MATH2, c_double
from ..program import Program, is_slider_name, is_spl_name

WAVE = 64
RNG_INDEX = "rand#index"      # hidden state: MT19937 outputs consumed since the start of the launch
HOLD_BITS = 0x7FF800005A5A5A5A   # "this frame left the variable alone": a quiet NaN no arithmetic produces (payload in the low
HOLD = np.array([HOLD_BITS], dtype=np.uint64).view(np.float64)[0]   # word, which a float -> double conversion leaves zero)
MT_N, MT_M = 624, 397

class MtStream:
    """MT19937 as za_mt_next (csrc/zart.h) runs it, in the form the kernels use: two generations side by side, the next one
    produced from the current one in three lane-parallel phases (element k of a new generation needs new[k - 227] from
    k = 227 on, so [0, 227), [227, 454) and [454, 623) are each parallel inside; element 623 closes the ring)."""

    def __init__(self, table=None, mti: int = 0):
        self.seeded_here = mti == 0
        if mti == 0:                       # first use: seed, position at the end -> the first word comes from the next generation
            t = np.zeros(MT_N, dtype=np.uint64)
            prev = 0x4141F00D
            t[0] = prev
            for k in range(1, MT_N):
                prev = (1812433253 * (prev ^ (prev >> 30)) + k) & 0xFFFFFFFF
                t[k] = prev
            self.cur, self.pos0 = t, MT_N
        else:
            self.cur, self.pos0 = np.asarray(table, dtype=np.uint64).copy(), int(mti)
        self.orig = (None if table is None else np.asarray(table).copy(), int(mti))
        self.nxt = self.twist(self.cur)

    @staticmethod
    def twist(cur):
        nxt = np.zeros(MT_N, dtype=np.uint64)

        def tw(a, b):
            y = (a & 0x80000000) | (b & 0x7FFFFFFF)
            return (y >> 1) ^ np.where(y & 1, 0x9908B0DF, 0).astype(np.uint64)

        k = np.arange(0, 227)
        nxt[k] = cur[k + MT_M] ^ tw(cur[k], cur[k + 1])
        k = np.arange(227, 454)
        nxt[k] = nxt[k - 227] ^ tw(cur[k], cur[k + 1])
        k = np.arange(454, 623)
        nxt[k] = nxt[k - 227] ^ tw(cur[k], cur[k + 1])
        nxt[623] = nxt[396] ^ tw(cur[623:624], nxt[0:1])[0]
        return nxt

    def word(self, idx):
        pos = self.pos0 + np.asarray(idx, dtype=np.int64)
        pos = np.clip(pos, 0, 2 * MT_N - 1)
        y = np.where(pos < MT_N, self.cur[np.minimum(pos, MT_N - 1)], self.nxt[np.maximum(pos - MT_N, 0)]).astype(np.uint64)
        y ^= y >> 11
        y ^= (y << 7) & 0x9D2C5680
        y ^= (y << 15) & 0xEFC60000
        y ^= y >> 18
        return (y & 0xFFFFFFFF).astype(np.float64)

    def end_chunk(self, total: int):
        """`total` words consumed so far: retire a generation once the last consumed word lies in the next one."""
        if self.pos0 + total > MT_N:
            self.cur = self.nxt
            self.nxt = self.twist(self.cur)
            self.pos0 -= MT_N

    def state(self, total: int):
        if total <= 0:
            return self.orig
        return self.cur.astype(np.uint32), self.pos0 + total


_MT_CTX: List[Optional[MtStream]] = [None]


def _truthy(a):
    return (a < 0.0) | (a > 0.0)


def _is_hold(a):
    return np.asarray(a, dtype=np.float64).view(np.uint64) == np.uint64(HOLD_BITS)


def _i32(a):
    """za_i32 of csrc/zart.h, element by element (tests only)."""
    a = np.asarray(a, dtype=np.float64)
    flat = a.reshape(-1)
    out = np.zeros(flat.shape, dtype=np.int64)
    for k, v in enumerate(flat):
        w = -(1 << 63) if not (-9.2233720368547758e18 < v < 9.2233720368547758e18) else int(v)
        w &= 0xFFFFFFFF
        out[k] = w - (1 << 32) if w >= (1 << 31) else w
    return out.reshape(a.shape)


def _np_op(op, a):
    if op == "+":
        return a[0] + a[1]
    if op == "-":
        return a[0] - a[1]
    if op == "*":
        return a[0] * a[1]
    if op == "/":
        return np.divide(a[0], a[1])
    if op == "neg":
        return 0.0 - a[0]
    if op == "not":
        return np.where(a[0] == 0.0, 1.0, 0.0)
    if op == "truth":
        return np.where(_truthy(a[0]), 1.0, 0.0)
    if op in ("<", "<=", ">", ">=", "=="):
        f = {"<": np.less, "<=": np.less_equal, ">": np.greater, ">=": np.greater_equal, "==": np.equal}[op]
        return np.where(f(a[0], a[1]), 1.0, 0.0)
    if op == "!=":
        return np.where((a[0] < a[1]) | (a[0] > a[1]), 1.0, 0.0)
    if op == "land":
        return np.where(_truthy(a[0]) & _truthy(a[1]), 1.0, 0.0)
    if op == "lor":
        return np.where(_truthy(a[0]) | _truthy(a[1]), 1.0, 0.0)
    if op == "sel":
        return np.where(_truthy(a[0]), a[1], a[2])
    if op in ("^", "pow"):
        return np.power(np.asarray(a[0], dtype=np.float64), a[1])
    if op in ("|", "&", "~", "<<", ">>", "%"):
        l, r = _i32(a[0]), _i32(a[1])
        if op == "|":
            v = l | r
        elif op == "&":
            v = l & r
        elif op == "~":
            v = l ^ r
        elif op == "<<":
            v = ((l & 0xFFFFFFFF) << (r & 31)) & 0xFFFFFFFF
            v = np.where(v >= 2 ** 31, v - 2 ** 32, v)
        elif op == ">>":
            v = l >> (r & 31)
        else:
            bad = (r == 0) | ((l == -2 ** 31) & (r == -1))
            rr = np.where(bad, 1, r)
            v = np.where(bad, 0, np.fmod(l, rr))           # C remainder: sign of the dividend
        return np.asarray(v, dtype=np.float64)
    if op == "min":
        return np.where(a[0] < a[1], a[0], a[1])
    if op == "max":
        return np.where(a[0] > a[1], a[0], a[1])
    if op == "sqr":
        return a[0] * a[0]
    if op == "sign":
        return np.where(a[0] > 0.0, 1.0, np.where(a[0] < 0.0, -1.0, 0.0))
    if op == "invsqrt":
        f = np.asarray(a[0], dtype=np.float64).astype(np.float32)
        bits = np.atleast_1d(f).view(np.int32)
        bits = (np.int32(0x5f3759df) - (bits >> 1)).astype(np.int32)
        y0 = bits.view(np.float32).astype(np.float64).reshape(np.shape(f))
        return y0 * (1.5 - (0.5 * a[0]) * (y0 * y0))
    if op == "atan2":
        return np.arctan2(a[0], a[1])
    if op == "mtout":
        return _MT_CTX[0].word(np.asarray(a[0]))
    if op == "addr":          # za_addr: trunc(base + index + 1e-5), negatives (and NaN) to 0
        x = np.asarray(a[0], dtype=np.float64) + a[1] + 1.0e-5
        return np.where(x > 0.0, np.trunc(np.where(x > 0.0, x, 0.0)), 0.0)
    if op in PURE_MATH1:
        f = {"sin": np.sin, "cos": np.cos, "sqrt": np.sqrt, "fabs": np.fabs, "floor": np.floor, "ceil": np.ceil, "asin": np.arcsin,
             "acos": np.arccos, "atan": np.arctan, "exp": np.exp, "log": np.log, "tan": np.tan, "log10": np.log10}[op]
        return f(np.asarray(a[0], dtype=np.float64))
    raise AssertionError(op)



__all__ = [_n for _n in dir() if not _n.startswith("__")]
