"""TEST INFRASTRUCTURE ONLY -- plain-Python restatement of the reference's gmem segment, for parity tests.

Follows src/DspJsfxGmem.cpp (clampCellIndex :67-77, load/store :189-207, bulkGet/bulkPut/fill/zero/copy :209-309,
pageSeq :311-318, bumpPage :178-187) and the argument coercion of src/DspJsfxRuntime.cpp:95-104 (clampIntArg) /
:519-572 (the gmem façade). The reference's implementation needs its generated JSFXDSP.h and POSIX shm and cannot be
built here, and no reference test pins gmem results: **parity unpinned** -- this restatement is the checker.
"""
from __future__ import annotations

import math

import numpy as np

DEFAULT_CELLS = 1024 * 1024
PAGE_CELLS = 1024
INT_MIN, INT_MAX = -(2 ** 31), 2 ** 31 - 1
U64_MAX = 2 ** 64 - 1


def clamp_cell_index(idx: float) -> int:
    if not math.isfinite(idx) or idx <= 0.0:
        return 0
    t = math.floor(idx + 1.0e-5)
    if t <= 0.0:
        return 0
    if t >= float(U64_MAX):
        return U64_MAX
    return int(t)


def clamp_int_arg(v: float) -> int:
    if not math.isfinite(v):
        return 0
    if v <= float(INT_MIN):
        return INT_MIN
    if v >= float(INT_MAX):
        return INT_MAX
    return int(math.floor(abs(v) + 0.5)) * (1 if v >= 0 else -1)     # llround: half away from zero


class GmemRef:
    def __init__(self, cells: int = DEFAULT_CELLS):
        self.cells = np.zeros(max(cells, DEFAULT_CELLS))
        self.n = len(self.cells)
        self.page_seq = np.zeros((self.n + PAGE_CELLS - 1) // PAGE_CELLS, dtype=np.uint64)
        self.page_writer = np.zeros_like(self.page_seq)
        self.global_seq = 0

    def _bump(self, page: int, writer: int):
        if page < len(self.page_seq):
            self.page_writer[page] = writer
            self.page_seq[page] += 1
        self.global_seq += 1

    def load(self, idx: float) -> float:
        c = clamp_cell_index(idx)
        return float(self.cells[c]) if c < self.n else 0.0

    def store(self, idx: float, value: float, writer: int = 1) -> float:
        c = clamp_cell_index(idx)
        if c >= self.n:
            return 0.0
        self.cells[c] = value
        self._bump(c // PAGE_CELLS, writer)
        return value

    def _span_bumps(self, dst: int, n: int, writer: int):
        last = -1
        for i in range(n):
            pg = (dst + i) // PAGE_CELLS
            if pg != last:
                self._bump(pg, writer)
                last = pg

    def get(self, mem: np.ndarray, dst_base: float, src_idx: float, count: float) -> int:
        d, s, c = clamp_int_arg(dst_base), clamp_int_arg(src_idx), clamp_int_arg(count)
        if c <= 0 or d < 0 or s < 0 or s >= self.n:
            return 0
        n = min(c, self.n - s)
        if d + n > len(mem):
            return 0                     # the reference would grow mem; callers size `mem` like the device arena
        mem[d:d + n] = self.cells[s:s + n]
        return n

    def put(self, mem: np.ndarray, dst_idx: float, src_base: float, count: float, writer: int = 1) -> int:
        d, s, c = clamp_int_arg(dst_idx), clamp_int_arg(src_base), clamp_int_arg(count)
        if c <= 0 or d < 0 or s < 0:
            return 0
        if s + c > len(mem) or d >= self.n:
            return 0
        n = min(c, self.n - d)
        self.cells[d:d + n] = mem[s:s + n]
        self._span_bumps(d, n, writer)
        return n

    def fill(self, dst_idx: float, value: float, count: float, writer: int = 1) -> int:
        d, c = clamp_int_arg(dst_idx), clamp_int_arg(count)
        if c <= 0 or d < 0 or d >= self.n:
            return 0
        n = min(c, self.n - d)
        self.cells[d:d + n] = value
        self._span_bumps(d, n, writer)
        return n

    def zero(self, dst_idx: float, count: float, writer: int = 1) -> int:
        return self.fill(dst_idx, 0.0, count, writer)

    def copy(self, dst_idx: float, src_idx: float, count: float, writer: int = 1) -> int:
        d, s, c = clamp_int_arg(dst_idx), clamp_int_arg(src_idx), clamp_int_arg(count)
        if c <= 0 or d < 0 or s < 0 or d >= self.n or s >= self.n:
            return 0
        n = min(c, self.n - d, self.n - s)
        tmp = self.cells[s:s + n].copy()
        self.cells[d:d + n] = tmp
        self._span_bumps(d, n, writer)
        return n

    def seq(self, page: float) -> float:
        p = clamp_int_arg(page)
        if p < 0:
            return float(self.global_seq)
        return float(self.page_seq[p]) if p < len(self.page_seq) else 0.0

    def page(self, idx: float) -> float:
        return float(clamp_cell_index(idx) // PAGE_CELLS)

    def size(self) -> float:
        return float(self.n)
