"""TEST INFRASTRUCTURE ONLY -- plain-Python restatement of the reference's sample-pool read path.

Follows src/DspJsfxSamplePool.cpp: entryFor :311-319, read :377-399, readInterp :401-410, read2 :412-441; and the
llround coercions of the rt_sample_* wrappers, src/JSFXJuceProcessor.cpp:5330-5476 (export_mem :5438-5476).
The reference class needs JUCE and a header that is not in its tree, so it cannot be built here and no reference test
pins these results: **parity unpinned** -- this restatement is the checker. Index arithmetic is exact integer math.
"""
from __future__ import annotations

import math

import numpy as np


def llround(v: float) -> int:
    if not math.isfinite(v):
        return 0
    return int(math.floor(abs(v) + 0.5)) * (1 if v >= 0 else -1)


class PoolRef:
    def __init__(self, samples):
        self.entries, chunks, off = [], [], 0
        for a in samples:
            a = np.asarray(a, dtype=np.float32)
            if a.ndim == 1:
                a = a[:, None]
            self.entries.append((off, a.shape[0], a.shape[1]))
            chunks.append(a.reshape(-1))
            off += a.size
        self.audio = np.concatenate(chunks) if chunks else np.zeros(0, np.float32)

    def entry(self, sample_id: float):
        i = llround(sample_id)
        return self.entries[i - 1] if 1 <= i <= len(self.entries) else None

    def read(self, sample_id, channel, frame) -> float:
        e = self.entry(sample_id)
        if e is None or e[1] == 0 or e[2] == 0:
            return 0.0
        if not math.isfinite(frame):
            frame = 0.0
        f = llround(frame)
        if f < 0 or f >= e[1]:
            return 0.0
        ch = min(max(llround(channel), 0), e[2] - 1)
        idx = e[0] + f * e[2] + ch
        return float(self.audio[idx]) if idx < len(self.audio) else 0.0

    def read_interp(self, sample_id, channel, phase) -> float:
        if not math.isfinite(phase):
            phase = 0.0
        base = math.floor(phase)
        frac = phase - base
        x0, x1 = self.read(sample_id, channel, base), self.read(sample_id, channel, base + 1.0)
        return x0 + (x1 - x0) * frac

    def read2(self, sample_id, phase, interp):
        e = self.entry(sample_id)
        if e is None or e[1] == 0 or e[2] == 0 or not math.isfinite(phase) or phase < 0.0 or phase > float(e[1] - 1):
            return 0.0, 0.0, 0.0
        rd = self.read_interp if interp else self.read
        l = rd(sample_id, 0, phase)
        r = rd(sample_id, 1, phase) if e[2] >= 2 else l
        return 1.0, l, r

    def export(self, mem, sample_id, dst, src, count, stereo):
        d, s, c = llround(dst), llround(src), llround(count)
        stride = 2 if stereo else 1
        if d < 0 or s < 0 or c <= 0 or d + c * stride > len(mem):
            return 0.0
        for i in range(c):
            if stereo:
                _, l, r = self.read2(sample_id, float(s + i), False)
                mem[d + 2 * i], mem[d + 2 * i + 1] = l, r
            else:
                mem[d + i] = self.read(sample_id, 0, float(s + i))
        return float(c)
