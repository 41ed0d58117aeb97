"""CPU restatement of the reference's runtime file handles (src/JSFXJuceProcessor.cpp:4893-5215). TEST INFRASTRUCTURE ONLY.

No reference test pins these (parity unpinned); the checker follows the C++ line by line: 1-based handles from a LIFO free
list, `(int64)(x + 1e-5)` coercions, -1 for an unassigned slot, cursor semantics of avail / var / mem / seek / rewind.
"""
from __future__ import annotations

import numpy as np


class FileRef:
    def __init__(self, n_slots: int = 16):
        self.slots = [None] * n_slots          # (items float64[n], channels, srate) or None
        self.handles = []                      # [slot, cursor] or None (closed)
        self.free = []

    def assign(self, slot, items, channels=1, srate=48000.0):
        self.slots[slot] = None if items is None else (np.asarray(items, dtype=np.float64), int(channels), float(srate))

    @staticmethod
    def _i(x):
        return int(np.trunc(x + 1.0e-5))

    def _h(self, handle):
        hid = self._i(handle)
        if hid <= 0 or hid > len(self.handles) or self.handles[hid - 1] is None:
            return None
        return self.handles[hid - 1]

    def _data(self, h):
        return self.slots[h[0]]

    def open(self, index):                                          # rt_file_open_common :4948-4990
        s = self._i(index)
        if s < 0 or s >= len(self.slots) or self.slots[s] is None:
            return -1.0
        if self.free:
            k = self.free.pop()
            self.handles[k] = [s, 0]
        else:
            k = len(self.handles)
            self.handles.append([s, 0])
        return float(k + 1)

    def close(self, handle):                                        # :5002-5014
        hid = self._i(handle)
        if self._h(handle) is None:
            return 0.0
        self.handles[hid - 1] = None
        self.free.append(hid - 1)
        return 0.0

    def rewind(self, handle):
        h = self._h(handle)
        if h is not None:
            h[1] = 0
        return 0.0

    def seek(self, handle, offset):                                 # :5029-5054
        h = self._h(handle)
        if h is None:
            return 0.0
        d = self._data(h)
        if d is None:
            h[1] = 0
            return 0.0
        off = min(max(self._i(offset), 0), len(d[0]))
        h[1] = off
        return float(off)

    def avail(self, handle):                                        # :5056-5072
        h = self._h(handle)
        if h is None or self._data(h) is None:
            return 0.0
        return float(max(0, len(self._data(h)[0]) - h[1]))

    def riff(self, handle, nch, sr):                                # :5087-5105 -> (ret, nch, sr)
        h = self._h(handle)
        if h is None:
            return 0.0, nch, sr
        d = self._data(h)
        if d is None or d[1] <= 0:
            return 0.0, 0.0, 0.0
        return 1.0, float(d[1]), d[2]

    def var(self, handle):                                          # :5107-5132 -> (ret, var)
        h = self._h(handle)
        if h is None:
            return 0.0, 0.0
        d = self._data(h)
        if d is None or h[1] >= len(d[0]):
            return 0.0, 0.0
        v = float(d[0][h[1]])
        h[1] += 1
        return v, v

    def mem(self, memarr, handle, dest, length):                    # :5134-5172
        h = self._h(handle)
        if h is None or self._data(h) is None:
            return 0.0
        dst, ln = max(self._i(dest), 0), self._i(length)
        if ln <= 0:
            return 0.0
        items = self._data(h)[0]
        n = min(ln, len(items) - h[1])
        if n <= 0:
            return 0.0
        memarr[dst:dst + n] = items[h[1]:h[1] + n]
        h[1] += n
        return float(n)
