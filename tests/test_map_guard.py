"""za_map_ok (csrc/zart.h): the run-time guard of the elementwise-loop lowering (zajit/emit.py _map_plan), compiled by g++ and
driven through a C entry point. It decides whether the trips of a loop are independent from one row per arena access:
(first address, stride per trip, kind: 0 load / 1 store / 2 load anywhere in [a0, a0 + ext)). A wrong "yes" here is a wrong
result on the device, so every accept / refuse rule is pinned, and a brute-force check over small random tables confirms that
whatever it accepts really has no cross-trip conflict."""
import ctypes as C
import itertools
import subprocess
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
CSRC = ROOT / "zorakaudio-experimental-plugins_amd" / "csrc"
SRC = '''
#include <stdint.h>
#include <math.h>
#define ZA_NV 1
#define ZA_NCH 2
#include "%s"
extern "C" int map_ok(const double* a0, const double* ext, const int* sig, const int* kind, int n, long long c) {
  ZaMapAcc A[24];
  for (int i = 0; i < n; ++i) { A[i].a0 = a0[i]; A[i].ext = ext[i]; A[i].sig = sig[i]; A[i].kind = kind[i]; A[i].ord = i; }
  return za_map_ok(A, n, c) ? 1 : 0;
}
extern "C" long long map_trips(double bound, double v, double step) { return za_map_trips(bound, v, step); }
''' % (CSRC / "zart.h")


@pytest.fixture(scope="module")
def lib(tmp_path_factory):
    d = tmp_path_factory.mktemp("mapguard")
    (d / "g.cpp").write_text(SRC)
    so = d / "libg.so"
    subprocess.run(["g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-w", "-o", str(so), str(d / "g.cpp"), "-lm"], check=True)
    L = C.CDLL(str(so))
    L.map_ok.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int, C.c_longlong]
    L.map_trips.restype = C.c_longlong
    L.map_trips.argtypes = [C.c_double, C.c_double, C.c_double]
    return L


def ok(L, rows, c):
    """rows: (a0, sig, kind[, ext])"""
    n = len(rows)
    a0 = (C.c_double * n)(*[float(r[0]) for r in rows])
    sg = (C.c_int * n)(*[int(r[1]) for r in rows])
    kd = (C.c_int * n)(*[int(r[2]) for r in rows])
    ex = (C.c_double * n)(*[float(r[3]) if len(r) > 3 else 0.0 for r in rows])
    return bool(L.map_ok(a0, ex, sg, kd, n, c))


def test_rules(lib):
    c = 100
    assert ok(lib, [(0, 1, 0), (1000, 1, 1)], c)                      # disjoint buffers
    assert ok(lib, [(50, 1, 0), (50, 1, 1)], c)                       # dest[i] op= dest[i]: one cell per trip, read then written
    assert not ok(lib, [(50, 1, 1), (50, 1, 0)], c)                   # written then read back inside the trip (stores are deferred)
    assert ok(lib, [(0, 2, 0), (0, 2, 1), (1, 2, 0), (1, 2, 1)], c)   # re / im interleave through idx = 2 k: buf[idx] *= g; buf[idx + 1] *= -g
    assert not ok(lib, [(0, 1, 0), (1, 1, 1)], c)                     # a[i + 1] = f(a[i]): each trip reads its neighbour's store
    assert not ok(lib, [(1, 1, 0), (0, 1, 1)], c)                     # a[i] = f(a[i + 1])
    assert ok(lib, [(0, 1, 0), (100, 1, 1)], c)                       # shift by the whole length: the ranges just miss
    assert not ok(lib, [(0, 1, 0), (99, 1, 1)], c)                    # ... and by one less they touch
    assert not ok(lib, [(0, 0, 1), (10, 1, 0)], c)                    # every trip stores to the same cell
    assert not ok(lib, [(0, 1, 1), (50, 2, 0)], c)                    # different strides over overlapping ranges
    assert ok(lib, [(0, 1, 1), (500, 2, 0)], c)                       # different strides, disjoint
    assert ok(lib, [(99, -1, 1), (200, 1, 0)], c)                     # downward store, disjoint
    assert not ok(lib, [(99, -1, 1), (0, 1, 0)], c)                   # downward store against an upward load of the same range
    assert ok(lib, [(0, 1, 1), (4096, 0, 2, 512)], c)                 # ring read elsewhere
    assert not ok(lib, [(0, 1, 1), (50, 0, 2, 512)], c)               # ring read that overlaps the stores
    assert not ok(lib, [(0.5, 1, 1)], c)                              # addresses must be whole
    assert not ok(lib, [(-5, 1, 1)], c)                               # and inside the arena's positive range
    assert not ok(lib, [(0, 1, 1)], 1)                                # one trip: nothing to share
    assert ok(lib, [(0, 1, 0), (0, 1, 0)], c) and ok(lib, [(0, 0, 0), (7, 1, 1)], c)   # loads may coincide; an invariant load elsewhere


def test_accepted_tables_have_no_cross_trip_conflict(lib):
    """Brute force over random small tables: if za_map_ok says yes, no store of one trip may hit an address another trip touches."""
    rng = np.random.default_rng(20261004)
    accepted = 0
    for _ in range(4000):
        n, c = int(rng.integers(1, 5)), int(rng.integers(2, 9))
        rows = []
        for _ in range(n):
            kind = int(rng.integers(0, 3))
            if kind == 2:
                rows.append((int(rng.integers(0, 24)), 0, 2, int(rng.integers(1, 8))))
            else:
                rows.append((int(rng.integers(0, 24)), int(rng.integers(-3, 4)), kind))
        if not ok(lib, rows, c):
            continue
        accepted += 1
        touched = []          # (trip, address, is_store)
        for k in range(c):
            for r in rows:
                if r[2] == 2:
                    touched += [(k, r[0] + j, False) for j in range(r[3])]
                else:
                    touched.append((k, r[0] + r[1] * k, r[2] == 1))
        for (k1, a1, s1), (k2, a2, s2) in itertools.combinations(touched, 2):
            assert not (k1 != k2 and a1 == a2 and (s1 or s2)), (rows, c, (k1, a1, s1), (k2, a2, s2))
            # inside one trip (list order = program order): no load of a cell after a store to it
            assert not (k1 == k2 and a1 == a2 and s1 and not s2), (rows, c, (k1, a1, s1), (k2, a2, s2))
    assert accepted > 200


def test_trip_count_of_the_while_form(lib):
    for bound, v, step, want in ((64, 0, 1, 64), (64, 60, 1, 4), (64, 64, 1, 0), (64, 70, 1, 0), (10.5, 0, 1, 11), (10, 0, 3, 4),
                                 (float("nan"), 0, 1, 0)):
        assert lib.map_trips(bound, v, step) == want, (bound, v, step)
