"""gmem[] / gmem_* builtins against oracle/gmem_ref.py, the restatement of src/DspJsfxGmem.cpp.
Indexing must be bit-exact: every case compares integer cell positions, return values and sequence counters exactly.
No reference test pins gmem (parity unpinned); the restatement is the checker on both CPU and GPU."""
import numpy as np
import pytest

OPS = {"store": 1, "load": 2, "get": 3, "put": 4, "fill": 5, "zero": 6, "copy": 7, "seq": 8, "page": 9, "size": 10,
       "addassign": 11}
N = 1024 * 1024
# (op, a, b, c): index clamps (floor(idx+1e-5), <=0 / NaN / inf -> 0), out-of-range, llround coercion, page edges
CASES = [
    ("store", 5.0, 1.25, 0), ("store", 5.99998, 2.5, 0), ("store", 5.99999, 3.5, 0), ("store", -3.0, 4.5, 0),
    ("store", float("nan"), 5.5, 0), ("store", float("inf"), 6.5, 0), ("store", N - 1, 7.5, 0), ("store", N, 8.5, 0),
    ("store", 1023.999995, 9.5, 0), ("store", 1e18, 1.0, 0),
    ("load", 7.0, 0, 0), ("load", 1024.0, 0, 0), ("load", N + 5, 0, 0), ("load", -1.0, 0, 0), ("load", 0.99999, 0, 0),
    ("fill", 1000.0, 0.75, 100.0), ("fill", N - 10, 0.5, 100.0), ("fill", -1.0, 1.0, 10.0), ("fill", 10.4, 2.0, 2.5),
    ("fill", 10.5, 3.0, 3.5), ("fill", N, 1.0, 4.0), ("fill", 20.0, 1.0, -4.0),
    ("zero", 1010.0, 30.0, 0),
    ("copy", 2000.0, 1000.0, 100.0), ("copy", 1050.0, 1000.0, 100.0), ("copy", 1000.0, 1050.0, 100.0),
    ("copy", N - 5, 1000.0, 50.0), ("copy", 0.0, N + 1, 5.0),
    ("put", 3000.0, 100.0, 50.0), ("put", 3000.0, 65530.0, 50.0), ("put", N - 3, 100.0, 50.0), ("put", -2.0, 100.0, 5.0),
    ("get", 200.0, 3000.0, 60.0), ("get", 65476.0, 3000.0, 60.0), ("get", 300.0, N - 4, 60.0), ("get", 300.0, -1.0, 6.0),
    ("seq", -1.0, 0, 0), ("seq", 0.0, 0, 0), ("seq", 1.0, 0, 0), ("seq", 2.6, 0, 0), ("seq", 5000.0, 0, 0),
    ("page", 1023.99998, 0, 0), ("page", 1023.999995, 0, 0), ("page", 4096.0, 0, 0), ("page", -7.0, 0, 0), ("size", 0, 0, 0),
    ("addassign", 5.0, 0.125, 0), ("addassign", N + 9, 1.0, 0),
]
MEM_CAP = 65536


def _ref_apply(g, mem, op, a, b, c):
    if op == "store":
        return g.store(a, b)
    if op == "load":
        return g.load(a)
    if op == "get":
        return g.get(mem, a, b, c)
    if op == "put":
        return g.put(mem, a, b, c)
    if op == "fill":
        return g.fill(a, b, c)
    if op == "zero":
        return g.zero(a, b)
    if op == "copy":
        return g.copy(a, b, c)
    if op == "seq":
        return g.seq(a)
    if op == "page":
        return g.page(a)
    if op == "size":
        return g.size()
    if op == "addassign":
        g.store(a, g.load(a) + b)
        return g.load(a)
    raise AssertionError(op)


def _seed():
    rng = np.random.default_rng(11)
    return rng.standard_normal(4096), rng.standard_normal(MEM_CAP)


def test_port_gmem_semantics():
    from oracle import port
    if not port.port_path("fx_gmemkat").exists():
        pytest.skip("fixture port not built")

    def dut(opc, a, b, c, g0, m0):
        p = port.Port("fx_gmemkat", 48000.0, mem_cap=MEM_CAP)
        p.set_sliders([0, 0, 0, 0]); p.prepare()
        p.gmem_write(0, g0); p.mem_write(0, m0)
        p.set_sliders([opc, a, b, c])
        p.process(np.zeros((1, 8), np.float32), 8)
        assert p.var("attached") == 1.0
        return (p.var("ret"), np.concatenate([p.gmem_read(0, 8192), p.gmem_read(1024 * 1024 - 64, 64)]), p.mem(0, MEM_CAP),
                p.gmem_seq(-1), np.array([p.gmem_seq(k) for k in range(8)] + [p.gmem_seq(1023)]))

    _run_cases_windowed(dut)


def _run_cases_windowed(make_dut):
    from oracle import gmem_ref
    g0, m0 = _seed()
    for op, a, b, c in CASES:
        ref = gmem_ref.GmemRef()
        ref.cells[:4096] = g0
        mem = m0.copy()
        want = _ref_apply(ref, mem, op, a, b, c)
        got, cells, dmem, gseq, pseq = make_dut(OPS[op], a, b, c, g0, m0)
        tag = (op, a, b, c)
        assert got == want, (tag, got, want)
        assert np.array_equal(cells, np.concatenate([ref.cells[:8192], ref.cells[-64:]])), tag
        assert np.array_equal(dmem, mem), tag
        assert gseq == ref.global_seq, tag
        assert np.array_equal(pseq, np.concatenate([ref.page_seq[:8], ref.page_seq[1023:1024]]).astype(np.int64)), tag


@pytest.mark.gpu
def test_gpu_gmem_semantics():
    import zabatch

    def dut(opc, a, b, c, g0, m0):
        with zabatch.Engine("fx_gmemkat", 1, mem_cap=MEM_CAP) as e:
            e.set_sliders([0, 0, 0, 0]); e.prepare()
            e.gmem_write(0, g0); e.write_mem(0, m0)
            e.set_sliders([opc, a, b, c])
            e.process_host(np.zeros((1, 1, 8), np.float32), block=8)
            v = e.read_vars()[0]; names = e.var_names()
            assert v[names.index("attached")] == 1.0
            return (v[names.index("ret")], np.concatenate([e.gmem_read(0, 8192), e.gmem_read(1024 * 1024 - 64, 64)]),
                    e.read_mem(0, MEM_CAP)[0], e.gmem_seq(-1), np.array([e.gmem_seq(k) for k in range(8)] + [e.gmem_seq(1023)]))

    _run_cases_windowed(dut)


@pytest.mark.gpu
def test_gpu_gmem_shared_between_instances():
    """200 instances of one engine write their own cell of the shared segment in the same launch (op 12)."""
    import zabatch
    n = 200
    with zabatch.Engine("fx_gmemkat", n, first_instance_id=10) as e:
        e.set_sliders([0, 0, 0, 0]); e.prepare()
        e.set_sliders([12, 0, 0, 0])
        e.process_host(np.zeros((n, 1, 8), np.float32), block=8)
        cells = e.gmem_read(0, 512)
        ids = np.arange(10, 10 + n)
        assert np.array_equal(cells[ids], ids * 0.5)
        assert np.count_nonzero(cells) == n
        assert e.gmem_seq(-1) == n and e.gmem_seq(0) == n
        v = e.read_vars(); names = e.var_names()
        assert np.array_equal(v[:, names.index("ret")], ids * 0.5)
