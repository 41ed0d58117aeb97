"""Sample-pool read builtins against oracle/pool_ref.py, the restatement of
src/DspJsfxSamplePool.cpp:377-441 + the rt_sample_* wrappers. Indexing is integer-exact, values are float32->f64 exact.
No reference test pins these (parity unpinned); the restatement is the checker on CPU (port) and GPU."""
import numpy as np
import pytest

MEM_CAP = 65536


def _samples():
    rng = np.random.default_rng(3)
    return [rng.standard_normal((500, 2)).astype(np.float32),      # id 1 stereo
            rng.standard_normal(333).astype(np.float32),           # id 2 mono
            rng.standard_normal((64, 4)).astype(np.float32),       # id 3 four channels
            np.zeros((0, 1), np.float32)]                          # id 4 empty


# (op, sample id, a, b)
CASES = [
    (1, 1, 0, 0), (1, 1, 1, 499), (1, 1, 1, 499.4), (1, 1, 1, 499.5), (1, 1, 1, 500), (1, 1, 0, -0.4), (1, 1, 0, -0.6),
    (1, 1, 5, 10), (1, 1, -3, 10), (1, 2, 1, 100), (1, 3, 2.6, 63), (1, 4, 0, 0), (1, 0, 0, 0), (1, 9, 0, 0), (1, 1.4, 0, 7),
    (1, 1, 0, float("nan")),
    (2, 1, 0, 10.25), (2, 1, 1, 498.75), (2, 1, 1, 499.0), (2, 1, 0, 499.5), (2, 2, 0, -0.5), (2, 2, 0, 332.999),
    (3, 1, 12.0, 0), (3, 1, 499.0, 0), (3, 1, 499.000001, 0), (3, 1, -0.000001, 0), (3, 2, 40.4, 0), (3, 2, 40.5, 0), (3, 4, 0, 0),
    (4, 1, 12.75, 0), (4, 1, 498.5, 0), (4, 2, 331.25, 0), (4, 3, 62.5, 0), (4, 7, 1.0, 0),
    (5, 1, 0, 0), (5, 2, 0, 0), (5, 4, 0, 0), (5, 5, 0, 0), (6, 1, 0, 0), (6, 3, 0, 0), (7, 2, 0, 0),
    (8, 0, 0, 0), (8, 0, 3, 0), (8, 0, 4, 0), (8, 0, -1, 0),
    (9, 1, 100, 450), (9, 2, 200.4, 300), (10, 1, 1000, 470), (10, 2, 3000, 10), (9, 1, 65500, 0),
    (11, 0, 0, 0), (12, 1, 0, 0),
]


def _want(ref, mem, samples, op, sid, a, b):
    rl = rr = -7.0
    if op == 1:
        ret = ref.read(sid, a, b)
    elif op == 2:
        ret = ref.read_interp(sid, a, b)
    elif op in (3, 4):
        ret, rl, rr = ref.read2(sid, a, op == 4)
    elif op == 5:
        e = ref.entry(sid); ret = float(e[1]) if e else 0.0
    elif op == 6:
        e = ref.entry(sid); ret = float(e[2]) if e else 0.0
    elif op == 7:
        ret = 44100.0 if ref.entry(sid) else 0.0
    elif op == 8:
        i = int(np.floor(abs(a) + 0.5)) * (1 if a >= 0 else -1)
        ret = float(i + 1) if 0 <= i < len(samples) else 0.0
    elif op in (9, 10):
        ret = ref.export(mem, sid, a, b, 64, op == 10)
    elif op == 11:
        ret = len(samples) * 100 + 3.0
    elif op == 12:
        x = samples[int(sid) - 1]
        ret = float(np.float32(np.abs(x).max())) + float(np.float32(np.sqrt(np.mean(x.astype(np.float64) ** 2))))
    return ret, rl, rr


def _check(make_dut):
    from oracle import pool_ref
    samples = _samples()
    m0 = np.random.default_rng(8).standard_normal(MEM_CAP)
    for op, sid, a, b in CASES:
        if op == 9 and a == 65500:      # would need mem growth: fixed arena reports overflow instead
            continue
        ref = pool_ref.PoolRef(samples)
        mem = m0.copy()
        want = _want(ref, mem, samples, op, sid, a, b)
        got, dmem = make_dut(op, sid, a, b, samples, m0)
        tag = (op, sid, a, b)
        assert got[0] == want[0] or (np.isnan(got[0]) and np.isnan(want[0])), (tag, got, want)
        if op in (3, 4):
            assert got[1:] == want[1:], (tag, got, want)
        assert np.array_equal(dmem, mem), tag


def test_port_pool_reads():
    from oracle import port
    if not port.port_path("fx_poolkat").exists():
        pytest.skip("fixture port not built")

    def dut(op, sid, a, b, samples, m0):
        p = port.Port("fx_poolkat", 48000.0, mem_cap=MEM_CAP)
        p.pool_upload(samples, [44100] * len(samples))
        p.set_sliders([0, sid, 0, 0, 1]); p.prepare()
        p.mem_write(0, m0)
        p.set_sliders([op, sid, a, b, 1])
        p.process(np.zeros((2, 4), np.float32), 4)
        return (p.var("ret"), p.var("retL"), p.var("retR")), p.mem(0, MEM_CAP)

    _check(dut)


def _playback_ref(samples, sid, rate, frames):
    from oracle import pool_ref
    ref = pool_ref.PoolRef(samples)
    out = np.zeros((2, frames), np.float32)
    ph = 0.0
    for t in range(frames):
        ok, l, r = ref.read2(sid, ph, True)
        out[0, t], out[1, t] = (l, r) if ok else (0.0, 0.0)
        ph += rate
    return out


def test_port_pool_playback():
    from oracle import port
    if not port.port_path("fx_poolkat").exists():
        pytest.skip("fixture port not built")
    samples = _samples()
    for sid, rate in ((1, 0.73), (2, 1.31)):
        p = port.Port("fx_poolkat", 48000.0, mem_cap=MEM_CAP)
        p.pool_upload(samples, [44100] * len(samples))
        p.set_sliders([0, sid, 0, 0, rate]); p.prepare()
        y = p.process(np.zeros((2, 800), np.float32), 256)
        assert np.array_equal(y, _playback_ref(samples, sid, rate, 800))


@pytest.mark.gpu
def test_gpu_pool_reads():
    import zabatch

    def dut(op, sid, a, b, samples, m0):
        with zabatch.Engine("fx_poolkat", 1, mem_cap=MEM_CAP) as e:
            e.pool_upload(samples, [44100] * len(samples))
            e.set_sliders([0, sid, 0, 0, 1]); e.prepare()
            e.write_mem(0, m0)
            e.set_sliders([op, sid, a, b, 1])
            e.process_host(np.zeros((1, 2, 4), np.float32), block=4)
            v = e.read_vars()[0]; nm = e.var_names()
            return (v[nm.index("ret")], v[nm.index("retL")], v[nm.index("retR")]), e.read_mem(0, MEM_CAP)[0]

    _check(dut)


@pytest.mark.gpu
def test_gpu_pool_playback_batch():
    """130 instances, each playing its own (sample id, rate) out of one shared arena in HBM."""
    import zabatch
    samples = _samples()
    n, frames = 130, 700
    rows = np.zeros((n, 64)); rows[:, 1] = 1 + (np.arange(n) % 3); rows[:, 4] = 0.5 + 0.01 * np.arange(n)
    with zabatch.Engine("fx_poolkat", n) as e:
        e.pool_upload(samples, [44100] * len(samples))
        e.set_sliders(rows); e.prepare()
        y = e.process_host(np.zeros((n, 2, frames), np.float32), block=256)
    for i in (0, 1, 2, 64, 129):
        assert np.array_equal(y[i], _playback_ref(samples, rows[i, 1], rows[i, 4], frames)), i
