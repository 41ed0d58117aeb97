"""Parity of the HIP hot path for Spatialization/DDT (the north-star leaf), through the C ABI.

Checker: golden vectors produced by the reference's own WDL/EEL2 VM (tests/golden/make_golden.py), and the CPU port
(oracle/port.py) for seeds / sizes the fixtures do not hold. Tolerances are the reference's (conftest).
"""
import numpy as np
import pytest

from conftest import AUDIO_EPS, SCALAR_EPS, assert_state_close, dbfs, golden_input, load_golden

pytestmark = pytest.mark.gpu

DDT_CASES = ["DDT_default", "DDT_far_extreme", "DDT_near_eco_direct", "DDT_diffuse_ragged"]
# the fast kernels run NW wavefronts per instance (picked from the batch size); ZAB_DDT_NW pins it so every variant is covered.
# fastN: zab_ddt_fast (filtered rings, long launches); wideN[p|d]: zab_ddt_wide, the single-history-ring kernel that takes
# short launches and delays too long for two rings (ZAB_DDT_KERNEL pins either; p / d pin wide's ring addressing mode)
# fastNw: the same kernel compiled for three waves per SIMD (168 registers, single-buffered taps; ZAB_DDT_MINW pins it -- unpinned
# it takes launches of >= 160 000 frames on batches of >= 1024 instances)
FAST_VARIANTS = ["fast1", "fast2", "fast4", "fast8", "fast2w", "fast4w", "fast8w", "wide1", "wide2", "wide8", "wide1p", "wide2p", "wide1d"]


@pytest.fixture(autouse=True)
def _unpin_nw(monkeypatch):
    monkeypatch.delenv("ZAB_DDT_NW", raising=False)
    monkeypatch.delenv("ZAB_DDT_RING", raising=False)
    monkeypatch.delenv("ZAB_DDT_KERNEL", raising=False)
    monkeypatch.delenv("ZAB_DDT_MINW", raising=False)


def _paths(zabatch, monkeypatch):
    """(name, path) pairs; selecting one pins the wave count through the environment."""
    def select(name):
        if name.startswith(("fast", "wide")):
            monkeypatch.setenv("ZAB_DDT_NW", name[4])
            monkeypatch.delenv("ZAB_DDT_RING", raising=False)
            monkeypatch.setenv("ZAB_DDT_KERNEL", name[:4])          # (unpinned, the launch length picks the kernel)
            monkeypatch.setenv("ZAB_DDT_MINW", "3" if name[5:] == "w" else "2")
            if name[5:] in ("p", "d"):          # power-of-two ring with masked offsets / doubled ring without wrap
                monkeypatch.setenv("ZAB_DDT_RING", {"p": "pow2", "d": "dbl"}[name[5:]])
            return zabatch.ZAB_PATH_FAST
        return zabatch.ZAB_PATH_GENERIC
    return select


def _run(zabatch, path, g, n=3):
    x1 = golden_input(g)
    x = np.repeat(x1[None], n, axis=0)
    with zabatch.Engine("DDT", n, srate=float(g["srate"]), path=path) as e:
        e.set_sliders(g["sliders"])
        e.prepare()
        prepared = e.read_vars()
        y = e.process_host(x, block=int(g["block"]))
        fast = e.used_fast_path()
        return y, prepared, e.read_vars(), e.read_mem(0, int(g["mem_high"])), e.mem_high(), e.var_names(), fast


@pytest.mark.parametrize("case", DDT_CASES)
@pytest.mark.parametrize("path_name", ["generic"] + FAST_VARIANTS)
def test_ddt_matches_reference_vm(case, path_name, monkeypatch):
    import zabatch
    g = load_golden(case)
    path = _paths(zabatch, monkeypatch)(path_name)
    y, prepared, vars_, mem, high, names, fast = _run(zabatch, path, g)
    assert fast == path_name.startswith(("fast", "wide"))
    assert names == [str(s) for s in g["var_names"]]
    # state after prepareToPlay (@init + @slider on the device)
    for i in range(y.shape[0]):
        assert_state_close(names, prepared[i], g["vars_prepared"], what=f"{case} prepared vars[{i}]")
    # audio
    err = np.abs(y.astype(np.float64) - g["out"].astype(np.float64)[None]).max()
    print(f"{case} [{path_name}] null test: max {dbfs(err):.1f} dBFS")
    assert err <= AUDIO_EPS
    if path_name == "generic":
        assert err == 0.0, "the serial device path is expected to be bit-identical to the reference VM on DDT"
    # final state: every instance got the same input, so every row must match the fixture
    want_mem = np.zeros(int(g["mem_high"]))
    want_mem[g["mem_idx"]] = g["mem_val"]
    for i in range(y.shape[0]):
        assert_state_close(names, vars_[i], g["vars"], what=f"{case} vars[{i}]")
        assert np.abs(mem[i] - want_mem).max() <= SCALAR_EPS
    assert (high >= int(g["mem_high"])).all()


@pytest.mark.parametrize("frames", [1237, 100, 257, 2048])
def test_fast_equals_generic_on_distinct_instances(frames, monkeypatch):
    """Distinct noise + distinct slider sets per instance; ragged / tiny / aligned frame counts; every device path vs the CPU port."""
    import zabatch
    from oracle import port
    from zajit import noise
    meta = zabatch.leaf_meta("DDT")
    n, block = 6, 300
    x = noise.white_noise(range(100, 100 + n), frames)
    rows = np.tile(np.array(meta["default_sliders"]), (n, 1))
    rows[:, 0] = [0, 15, 45, 70, 100, 30]
    rows[:, 4] = [0, 1, 2, 3, 4, 2]
    rows[:, 7] = [0, 1, 2, 3, 0, 0]
    rows[:, 8] = [5, 25, 50, 75, 100, 50]
    res = {}
    for name in ["generic"] + FAST_VARIANTS:
        path = _paths(zabatch, monkeypatch)(name)
        with zabatch.Engine("DDT", n, path=path) as e:
            e.set_sliders(rows)
            e.prepare()
            y = e.process_host(x, block=block)
            res[name] = (y, e.read_vars(), e.read_mem(0, 33248))
            names = e.var_names()
    for i in range(n):
        p = port.Port("DDT", 48000.0)
        p.set_sliders(rows[i]); p.prepare()
        ref = p.process(x[i], block)
        for name in res:
            y, v, m = res[name]
            err = np.abs(y[i].astype(np.float64) - ref).max()
            assert err <= AUDIO_EPS, (name, i, err)
            assert_state_close(names, v[i], p.vars(), what=f"{name} vars[{i}]")
            assert np.abs(m[i] - p.mem(0, 33248)).max() <= SCALAR_EPS, (name, i)
    assert np.array_equal(res["generic"][0], res["generic"][0])


def test_multi_call_continuity_and_slider_change(monkeypatch):
    """Three zab_process calls with a slider move in between == one reference run with the same schedule."""
    import zabatch
    from oracle import port
    from zajit import noise
    meta = zabatch.leaf_meta("DDT")
    n = 2
    x = noise.white_noise([7, 8], 3 * 700)
    row2 = np.array(meta["default_sliders"]); row2[0] = 62.0; row2[8] = 80.0
    for name in ["generic"] + FAST_VARIANTS:
        path = _paths(zabatch, monkeypatch)(name)
        with zabatch.Engine("DDT", n, path=path) as e:
            e.set_sliders(meta["default_sliders"]); e.prepare()
            ys = [e.process_host(x[:, :, 0:700], block=512)]
            e.set_sliders(row2)
            ys.append(e.process_host(x[:, :, 700:1400], block=512))
            ys.append(e.process_host(x[:, :, 1400:2100], block=512))
            y = np.concatenate(ys, axis=2)
            v = e.read_vars()
            names = e.var_names()
        for i in range(n):
            p = port.Port("DDT", 48000.0)
            p.set_sliders(meta["default_sliders"]); p.prepare()
            r = [p.process(x[i, :, 0:700], 512)]
            p.set_sliders(row2); p.run_slider()
            r.append(p.process(x[i, :, 700:1400], 512))
            r.append(p.process(x[i, :, 1400:2100], 512))
            ref = np.concatenate(r, axis=1)
            assert np.abs(y[i].astype(np.float64) - ref).max() <= AUDIO_EPS
            assert_state_close(names, v[i], p.vars(), what=f"vars[{i}] path {path}")


def test_long_launches_pick_the_filtered_ring_kernel_and_match_the_port():
    """Unpinned: a 100 000-frame launch goes to zab_ddt_fast (>= 96 000 frames), a 5 000-frame one to zab_ddt_wide. Random slider rows
    per instance (distances, sizes, tap densities, monitor modes), a slider move between two long launches (the a^n K term
    carries a real state into changed taps), then a short launch: audio and every var against the CPU port."""
    import zabatch
    from oracle import port
    from zajit import noise, sliders as zs
    meta = zabatch.leaf_meta("DDT")
    rng = np.random.default_rng(20261004)
    n, f1, f2, f3 = 6, 100_000, 96_000, 5_000
    rows = np.tile(np.array(meta["default_sliders"]), (n, 1))
    rows2 = rows.copy()
    for r in (rows, rows2):
        r[:, 0] = rng.uniform(0, 100, n); r[:, 1] = rng.uniform(0, 100, n); r[:, 2] = rng.uniform(0, 100, n)
        r[:, 3] = rng.uniform(0, 100, n); r[:, 4] = rng.integers(0, 5, n); r[:, 8] = rng.uniform(0, 100, n)
    rows[:, 7] = [0, 1, 2, 3, 0, 0]; rows2[:, 7] = rows[:, 7]
    x = noise.white_noise(range(300, 300 + n), f1 + f2 + f3)
    with zabatch.Engine("DDT", n) as e:
        e.set_sliders(rows); e.prepare()
        ys = [e.process_host(x[:, :, :f1], block=512)]
        k1 = e.last_kernel_name()
        e.set_sliders(rows2)
        ys.append(e.process_host(x[:, :, f1:f1 + f2], block=512))
        k2 = e.last_kernel_name()
        ys.append(e.process_host(x[:, :, f1 + f2:], block=512))
        k3 = e.last_kernel_name()
        v = e.read_vars(); names = e.var_names()
    assert k1.startswith("zab_ddt_fast") and k2.startswith("zab_ddt_fast") and k3.startswith("zab_ddt_wide"), (k1, k2, k3)
    y = np.concatenate(ys, axis=2)
    for i in range(n):
        p = port.Port("DDT", 48000.0)
        p.set_sliders(rows[i]); p.prepare()
        r = [p.process(x[i, :, :f1], 512)]
        p.set_sliders(rows2[i]); p.run_slider()
        r.append(p.process(x[i, :, f1:f1 + f2], 512))
        r.append(p.process(x[i, :, f1 + f2:], 512))
        ref = np.concatenate(r, axis=1)
        err = np.abs(y[i].astype(np.float64) - ref).max()
        assert err <= AUDIO_EPS, (i, err)
        assert_state_close(names, v[i], p.vars(), what=f"vars[{i}]")


def test_device_resident_buffers_and_noise_generator():
    import zabatch
    from zajit import noise
    meta = zabatch.leaf_meta("DDT")
    n, frames = 5, 2048
    with zabatch.Engine("DDT", n) as e:
        e.set_sliders(meta["default_sliders"]); e.prepare()
        nbytes = n * 2 * frames * 4
        d_in, d_out = e.device_alloc(nbytes), e.device_alloc(nbytes)
        e.device_noise(d_in, frames)
        x = e.download(d_in, (n, 2, frames))
        assert np.array_equal(x, noise.white_noise(range(n), frames)), "device noise != host twin"
        e.process_device(d_in, d_out, frames); e.sync()
        y = e.download(d_out, (n, 2, frames))
        ms, launches = e.last_timing()
        assert ms > 0 and launches >= 1
    with zabatch.Engine("DDT", n) as e2:
        e2.set_sliders(meta["default_sliders"]); e2.prepare()
        assert np.array_equal(e2.process_host(x, 512), y)


def test_mem_overflow_fails_loudly():
    import zabatch
    meta = zabatch.leaf_meta("DDT")
    with zabatch.Engine("DDT", 2, mem_cap=4096) as e:
        e.set_sliders(meta["default_sliders"])
        with pytest.raises(zabatch.ZabError) as ei:
            e.prepare()
            e.process_host(np.zeros((2, 2, 64), np.float32))
        assert ei.value.code == -4
