"""Catalog leaves through the translator-generated device kernels, against the reference VM's golden vectors.

Every leaf here runs @init, @slider, @block and @sample on the GPU (one lane per instance) and must reproduce what the
reference's own WDL/EEL2 VM produced for the same sliders and seeded noise: audio within 1e-5, vars/mem within 1e-8,
write high-water mark exactly. Instances are replicated so that several lanes of a wave and more than one wave are live.
"""
import numpy as np
import pytest

from conftest import AUDIO_EPS, GOLDEN, SCALAR_EPS, assert_state_close, dbfs, golden_input, leaf_of, load_golden

pytestmark = pytest.mark.gpu

CASES = sorted(p.stem for p in GOLDEN.glob("*_*.npz")
               if p.stem.endswith(("_default", "_alt", "_dense", "_slow")) and not p.stem.startswith("DDT"))


@pytest.mark.parametrize("ipw", ["auto", "64", "4"])
@pytest.mark.parametrize("case", CASES)
def test_leaf_matches_reference_vm(case, ipw, monkeypatch):
    """ipw = instances per wavefront of the lane-per-instance kernels (the engine spreads small batches over more wavefronts;
    ZAB_IPW pins it so that full, partial and single-lane wavefronts are all exercised)."""
    import zabatch
    if ipw != "auto":
        monkeypatch.setenv("ZAB_IPW", ipw)
    else:
        monkeypatch.delenv("ZAB_IPW", raising=False)
    leaf = leaf_of(case)
    if not zabatch.module_path(leaf).exists():
        pytest.skip(f"module for {leaf} not built")
    g = load_golden(case)
    big = int(g["mem_high"]) > (1 << 22)     # Contour / Texture / TextureXY address tens of millions of cells (options:maxmem)
    n = 3 if big else 70                     # two waves, the second one partially filled
    x = np.repeat(golden_input(g)[None], n, axis=0)
    with zabatch.Engine(leaf, n, srate=float(g["srate"]), mem_cap=max(65536, int(g["mem_high"]) + 64)) as e:
        assert e.nch == int(g["nch"])
        e.set_sliders(g["sliders"])
        e.prepare()
        names = e.var_names()
        assert names == [str(s) for s in g["var_names"]]
        prepared = e.read_vars()
        y = e.process_host(x, block=int(g["block"]))
        v = e.read_vars()
        high = e.mem_high()
        mem = e.read_mem(0, int(g["mem_high"])) if int(g["mem_high"]) and not big else None
        pages = {}
        if big:                              # every 1024-cell page the VM left something in, and a few it left empty
            want_pages = sorted(set((g["mem_idx"] // 1024).tolist()) | {0, 7, int(g["mem_high"]) // 1024 - 1})
            pages = {pg: e.read_mem(pg * 1024, 1024) for pg in want_pages}
    err = np.abs(y.astype(np.float64) - g["out"].astype(np.float64)[None]).max()
    print(f"{case}: null test max {dbfs(err):.1f} dBFS")
    assert err <= AUDIO_EPS
    for i in sorted({0, 1, min(63, n - 1), min(64, n - 1), n - 1}):
        assert_state_close(names, prepared[i], g["vars_prepared"], what=f"{case} prepared[{i}]")
        assert_state_close(names, v[i], g["vars"], what=f"{case} vars[{i}]")
    if mem is not None:
        want = np.zeros(int(g["mem_high"]))
        want[g["mem_idx"]] = g["mem_val"]
        assert np.abs(mem - want[None]).max() <= SCALAR_EPS
    for pg, got in pages.items():
        want = np.zeros(1024)
        sel = (g["mem_idx"] // 1024) == pg
        want[g["mem_idx"][sel] - pg * 1024] = g["mem_val"][sel]
        assert np.abs(got - want[None]).max() <= SCALAR_EPS, (case, pg)
    assert (high == int(g["mem_high"])).all()
    assert np.array_equal(y[0], y[n - 1]), "identical instances must produce identical audio"


def test_processor_mirror_parameter_push():
    """JsfxBatchProcessor: host parameter values go through the reference's float32 quantiser, changed rows re-run
    @slider at the next processBlock, untouched instances keep their state."""
    import zabatch
    from oracle import port
    from zajit import noise
    if not zabatch.module_path("DPT").exists():
        pytest.skip("DPT not built")
    n, frames = 4, 512
    proc = zabatch.JsfxBatchProcessor("DPT", n)
    proc.prepareToPlay(48000.0, 256)
    x = noise.white_noise(range(n), 2 * frames)
    y1 = proc.processBlock(x[:, :, :frames])
    first = min(proc.decls)                   # move the first declared slider of instance 2 only
    d = proc.decls[first]
    proc.set_parameter(first, d.vmin + 0.37 * (d.vmax - d.vmin), instance=2)
    y2 = proc.processBlock(x[:, :, frames:])
    rows = proc._slider_rows()
    proc.releaseResources()
    for i in range(n):
        p = port.Port("DPT", 48000.0)
        base = np.array(zabatch.leaf_meta("DPT")["default_sliders"])
        p.set_sliders(base); p.prepare()
        r1 = p.process(x[i, :, :frames], 256)
        if i == 2:
            p.set_sliders(rows[2]); p.run_slider()
        r2 = p.process(x[i, :, frames:], 256)
        assert np.abs(y1[i].astype(np.float64) - r1).max() <= AUDIO_EPS
        assert np.abs(y2[i].astype(np.float64) - r2).max() <= AUDIO_EPS


def test_dot_device_vs_port():
    """Spatialization/DOT (fft/fft_permute/ifft in @slider building a min-phase kernel). The reference's own EEL2 VM
    parses DOT.jsfx:372 differently from its AOT compiler (see tests/golden/make_golden.py), so this leaf is checked
    against the CPU port of the AOT lowering instead of a VM fixture: same translator text, g++ vs hipcc + device libm."""
    import zabatch
    from oracle import port
    from zajit import noise
    if not zabatch.module_path("DOT").exists() or not port.port_path("DOT").exists():
        pytest.skip("DOT not built")
    meta = zabatch.leaf_meta("DOT")
    n, frames = 66, 1536
    x = noise.white_noise(range(n), frames)
    rows = np.tile(np.array(meta["default_sliders"]), (n, 1))
    rows[1::3, 0] = 1; rows[2::3, 0] = 2; rows[5::7, 0] = 3        # all four topologies
    rows[:, 2] = np.linspace(5, 95, n)                             # brightness sweep -> different kernel lengths
    with zabatch.Engine("DOT", n, mem_cap=1 << 16) as e:
        e.set_sliders(rows); e.prepare()
        y = e.process_host(x, block=512)
        v = e.read_vars(); names = e.var_names()
        mem = e.read_mem(0, 57344)
    for i in (0, 1, 2, 5, 33, 65):
        p = port.Port("DOT", 48000.0, mem_cap=1 << 16)
        p.set_sliders(rows[i]); p.prepare()
        ref = p.process(x[i], 512)
        assert np.abs(y[i].astype(np.float64) - ref).max() <= AUDIO_EPS, i
        assert_state_close(names, v[i], p.vars(), what=f"DOT vars[{i}]")
        assert np.abs(mem[i] - p.mem(0, 57344)).max() <= SCALAR_EPS, i


BUS_LEAVES = ["IPCProbeA", "IPCProbeB", "3DPannerManager", "3DPanner"]     # scalar message bus


@pytest.mark.parametrize("leaf", BUS_LEAVES)
def test_bus_leaves_single_instance_device_vs_port(leaf):
    """The reference's leaves that talk over the message bus, one instance per engine (every send finds no peer and is counted
    as dropped, peer queries see only the instance itself), device vs CPU port; tests/test_msg_bus.py covers several instances."""
    import zabatch
    from oracle import port
    from zajit import noise
    if not zabatch.module_path(leaf).exists() or not port.port_path(leaf).exists():
        pytest.skip(f"{leaf} not built")
    meta = zabatch.leaf_meta(leaf)
    assert "msg" in meta["features"] and not any(f.startswith("host:") for f in meta["features"])
    nch, frames, cap = int(meta["nch"]), 1536, 1 << 16
    x = np.zeros((1, nch, frames), np.float32)
    x[:, :2] = noise.white_noise([5], frames)[:, :min(2, nch)]
    with zabatch.Engine(leaf, 1, mem_cap=cap, max_block=256) as e:
        e.set_sliders(meta["default_sliders"]); e.prepare()
        y = e.process_host(x, block=256)
        v = e.read_vars(); names = e.var_names()
    p = port.Port(leaf, 48000.0, mem_cap=cap)
    p.set_sliders(meta["default_sliders"]); p.prepare()
    ref = p.process(x[0], 256)
    assert np.abs(y[0].astype(np.float64) - ref).max() <= AUDIO_EPS
    assert_state_close(names, v[0], p.vars(), what=f"{leaf} vars")


def test_bus_rejects_more_instances_than_it_holds():
    import zabatch
    if not zabatch.module_path("IPCProbeA").exists():
        pytest.skip("IPCProbeA not built")
    with pytest.raises(zabatch.ZabError) as ei:
        zabatch.Engine("IPCProbeA", 257)
    assert ei.value.code == -1, ei.value


def test_host_only_builtins_are_refused_loudly():
    """Builtins that need the host itself (buffer messages, peer names, sample previews) go through the translator and load,
    but the engine must not run them with stubbed host calls: the device latches ZA_ERR_UNSUPPORTED and the C ABI returns
    ZAB_E_UNSUPPORTED."""
    import zabatch
    meta = zabatch.leaf_meta("fx_hostonly")
    assert any(f.startswith("host:") for f in meta["features"])
    with zabatch.Engine("fx_hostonly", 3, mem_cap=1 << 16) as e:
        e.set_sliders(meta["default_sliders"])
        with pytest.raises(zabatch.ZabError) as ei:
            e.prepare()
            e.process_host(np.zeros((3, e.nch, 64), np.float32), block=64)
        assert ei.value.code == -5, ei.value


FILE_LEAVES = ["PsychoConvolver", "Contour", "TextureXY", "Texture"]


@pytest.mark.parametrize("loaded", [False, True])
@pytest.mark.parametrize("leaf", FILE_LEAVES)
def test_file_slot_leaves_device_vs_port(leaf, loaded):
    """Leaves that load audio through file_open(0) / file_riff / file_mem (impulse responses, textures). The host side of
    the reference decodes the file; here the decoded items are handed over with zab_file_slot_set. Checked device vs CPU port
    (no VM fixture: the reference's shadow VM has no file slots here), with the slot empty (file_open -> -1, the state of a
    freshly inserted plugin) and with a short stereo noise burst assigned to slot 0."""
    import zabatch
    from oracle import port
    from zajit import noise
    if not zabatch.module_path(leaf).exists() or not port.port_path(leaf).exists():
        pytest.skip(f"{leaf} not built")
    meta = zabatch.leaf_meta(leaf)
    nch = int(meta["nch"])
    n, frames, cap = 5, 2048, 1 << 25
    ir = (noise.white_noise([321], 3000)[0].T * np.exp(-np.arange(3000) / 400.0)[:, None]).reshape(-1).astype(np.float64)   # interleaved L,R
    x = np.zeros((n, nch, frames), np.float32)
    x[:, :2] = noise.white_noise(range(n), frames)[:, :min(2, nch)]
    with zabatch.Engine(leaf, n, mem_cap=cap) as e:
        if loaded:
            e.file_slot_set(0, ir, channels=2, sample_rate=48000.0)
        e.set_sliders(meta["default_sliders"]); e.prepare()
        y = e.process_host(x, block=512)
        v = e.read_vars(); names = e.var_names()
        high = int(e.mem_high().max())
        mem = e.read_mem(0, min(high, 1 << 20), 0, 1)[0] if high else None
    p = port.Port(leaf, 48000.0, mem_cap=cap)
    if loaded:
        p.file_slot_set(0, ir, 2, 48000.0)
    p.set_sliders(meta["default_sliders"]); p.prepare()
    for i in (0, n - 1):
        q = p if i == 0 else None
        if q is None:
            q = port.Port(leaf, 48000.0, mem_cap=cap)
            if loaded:
                q.file_slot_set(0, ir, 2, 48000.0)
            q.set_sliders(meta["default_sliders"]); q.prepare()
        ref = q.process(x[i], 512)
        assert np.abs(y[i].astype(np.float64) - ref).max() <= AUDIO_EPS, (leaf, loaded, i)
        assert_state_close(names, v[i], q.vars(), what=f"{leaf} vars[{i}]")
        if i == 0 and mem is not None:
            assert int(q.mem_high) == high
            assert np.abs(mem - q.mem(0, len(mem))).max() <= SCALAR_EPS
    if loaded and leaf == "PsychoConvolver":
        assert np.abs(y).max() > 1.0          # the loaded impulse response is really convolved in


def test_cmd_bus_leaf_device_vs_port():
    """Spectral/CMD talks over gmem only (comm_join / instance_set_name succeed: an engine is one domain with one segment).
    One instance on the device against the CPU port: audio, vars, mem and the cells it wrote to the shared segment."""
    import zabatch
    from oracle import port
    from zajit import noise
    if not zabatch.module_path("CMD").exists() or not port.port_path("CMD").exists():
        pytest.skip("CMD not built")
    meta = zabatch.leaf_meta("CMD")
    frames = 2048
    x = noise.white_noise([3], frames)
    with zabatch.Engine("CMD", 1, mem_cap=1 << 20) as e:
        e.set_sliders(meta["default_sliders"]); e.prepare()
        y = e.process_host(x, block=512)
        v = e.read_vars(); names = e.var_names()
        cells = e.gmem_read(0, 1 << 16)
        high = int(e.mem_high()[0])
        mem = e.read_mem(0, high)[0]
    p = port.Port("CMD", 48000.0, mem_cap=1 << 20)
    p.set_sliders(meta["default_sliders"]); p.prepare()
    ref = p.process(x[0], 512)
    assert np.abs(y[0].astype(np.float64) - ref).max() <= AUDIO_EPS
    assert_state_close(names, v[0], p.vars(), what="CMD vars")
    assert int(p.mem_high) == high and np.abs(mem - p.mem(0, high)).max() <= SCALAR_EPS
    want = p.gmem_read(0, 1 << 16)
    assert np.array_equal(cells != 0, want != 0) and np.count_nonzero(cells) > 0        # same cells written
    assert np.abs(cells - want).max() <= SCALAR_EPS * max(1.0, np.abs(want).max())


@pytest.mark.parametrize("pooled", [False, True])
def test_sample_leaf_device_vs_port(pooled):
    """Generator/Sample, the catalog's largest script (4058 vars, 754 specialised functions: its user functions are real
    calls on the device, zajit/codegen.py ZA_OUTLINE_FNS): sample pool, gmem, rand, FFT, slider_change builtins in one leaf.
    No MIDI ports in a batch engine, so no note starts; the pool is queried every block all the same. Device vs CPU port."""
    import zabatch
    from oracle import port
    from zajit import noise
    if not zabatch.module_path("Sample").exists() or not port.port_path("Sample").exists():
        pytest.skip("Sample not built")
    meta = zabatch.leaf_meta("Sample")
    n, frames, cap = 3, 2048, 1 << 19
    x = noise.white_noise(range(n), frames)
    rng = np.random.default_rng(3)
    samples = [rng.standard_normal((3000, 2)).astype(np.float32) * 0.2, rng.standard_normal(1500).astype(np.float32) * 0.1]
    with zabatch.Engine("Sample", n, mem_cap=cap) as e:
        if pooled:
            e.pool_upload(samples, [44100, 48000])
        e.set_sliders(meta["default_sliders"]); e.prepare()
        y = e.process_host(x, block=512)
        v = e.read_vars(); names = e.var_names()
        assert int(e.mem_high().max()) <= cap
    for i in (0, n - 1):
        p = port.Port("Sample", 48000.0, mem_cap=cap)
        if pooled:
            p.pool_upload(samples, [44100, 48000])
        p.set_sliders(meta["default_sliders"]); p.prepare()
        ref = p.process(x[i], 512)
        assert np.abs(y[i].astype(np.float64) - ref).max() <= AUDIO_EPS, i
        assert_state_close(names, v[i], p.vars(), what=f"Sample vars[{i}]")
    assert np.abs(y).max() > 0


@pytest.mark.parametrize("ipw", ["auto", "1", "16", "64"])
def test_accumulation_loops_on_replica_lanes(ipw, monkeypatch):
    """tests/fixtures/coopkat.jsfx: every loop variant of the replica-lane form (zajit/emit.py e_Loop) against the CPU port, which
    runs the same loops serially: 64, 16, 4 and 1 lanes per instance, tap counts from 1 (serial fallback: fewer trips than lanes)
    to 200, one slider row per instance. Sums differ from the serial order by rounding only."""
    import zabatch
    from oracle import port
    from zajit import noise
    if ipw != "auto":
        monkeypatch.setenv("ZAB_IPW", ipw)
        monkeypatch.setenv("ZAB_LMEM", "0")          # (the LDS window would thin the waves further)
    else:
        monkeypatch.delenv("ZAB_IPW", raising=False)
    meta = zabatch.leaf_meta("fx_coopkat")
    assert "coop" in meta["features"]
    n, frames = 7, 1024
    rows = np.tile(np.array(meta["default_sliders"], dtype=np.float64), (n, 1))
    rows[:, 0] = [1, 2, 7, 48, 64, 129, 200]
    x = noise.white_noise(range(n), frames)
    with zabatch.Engine("fx_coopkat", n) as e:
        e.set_sliders(rows); e.prepare()
        y = e.process_host(x, block=256)
        v = e.read_vars(); names = e.var_names()
        assert e.launch_shape()[0] == (int(ipw) if ipw != "auto" else e.launch_shape()[0])
    for i in range(n):
        p = port.Port("fx_coopkat", 48000.0)
        p.set_sliders(rows[i]); p.prepare()
        ref = p.process(x[i], 256)
        assert np.abs(y[i].astype(np.float64) - ref).max() <= AUDIO_EPS, (i, rows[i][0])
        assert_state_close(names, v[i], p.vars(), what=f"coopkat vars[{i}] taps {rows[i][0]}")


@pytest.mark.parametrize("ipw", ["auto", "1", "4", "16", "64"])
def test_map_loops_on_replica_lanes(ipw, monkeypatch):
    """tests/fixtures/mapkat.jsfx: every variant of the elementwise-loop form (zajit/emit.py _map_plan, zart.h za_map_ok) and the
    shared memcpy / memset against the CPU port, which runs the same loops serially: independent trips (disjoint buffers, one
    cell per trip, interleaved pairs, modulo loads, aliasing function arguments, downward and double counters) and trips that
    depend on each other (overlapping shifts, which the guard must send to the serial form), lengths from fewer trips than
    lanes to 400, at 64 ... 1 lanes per instance. Same operations per element: the arena must match to the state tolerance
    (a trip run twice, skipped or out of order shows at the size of the data)."""
    import zabatch
    from oracle import port
    from zajit import noise
    if ipw != "auto":
        monkeypatch.setenv("ZAB_IPW", ipw)
        monkeypatch.setenv("ZAB_LMEM", "0")
    else:
        monkeypatch.delenv("ZAB_IPW", raising=False)
    meta = zabatch.leaf_meta("fx_mapkat")
    assert "coopmap" in meta["features"] and "coop" in meta["features"]
    n, frames = 7, 700
    rows = np.tile(np.array(meta["default_sliders"], dtype=np.float64), (n, 1))
    rows[:, 0] = [1, 3, 8, 96, 129, 256, 400]
    x = noise.white_noise(range(n), frames)
    with zabatch.Engine("fx_mapkat", n) as e:
        e.set_sliders(rows); e.prepare()
        y = e.process_host(x, block=256)
        v = e.read_vars(); names = e.var_names()
        mem = e.read_mem(0, 4608)
        high = e.mem_high()
    for i in range(n):
        p = port.Port("fx_mapkat", 48000.0)
        p.set_sliders(rows[i]); p.prepare()
        ref = p.process(x[i], 256)
        bad = np.flatnonzero(np.abs(mem[i] - p.mem(0, 4608)) > SCALAR_EPS)       # (@init's sin / cos differ by an ulp between device and host)
        assert bad.size == 0, (i, rows[i][0], bad[:8])
        assert np.abs(y[i].astype(np.float64) - ref).max() <= AUDIO_EPS, (i, rows[i][0])
        assert_state_close(names, v[i], p.vars(), what=f"mapkat vars[{i}] length {rows[i][0]}")
        assert high[i] == p.mem_high, (i, high[i], p.mem_high)


@pytest.mark.parametrize("leaf,n,want", [("fx_stft", 1024, 1), ("fx_stft", 2048, 2), ("fx_stft", 8192, 8), ("fx_coopkat", 512, 1),
                                         ("fx_fftbench_full", 256, 1), ("fx_fftbench_full", 1024, 4)])
def test_replica_lane_leaves_are_given_a_wavefront_count(leaf, n, want, monkeypatch):
    """csrc/zabatch.hip: leaves with cooperative transforms / shared loops on their audio path run on ~1024 wavefronts (~256 when
    each holds a 64 KB transform buffer), from ONE instance per wavefront up."""
    import zabatch
    monkeypatch.delenv("ZAB_IPW", raising=False)
    monkeypatch.delenv("ZAB_FFT_WAVES", raising=False)
    with zabatch.Engine(leaf, n) as e:
        e.set_sliders(zabatch.leaf_meta(leaf)["default_sliders"]); e.prepare()
        assert e.launch_shape()[0] == want


def test_modulo_shortcuts_match_the_division():
    """csrc/zart.h za_mod: the device skips the integer division for 0 <= l < r, l == r and power-of-two divisors; the CPU port
    always divides. tests/fixtures/modkat.jsfx walks counters through divisors of every sign (and zero) -- the accumulated sum
    of all results must agree exactly."""
    import zabatch
    from oracle import port
    from zajit import noise
    meta = zabatch.leaf_meta("fx_modkat")
    divs = [7, 1, 16, -5, 0, 33, -16, 40]
    n, frames = len(divs), 3000
    rows = np.tile(np.array(meta["default_sliders"], dtype=np.float64), (n, 1))
    rows[:, 0] = divs
    rows[:, 1] = [0, 5, -37, 100, 3, -100, 64, -1]
    x = noise.white_noise(range(n), frames)
    with zabatch.Engine("fx_modkat", n) as e:
        e.set_sliders(rows); e.prepare()
        y = e.process_host(x, block=512)
        v = e.read_vars(); names = e.var_names()
    for i in range(n):
        p = port.Port("fx_modkat", 48000.0)
        p.set_sliders(rows[i]); p.prepare()
        ref = p.process(x[i], 512)
        pv = p.vars()
        for k, nm in enumerate(names):
            if nm.startswith("m") or nm == "acc":
                assert v[i][k] == pv[k], (divs[i], nm, v[i][k], pv[k])
        assert np.abs(y[i].astype(np.float64) - ref).max() <= AUDIO_EPS, (i, divs[i])


def test_script_originated_slider_changes_reach_the_host_mirror():
    """consumeDspSliderChanges / pushParamsToStateSliders (src/JSFXJuceProcessor.cpp:5665-5739, 9286-9357) through
    JsfxBatchProcessor: a slider the script sets and announces with sliderchange() becomes the host parameter and is not
    overwritten by the next push; @slider runs for it exactly once; an unannounced slider write is overwritten by the host
    value without re-running @slider."""
    import zabatch
    from zajit import noise
    if not zabatch.module_path("fx_slidewrite").exists():
        pytest.skip("fx_slidewrite not built")
    n, block, nblocks = 3, 64, 7
    proc = zabatch.JsfxBatchProcessor("fx_slidewrite", n)
    proc.prepareToPlay(48000.0, block)
    x = noise.white_noise(range(n), block * nblocks)
    outs = [proc.processBlock(x[:, :, k * block:(k + 1) * block]) for k in range(nblocks)]
    names = proc.engine.var_names()
    v = proc.engine.read_vars()
    sl = proc.engine.get_sliders()
    host = proc.host_params.copy()
    proc.releaseResources()
    y = np.concatenate(outs, axis=2)
    gain = np.ones(block * nblocks); gain[2 * block:] = 1.45            # @slider runs inside block 3, before its samples
    want = (x.astype(np.float64) * gain[None, None, :]).astype(np.float32)
    assert np.abs(y.astype(np.float64) - want).max() <= AUDIO_EPS
    assert (v[:, names.index("nsl")] == 2).all(), v[:, names.index("nsl")]       # prepare + the announced change, nothing else
    assert (v[:, names.index("cnt")] == nblocks).all()
    assert np.allclose(v[:, names.index("g")], 0.45, atol=1e-12)
    assert (sl[:, 0] == 4.5).all() and (host[:, 0] == 4.5).all()
    assert (sl[:, 1] == 3.0).all() and (v[:, names.index("seen2")] == 3.0).all()   # the unannounced 6.3 was pushed over


@pytest.mark.parametrize("path", ["fast", "generic"])
def test_unconsumed_slider_changes_survive_a_checkpoint(path):
    """the OR of the slider masks a script raised since the host last looked lives in its own word; a checkpoint
    taken between the launch that raised one and the host's next zab_consume_slider_changes must carry it (on both kernels:
    the time-parallel kernel runs @block / @slider itself and collects the masks the same way)."""
    import zabatch
    from zajit import noise
    if not zabatch.module_path("fx_slidewrite").exists():
        pytest.skip("fx_slidewrite not built")
    sel = zabatch.ZAB_PATH_FAST if path == "fast" else zabatch.ZAB_PATH_GENERIC
    n, block = 3, 64
    x = noise.white_noise(range(n), block * 4)
    with zabatch.Engine("fx_slidewrite", n, max_block=block, path=sel) as e:
        e.set_sliders(zabatch.leaf_meta("fx_slidewrite")["default_sliders"]); e.prepare()
        e.consume_slider_changes()
        e.process_host(x, block=block)                       # block 3 of the launch raises slider 1's mask
        ck = e.checkpoint()
        assert (ck["changes"] == 1).all(), ck["changes"]
        masks, rows = e.consume_slider_changes()
        assert (masks == 1).all() and (rows[:, 0] == 4.5).all()
        assert (e.consume_slider_changes()[0] == 0).all()
    with zabatch.Engine("fx_slidewrite", n, max_block=block, path=sel) as e2:
        e2.set_sliders(zabatch.leaf_meta("fx_slidewrite")["default_sliders"]); e2.prepare()
        e2.restore(ck)
        masks, rows = e2.consume_slider_changes()
        assert (masks == 1).all() and (rows[:, 0] == 4.5).all()
