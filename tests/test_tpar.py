"""Time-parallel lowering (zajit/tpar.py): one wavefront per instance, lane = frame.

CPU part (no GPU): the analysis and its staged algorithm -- classification of every recurrence, affine coefficients,
scans, shifts, shared serial loops, chunk carries, partial chunks -- through `Plan.simulate`, a numpy restatement with one
array element per lane, against the reference VM's golden vectors (repo-authored recurrence zoo always; the catalog leaves
where their scripts are present, i.e. in the dev container).
GPU part: the generated `zab_<leaf>_tpar` kernels against the same golden vectors, against the generic kernel over long
runs, and across launch boundaries.
"""
from pathlib import Path

import numpy as np
import pytest

from conftest import AUDIO_EPS, GOLDEN, ROOT, SCALAR_EPS, assert_state_close, dbfs, golden_input, leaf_of, load_golden

REF_PLUGINS = Path("/root/reference/plugins")
FIXTURES = ROOT / "tests" / "fixtures"
TPAR_CATALOG = ["ADS", "ATTACK", "RTT", "SaliencePush", "BedRock", "DPT", "Roomalizer", "EasyExpander", "SOMA"]
# leaves whose @sample has a rare heavy branch (tpar.split_events): the frame it falls on runs with the serial section code
TPAR_EVENT_LEAVES = ["PsychoConvolver", "PsychoConvolver+IR", "NeuroCV", "fx_evtkat", "fx_evtkat2", "fx_guardkat", "fx_stft", "fx_stft4k", "fx_stftparts", "fx_convkat", "fx_realperm", "fx_mapkat", "fx_ringio"]
TPAR_BLOCK_CATALOG = ["ERBTilt", "SpectralStabilizer", "TSEQ"]        # leaves with @block: the kernel runs it between the blocks
TPAR_FIXTURES = ["fx_dynkat_default", "fx_dynkat_hot", "fx_randkat_default", "fx_ringkat_default", "fx_ringkat_long",
                 "fx_delaytaps_default", "fx_delaytaps_far",
                 "fx_statekat_default", "fx_statekat_alt"]       # round 4: three / four coupled states, floor() wraps (zt_scanN)
# round 4: voices in mem[] beside gathers, a feedback echo above / below a chunk's length, two delay lines in one buffer
TPAR_R4_FIXTURES = ["fx_voicekat_default", "fx_voicekat_alt"]
# catalog leaves that got their time-parallel kernel in round 4 (statements the lowering cannot take run as events)
TPAR_R4_CATALOG = ["Alias", "Contour", "Texture", "TextureXY"]
TPAR_ABORTS = ["fx_ringabort_default", "fx_ringabort_stride2",      # launches the kernel hands (partly) to the generic code
               "fx_voicekat_dense"]                                 # (a gather that hits a per-trip cell a loop stores to)


def _source(leaf):
    if leaf.startswith("fx_"):
        return FIXTURES / (leaf[3:] + ".jsfx")
    hits = sorted(REF_PLUGINS.glob(f"*/{leaf}/src/*.jsfx")) if REF_PLUGINS.exists() else []
    return hits[0] if hits else None


def _plan(leaf):
    from zajit import program, tpar
    src = _source(leaf)
    if src is None or not src.exists():
        pytest.skip(f"script of {leaf} not present here")
    prog = program.analyse_file(src)
    nch = max(1, min(64, int(prog.io["process"])))
    return tpar.build_plan(prog, nch), prog


def _prepared_arena(leaf, g):
    """mem[] and its write high-water mark after prepare (@init / @slider), from the CPU port."""
    from oracle import port
    if not port.port_path(leaf).exists():
        pytest.skip(f"port of {leaf} not built")
    p = port.Port(leaf, float(g["srate"]))
    p.set_sliders(g["sliders"]); p.prepare()
    return p.mem(0, max(1 << 16, int(g["mem_high"]) + 64)), p.mem_high


def test_delay_line_conditions_are_checked_chunk_by_chunk():
    """A write position that jumps twice inside one chunk is not a ring's wrap: the staged algorithm stops at that chunk (the
    kernel hands the rest of the launch to the generic code); a stride of two stops it at frame 0."""
    from zajit import tpar
    plan, _ = _plan("fx_ringabort")
    for case, f0 in (("fx_ringabort_default", 960), ("fx_ringabort_stride2", 0)):
        g = load_golden(case)
        names = [str(s) for s in g["var_names"]]
        v0 = {n: (0.0 if np.isnan(v) else float(v)) for n, v in zip(names, g["vars_prepared"])}
        mem0, _ = _prepared_arena("fx_ringabort", g)
        with pytest.raises(tpar.TparAbort) as ei:
            plan.simulate(v0, golden_input(g), sliders=g["sliders"], mem=mem0)
        assert ei.value.f0 == f0


def test_recurrence_zoo_classification():
    plan, _ = _plan("fx_dynkat")
    kinds = {}
    for it in plan.items:
        if it[0] == "scan":
            kinds[tuple(it[1].names)] = "scan"
        elif it[0] in ("serial", "spec"):
            for c in it[1]:
                kinds[tuple(c.names)] = it[0]
        elif it[0] == "shift":
            kinds[(it[1],)] = "shift"
    assert kinds[("xpL",)] == kinds[("xpR",)] == kinds[("lastsign",)] == "shift"               # delayed signals
    for one in ("dcL", "dcR", "lpL", "lpR", "cnt", "flips", "heldv", "tv"):                   # affine, one state
        assert kinds[(one,)] == "scan", one
    assert kinds[("z1", "z2")] == "scan" and kinds[("swa", "swb")] == "scan"                    # coupled affine pairs
    for one in ("gr", "pk", "hold", "__fnlocal__sample__follow__e"):                          # state-dependent conditions
        assert kinds[(one,)] == "spec", one
    # y = y + step: a step that is itself a signal has no value the sum is meant to land on, so these re-associate like any
    # other affine recurrence (a block-constant fractional step keeps its serial order: test_constant_steps_keep_their_order)
    assert kinds[("ph",)] == "spec" and kinds[("acc",)] == "scan"
    assert plan.stats["spec_loops"] == 2


def test_switched_recurrences_fall_back_to_the_serial_loop(monkeypatch):
    """With the iteration budget cut to one pass some chunks of the peak hold and the smoothers do not reach their fixed point;
    the serial loop takes over for those chunks and the result is unchanged."""
    from zajit import tpar
    plan, _ = _plan("fx_dynkat")
    g = load_golden("fx_dynkat_default")
    names = [str(s) for s in g["var_names"]]
    v0 = {n: (0.0 if np.isnan(v) else float(v)) for n, v in zip(names, g["vars_prepared"])}
    monkeypatch.setattr(tpar, "SPEC_MAX", 1)
    y, va, _ = plan.simulate(v0, golden_input(g), sliders=g["sliders"], srate=float(g["srate"]))
    assert any(not ok for _, _, ok in plan.spec_log) and any(ok for _, _, ok in plan.spec_log)
    assert np.abs(y.astype(np.float64) - g["out"]).max() <= AUDIO_EPS
    assert_state_close(names, [va.get(n, 0.0) for n in names], g["vars"], what="vars")


def _plan_of_text(sample, init="", block=""):
    from zajit import program, tpar
    text = "desc:t\n@init\n" + init + ("\n@block\n" + block if block else "") + "\n@sample\n" + sample + "\n"
    return tpar.try_plan(program.analyse(text, name="t"), 2)


def test_rare_heavy_branches_become_events():
    """`cond ? ( loops / builtins with effects )` statements are cut out of the frame (tpar.split_events): the plan keeps their
    conditions, evaluates them first in every chunk, and the frame one falls on runs with the script's own section code."""
    from zajit import program, tpar
    for name in ("stft", "stft4k", "stftparts", "convkat", "ringio", "mapkat"):
        plan, msg = tpar.try_plan(program.analyse_file(FIXTURES / f"{name}.jsfx"), 2)
        assert plan is not None and plan.stats["events"] == 1, (name, msg)
        kinds = [it[0] for it in plan.top.items]
        assert kinds.count("cut") == 1
        before = plan.top.items[:kinds.index("cut")]
        assert all(it[0] in ("par", "scan", "shift", "spec", "serial") for it in before)       # no loads, stores or loops before the cut
        with pytest.raises(NotImplementedError):      # (the simulator stops where an event is due: within a hop of the start --
            plan.simulate({}, np.zeros((2, 4096), dtype=np.float32))      # the hop sizes are literals now, not zeroed variables)
    # nested in a block-constant conditional: the event's condition carries the path's
    plan, msg = _plan_of_text("on ? ( buf[n] = spl0; n += 1; n >= 256 ? ( fft(buf, 256); n = 0; ); ); spl0 = buf[0];")
    assert plan is not None and plan.stats["events"] == 1, msg
    # a body that reads what the frame before left in a variable the rest of the frame writes first: every chunk stores it
    plan, msg = _plan_of_text("n += 1; n >= 64 ? ( y = t + 1; fft(buf, 64); n = 0; ); t = spl0 * 2; spl0 = y;")
    assert plan is not None and plan.event_exposed == ["t"], msg
    plan, msg = _plan_of_text("n += 1; t = spl0 * 2; n >= 64 ? ( y = t + 1; fft(buf, 64); n = 0; ); spl0 = y;")
    assert plan is not None and plan.event_exposed == [], msg
    # with an else arm: left alone (and then unsupported for what the body holds)
    plan, msg = _plan_of_text("n += 1; n >= 64 ? ( fft(buf, 64); n = 0; ) : ( buf[n] = spl0; );")
    assert plan is None, msg


def test_statements_the_lowering_cannot_take_become_events():
    """Round 4: split_events picks the obvious rare branches up front; beyond that ANY conditional without an else arm, loop or
    right operand of && / || whose body the walk cannot lower -- wherever @sample reaches it, user functions included -- runs as
    an event under its own condition (FrameGraph._or_event), and the innermost such statement is the one chosen. Inside a uniform
    loop the condition must be wave-uniform per trip: it is tested trip by trip before a segment starts (LoopInfo.guards)."""
    from zajit import tpar
    # a condition over a mem[] cell (a named state, not a read at a moving address)
    plan, msg = _plan_of_text("buf[9] += 1; buf[9] >= 64 ? ( fft(buf + 64, 64); buf[9] = 0; ); spl0 = buf[70];")
    assert plan is not None and plan.stats["events"] == 1 and plan.stats["dyn_events"] == 1, msg
    # inside a function, two levels down
    plan, msg = _plan_of_text("n += 1; tick(n); spl0 = buf[0];",
                              init="function work() ( fft(buf, 64); ); function tick(k) ( k >= 64 ? ( work(); n = 0; ); );")
    assert plan is not None and plan.stats["events"] == 1 and "fft" in plan.dyn_events[0], msg
    # a loop whose trip count differs from frame to frame: the frames in which it runs at all
    plan, msg = _plan_of_text("n = floor(abs(spl0) * 8); k = 0; loop(n, k += 1); spl0 = k;")
    assert plan is not None and plan.stats["events"] == 1 and "count differs" in plan.dyn_events[0], msg
    plan, msg = _plan_of_text("k = 0; while (k < abs(spl0) * 4) ( k += 1; ); spl1 = k;")
    assert plan is not None and plan.stats["events"] == 1, msg
    # the right operand of a short circuit
    plan, msg = _plan_of_text("n += 1; (n >= 64) && ( fft(buf, 64); n = 0; ); spl0 = buf[0];")
    assert plan is not None and plan.stats["events"] == 1, msg
    # found after the walk: a conditional store into a delay line that @sample reads
    plan, msg = _plan_of_text("abs(spl0) > 0.5 ? ( ring[wp] = spl0; ); spl1 = ring[wp - 5]; wp += 1;", init="ring = 3000; wp = 100;")
    assert plan is not None and plan.stats["events"] == 1 and "conditional store" in plan.dyn_events[0], msg
    # inside a uniform loop: a guard of the loop, wave-uniform per trip (here: a cell only the event's body stores to)
    plan, msg = _plan_of_text("k = 0; loop(4, act[k] > 0 ? ( fft(buf + 64 * k, 64); act[k] = 0; ); k += 1; ); spl0 = buf[0];",
                              init="act = 500; buf = 1000;")
    assert plan is not None and plan.stats["events"] == 0 and plan.stats["loop_guards"] == 1, msg
    mem = np.zeros(4096)
    plan.simulate({"act": 500.0, "buf": 1000.0}, np.zeros((2, 64), dtype=np.float32), mem=mem)      # no voice active: nothing is due
    mem[502] = 1.0
    with pytest.raises(NotImplementedError):                      # (the frame an event falls on runs the section code: device only)
        plan.simulate({"act": 500.0, "buf": 1000.0}, np.zeros((2, 64), dtype=np.float32), mem=mem)
    text = tpar.emit_hip(plan, plan.g.p)
    assert "zpg" in text and "zt_frame(" in text
    # ... where the condition differs from frame to frame the loop itself is the event
    plan, msg = _plan_of_text("k = 0; loop(nb, abs(spl0) > thr[k] ? ( fft(buf + 64 * k, 64); ); k += 1; ); spl0 = buf[0];",
                              init="thr = 500; buf = 1000; nb = 2; nb = 4;")      # (two stores: a count the planner knows as an invariant, not as a number)
    assert plan is not None and plan.stats["loops"] == 0 and plan.stats["events"] == 1, msg
    # a statement that runs in every frame is no event
    plan, msg = _plan_of_text("k = 0; loop(4, buf[pos + k] = spl0; k += 1; ); pos += 1; spl0 = buf[pos - 9];", init="buf = 1000; pos = 50;")
    assert plan is None and "inside a loop" in msg, msg


def test_wrapped_counters_have_a_closed_form(monkeypatch):
    """pos = (pos + K) % N with K, N constant over a block: the state before frame t of a chunk is (pos + t K) % N -- over
    non-negative integers, checked per chunk; anything else (a fractional step, a negative start) takes the serial loop."""
    rng = np.random.default_rng(5)
    x = rng.standard_normal((2, 300)).astype(np.float32)
    for text, cases in (
            ("ring[pos] = spl0; pos = (pos + step) % len; spl0 = ring[pos] * 0.5; spl1 = pos;",       # a ring position
             ({"ring": 100.0, "len": 150.0, "step": 1.0, "pos": 0.0}, {"ring": 100.0, "len": 150.0, "step": 1.0, "pos": 149.0})),
            ("pos = (pos + step) % len; spl0 = spl0 + pos * 0.01; spl1 = pos;",
             ({"len": 37.0, "step": 5.0, "pos": 36.0},
              {"len": 64.0, "step": 3.0, "pos": 70.0},            # starts beyond the length: the first frame sees it as it is
              {"len": 37.0, "step": 1.5, "pos": 2.0},             # a fractional step: the serial loop (za_mod truncates)
              {"len": 37.0, "step": 2.0, "pos": -5.0}))):         # a negative start: the serial loop
        monkeypatch.delenv("ZA_TPAR_NO_MODC", raising=False)
        plan, msg = _plan_of_text(text)
        assert plan is not None and plan.stats["wrapped_counters"] == 1 and plan.stats["serial_loops"] == 0, msg
        monkeypatch.setenv("ZA_TPAR_NO_MODC", "1")
        ref_plan, _ = _plan_of_text(text)
        assert ref_plan.stats["wrapped_counters"] == 0 and ref_plan.stats["serial_loops"] == 1
        for v0 in cases:
            y, va, _ = plan.simulate(dict(v0), x)
            yr, vr, _ = ref_plan.simulate(dict(v0), x)
            assert np.array_equal(y, yr) and va["pos"] == vr["pos"], v0
            assert np.array_equal(plan.mem_after, ref_plan.mem_after)


def test_arms_of_block_constant_conditions_run_under_uniform_branches(monkeypatch):
    """Nodes whose every use is one arm of selects on a single block-constant condition are emitted under a wave-uniform branch
    (tpar._uniform_guards; on for the leaves where it was measured to pay, ZA_TPAR_BRANCHES=1 forces it)."""
    from zajit import tpar
    monkeypatch.setenv("ZA_TPAR_BRANCHES", "1")
    plan, msg = _plan_of_text("a = spl0 * 2; mode ? ( y = exp(a) * sin(a); ) : ( y = sqrt(abs(a)) + 1; ); z += (y - z) * 0.1; spl0 = z; spl1 = a;")
    assert plan is not None, msg
    arms = {}
    for i, (cn, arm) in plan.node_guard.items():
        arms.setdefault(arm, []).append(plan.g.nodes[i].op)
        assert cn.kind == "inv" and cn.name == "mode"
    assert "exp" in arms[True] and "sin" in arms[True] and "sqrt" in arms[False]
    text = tpar.emit_hip(plan, plan.g.p)
    assert "const bool zg" in text and "if (zg" in text and "if (!zg" in text
    # `a` feeds both arms and an output: always computed; the recurrence's own nodes too
    always = [n for n in plan.g.nodes if n.kind == "op" and n.op == "*" and n.i not in plan.node_guard]
    assert always
    monkeypatch.setenv("ZA_TPAR_BRANCHES", "0")
    plan0, _ = _plan_of_text("a = spl0 * 2; mode ? ( y = exp(a) * sin(a); ) : ( y = sqrt(abs(a)) + 1; ); z += (y - z) * 0.1; spl0 = z; spl1 = a;")
    assert plan0.node_guard == {}


def test_unsupported_scripts_keep_the_generic_kernel_only():
    from zajit import program, tpar
    for sample, why in (
            ("buf[wp & 63] = spl0; wp += 1; fft(buf, 64); spl0 = buf[3];", "builtin fft"),      # (in every frame: no event to cut out)
            ("acc = 0; k = 0; loop(4, acc = acc * 0.5 + st; k += 1; ); st = acc + spl0; spl0 = st;", "runs through a loop"),
            ("k = 0; loop(4, buf[pos + k] = spl0; k += 1; ); pos += 1; spl0 = buf[pos - 9];", "inside a loop"),
            ("k = 0; loop(3, j = 0; loop(2, tab[k * 2 + j] += spl0; j += 1; ); k += 1; );", "nested loop"),
            ("k = 0; loop(3, r = rand(1); k += 1; ); spl0 = r;", "rand() inside a loop")):
        plan, msg = _plan_of_text(sample, init="buf = 1000; ring = 3000; tab = 5000; wp = 100; pos = 50;")
        assert plan is None and why in msg, (sample, msg)


def test_two_writes_that_touch_one_cell_hand_the_chunk_back():
    """`buf[pos] = a; buf[pos + 1] = b; pos += 1` -- two store sites whose spans overlap inside a chunk. Written
    site by site a later frame's first store would land after an earlier frame's second one; the spans of all writes of a chunk
    are compared pairwise and the chunk goes to the serial code."""
    from zajit import tpar
    plan, msg = _plan_of_text("buf[pos] = spl0; buf[pos + 1] = -1000 - spl0; pos += 1; spl1 = buf[pos - 20];", init="buf = 1000; pos = 50;")
    assert plan is not None, msg
    x = np.zeros((2, 200), dtype=np.float32); x[0] = np.arange(200) * 0.001
    with pytest.raises(tpar.TparAbort) as ei:
        plan.simulate({"buf": 1000.0, "pos": 50.0}, x)
    assert ei.value.f0 == 0 and "touch one cell" in ei.value.why
    # far enough apart the same two writes are two delay lines
    plan, msg = _plan_of_text("buf[pos] = spl0; buf[pos + 4096] = -spl0; pos += 1; spl1 = buf[pos - 20] + (pos - 3)[buf + 4096];",
                              init="buf = 1000; pos = 50;")
    y, va, _ = plan.simulate({"buf": 1000.0, "pos": 50.0}, x)
    assert va["pos"] == 250.0 and plan.mem_after[1000 + 50 + 4096 + 7] == -x[0, 7]
    assert y[1, 100] == np.float32(np.float64(x[0, 81]) - np.float64(x[0, 98]))


def test_writes_into_one_delay_line_move_in_step():
    """Round 4. Spectral/Alias: its six intdelay() lines all live at mem[0] (the instance variable `buf` is never set, so it does
    not tell buffers apart: FrameGraph.never_assigned). Several writes into one delay line per frame are fine while they move
    in step -- the same cell in the same frame: a read takes the last write in front of it in (frame, program) order, the late
    stores go out in program order; writes that meet any other way hand the chunk back. Ring positions written with a mask,
    `pos = (pos + 1) & 2047` under a block-constant condition, are wrapped counters."""
    plan, _ = _plan("Alias")
    assert plan.stats["guards"] == 1 and plan.stats["delay_writes"] == 9 and plan.stats["sparse_writes"] == 3
    assert plan.stats["wrapped_counters"] == 6
    assert len({s.region for s in plan.stores if s.mode == "late"}) == 1
    for case in ("Alias_default", "Alias_alt"):
        g = load_golden(case)
        names = [str(s) for s in g["var_names"]]
        v0 = {n: (0.0 if np.isnan(v) else float(v)) for n, v in zip(names, g["vars_prepared"])}
        mem0, high0 = _prepared_arena("Alias", g)
        y, va, _ = plan.simulate(v0, golden_input(g), sliders=g["sliders"], srate=float(g["srate"]), mem=mem0)
        assert np.abs(y.astype(np.float64) - g["out"]).max() <= AUDIO_EPS
        assert_state_close(names, [va.get(n, 0.0) for n in names], g["vars"], what=f"{case} vars")
        want = np.zeros(len(mem0)); want[g["mem_idx"]] = g["mem_val"]
        assert np.abs(plan.mem_after - want).max() <= SCALAR_EPS
        assert max(plan.mem_high_after, high0) == int(g["mem_high"])
    # read-modify-write of the cell just written, and a second line through the same cells one frame apart (not in step)
    rng = np.random.default_rng(3)
    x = rng.standard_normal((2, 200)).astype(np.float32)
    plan, msg = _plan_of_text("ring[wp] = spl0; ring[wp] += spl1; wp += 1; spl0 = ring[wp - 7];", init="ring = 3000; wp = 100;")
    assert plan is not None, msg
    y, va, _ = plan.simulate({"ring": 3000.0, "wp": 100.0}, x)
    want = np.zeros(200); want[6:] = (x[0].astype(np.float64) + x[1])[:-6]
    assert np.array_equal(y[0], want.astype(np.float32)) and va["wp"] == 300.0
    from zajit import tpar
    plan, msg = _plan_of_text("ring[wp] = spl0; ring[wp + 1] = spl1; wp += 1; spl0 = ring[wp - 7];", init="ring = 3000; wp = 100;")
    with pytest.raises(tpar.TparAbort) as ei:
        plan.simulate({"ring": 3000.0, "wp": 100.0}, x)
    assert "touch one cell" in ei.value.why


def test_cells_of_a_loop_under_a_condition_stay_put_where_it_is_false(monkeypatch):
    """Round 4 (a latent error of round 3's loops): `c ? ( loop(n, z[k] = ...) )` with c a per-frame value. The loop's per-trip
    cells leave the walk's environment when the loop ends, i.e. before the conditional merges what its arms assigned, so the path
    condition the loop stands under has to be applied to them there. (split_events hides the case for conditionals it takes as
    events; one with an else arm, or a condition that is no plain expression, reaches the loop.) A chunk none of whose frames
    takes the branch skips the loop altogether."""
    from zajit import tpar
    monkeypatch.setenv("ZA_TPAR_NO_EVENTS", "1")
    text = "c = spl1 > 0; s = 0; c ? ( k = 0; loop(NB, z[k] = z[k] * 0.5 + spl0; s += z[k]; k += 1; ); ); spl0 = s;"
    plan, msg = _plan_of_text(text, init="NB = 3; z = 100;")
    assert plan is not None and plan.stats["loops"] == 1 and plan.stats["trip_cells_stored"] == 1, msg
    rng = np.random.default_rng(1)
    x = rng.standard_normal((2, 200)).astype(np.float32)
    x[1, 64:128] = -1.0                     # one whole chunk with the condition false
    y, va, _ = plan.simulate({"NB": 3.0, "z": 100.0}, x)
    zz, ref = np.zeros(3), np.zeros(200)
    for t in range(200):
        acc = 0.0
        if x[1, t] > 0:
            for k in range(3):
                zz[k] = zz[k] * 0.5 + np.float64(x[0, t]); acc += zz[k]
        ref[t] = acc
    assert np.array_equal(y[0], ref.astype(np.float32))
    assert np.allclose(plan.mem_after[100:103], zz, rtol=0, atol=1e-15) and plan.mem_high_after == 103
    text = tpar.emit_hip(plan, plan.g.p)
    assert "if (__ballot(valid && za_truthy(" in text        # the skip


def test_feedback_through_a_delay_line_cuts_the_chunk():
    """Round 4: y[t] = x[t] + g y[t - D] through a ring. While D >= 64 no frame of a chunk reads what the chunk writes and the loop
    closes over memory; a shorter delay ends the chunk before the first frame that would (the next segment starts there);
    below 16 frames the serial code takes over."""
    from zajit import tpar
    text = "y = spl0 + 0.5 * ring[rp]; ring[wp] = y; wp = (wp + 1) & 127; rp = (rp + 1) & 127; spl0 = y; spl1 = rp;"
    plan, msg = _plan_of_text(text, init="ring = 1000;")
    assert plan is not None and [sorted(ld.fb) for ld in plan.loads] == [[0]], msg
    kinds = [it[0] for it in plan.top.items]
    assert kinds.count("cut") == 1 and "site" not in kinds[:kinds.index("cut")]
    rng = np.random.default_rng(1)
    x = rng.standard_normal((2, 500)).astype(np.float32)
    for D, cuts in ((100, 0), (64, 0), (40, 12), (17, 29)):
        y, va, _ = plan.simulate({"ring": 1000.0, "wp": 5.0, "rp": float((5 - D) & 127)}, x)
        ref = np.zeros(500)
        for t in range(500):
            ref[t] = np.float64(x[0, t]) + 0.5 * (ref[t - D] if t >= D else 0.0)
        assert np.array_equal(y[0], ref.astype(np.float32)) and plan.fb_cuts == cuts, D
        assert va["wp"] == float((5 + 500) & 127)
    with pytest.raises(tpar.TparAbort) as ei:
        plan.simulate({"ring": 1000.0, "wp": 5.0, "rp": float((5 - 10) & 127)}, x)
    assert "shorter than 16" in ei.value.why
    text = tpar.emit_hip(plan, plan.g.p)
    assert "zt_fbc" in text


def test_what_used_to_be_unsupported_now_plans():
    """@block, scripts that raise slider masks, uniform loops with per-trip cells and gathers, conditional stores into buffers
    @sample never reads: round 3 took these blockers out."""
    from zajit import program, tpar
    plan, msg = tpar.try_plan(program.analyse_file(FIXTURES / "delaytaps.jsfx"), 2)
    assert plan is not None and plan.stats["loops"] == 1 and plan.stats["gathers"] == 4 and plan.stats["early_writes"] == 2, msg
    plan, msg = tpar.try_plan(program.analyse_file(FIXTURES / "slidewrite.jsfx"), 2)
    assert plan is not None and plan.has_block and plan.has_pending, msg
    plan, msg = _plan_of_text("k = 0; s = 0; while (k < NB) ( z[k] = z[k] * 0.5 + spl0 * g[k]; s += z[k]; k += 1; ); spl0 = s;",
                              init="NB = 5; z = 100; g = 200;")
    assert plan is not None, msg
    assert plan.stats["trip_cells"] == 2 and plan.stats["trip_cells_stored"] == 1
    assert "k" not in plan.st and "k" not in plan.holdvars          # a counter set before it is read: no state at all
    plan, msg = _plan_of_text("s = spl0 + spl1; cnt += 1; cnt >= 100 ? ( hist[hp] = s; hp += 1; hp >= 32 ? hp = 0; cnt = 0; tmp = s; );",
                              init="hist = 400;")
    assert plan is not None, msg
    assert plan.stats["sparse_writes"] == 1 and plan.holdvars == ["tmp"]


@pytest.mark.parametrize("case", TPAR_FIXTURES + [f"{l}_{c}" for l in TPAR_CATALOG for c in ("default", "alt")] + ["DDT_default", "DDT_far_extreme"])
def test_staged_algorithm_matches_reference_vm(case):
    leaf = leaf_of(case)
    plan, _ = _plan(leaf)
    g = load_golden(case)
    names = [str(s) for s in g["var_names"]]
    v0 = {n: (0.0 if np.isnan(v) else float(v)) for n, v in zip(names, g["vars_prepared"])}
    x = golden_input(g)
    mem0, high0 = _prepared_arena(leaf, g) if plan.stats["mem_cells"] + plan.stats["delay_writes"] + plan.stats["trip_cells"] else (None, 0)
    y, va, _ = plan.simulate(v0, x, sliders=g["sliders"], srate=float(g["srate"]), mem=mem0)
    assert np.abs(y.astype(np.float64) - g["out"]).max() <= AUDIO_EPS
    assert_state_close(names, [va.get(n, 0.0) for n in names], g["vars"], what=f"{case} vars")
    if mem0 is not None:
        want = np.zeros(len(mem0)); want[g["mem_idx"]] = g["mem_val"]
        assert np.abs(plan.mem_after - want).max() <= SCALAR_EPS
        assert max(plan.mem_high_after, high0) == int(g["mem_high"])
    if case == "fx_randkat_default":            # 6021 draws: nine generations of the generator, conditional bursts included
        assert va["draws"] == 6021.0 and plan.mt_after[1] == 6021 - 9 * 624


def test_rand_stream_continues_across_launches():
    plan, _ = _plan("fx_randkat")
    g = load_golden("fx_randkat_default")
    names = [str(s) for s in g["var_names"]]
    v0 = {n: (0.0 if np.isnan(v) else float(v)) for n, v in zip(names, g["vars_prepared"])}
    x = golden_input(g)
    ys, va, sp, mt = [], v0, None, None
    for lo, hi in ((0, 100), (100, 101), (101, 1500), (1500, 3000)):
        y, va, sp = plan.simulate(va, x[:, lo:hi], sliders=g["sliders"], spl0=sp, mt=mt)
        mt = plan.mt_after
        ys.append(y)
    assert np.abs(np.concatenate(ys, axis=1).astype(np.float64) - g["out"]).max() <= AUDIO_EPS
    assert_state_close(names, [va.get(n, 0.0) for n in names], g["vars"], what="vars")


def test_staged_algorithm_is_independent_of_launch_boundaries():
    """Three launches of awkward lengths (chunk remainders 1, 63, 0) leave the same audio and state as one launch."""
    plan, _ = _plan("fx_dynkat")
    g = load_golden("fx_dynkat_default")
    names = [str(s) for s in g["var_names"]]
    v0 = {n: (0.0 if np.isnan(v) else float(v)) for n, v in zip(names, g["vars_prepared"])}
    x = golden_input(g)[:, :1000]
    y1, va1, sp1 = plan.simulate(v0, x, sliders=g["sliders"])
    ys, va, sp = [], v0, None
    for lo, hi in ((0, 65), (65, 65 + 447), (65 + 447, 1000)):
        y, va, sp = plan.simulate(va, x[:, lo:hi], sliders=g["sliders"], spl0=sp)
        ys.append(y)
    y3 = np.concatenate(ys, axis=1)
    assert np.abs(y3.astype(np.float64) - y1).max() <= 1e-7
    for n in names:
        assert abs(va.get(n, 0.0) - va1.get(n, 0.0)) <= 1e-9, n


# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("case", TPAR_FIXTURES + TPAR_R4_FIXTURES + TPAR_ABORTS
                         + [f"{l}_{c}" for l in TPAR_CATALOG + TPAR_BLOCK_CATALOG for c in ("default", "alt")]
                         + ["Alias_default", "Alias_alt", "Contour_default"])
def test_tpar_kernel_matches_reference_vm(case):
    import zabatch
    leaf = leaf_of(case)
    if not zabatch.module_path(leaf).exists():
        pytest.skip(f"module for {leaf} not built")
    assert "unsupported" not in zabatch.leaf_meta(leaf)["tpar"], zabatch.leaf_meta(leaf)["tpar"]
    g = load_golden(case)
    n = 5 if int(g["mem_high"]) <= (1 << 22) else 2
    x = np.repeat(golden_input(g)[None], n, axis=0)
    res = {}
    for label, path in (("tpar", zabatch.ZAB_PATH_FAST), ("generic", zabatch.ZAB_PATH_GENERIC)):
        with zabatch.Engine(leaf, n, srate=float(g["srate"]), path=path, mem_cap=max(65536, int(g["mem_high"]) + 64)) as e:
            e.set_sliders(g["sliders"])
            e.prepare()
            names = e.var_names()
            y = e.process_host(x, block=int(g["block"]))
            assert e.used_fast_path() == (label == "tpar")
            if label == "tpar":
                assert e.last_kernel_name().endswith("_tpar")
            high = e.mem_high()
            mem = e.read_mem(0, int(g["mem_high"])) if int(g["mem_high"]) else None
            res[label] = (y, e.read_vars(), high, mem)
    for label, (y, v, high, mem) in res.items():
        err = np.abs(y.astype(np.float64) - g["out"].astype(np.float64)[None]).max()
        print(f"{case} [{label}]: null test max {dbfs(err):.1f} dBFS")
        assert err <= AUDIO_EPS, label
        for i in (0, n - 1):
            assert_state_close(names, v[i], g["vars"], what=f"{case} {label} vars[{i}]")
        assert (high == int(g["mem_high"])).all(), (label, high)
        if mem is not None:
            want = np.zeros(int(g["mem_high"])); want[g["mem_idx"]] = g["mem_val"]
            assert np.abs(mem - want[None]).max() <= SCALAR_EPS, label
    assert np.array_equal(res["tpar"][0][0], res["tpar"][0][n - 1])


@pytest.mark.gpu
@pytest.mark.parametrize("leaf", ["fx_dynkat", "fx_randkat", "fx_ringkat", "fx_ringabort", "fx_delaytaps", "fx_statekat"] + TPAR_CATALOG + TPAR_BLOCK_CATALOG
                         + ["CMD", "DOT"] + TPAR_EVENT_LEAVES + ["fx_voicekat"] + TPAR_R4_CATALOG
                         + ["Contour+IR", "TextureXY+IR", "Texture+IR"])      # (a texture in file slot 0: grains spawn, voices run)
def test_tpar_kernel_tracks_generic_kernel_over_a_long_run_and_across_launches(leaf):
    """One second of audio, distinct noise and sliders per instance: the time-parallel kernel in ragged launches (lengths with
    chunk remainders 1, 63, 0 and a single frame) against the generic kernel in one launch -- audio within the reference's
    1e-5, final state within 1e-8 (affine recurrences are re-associated, nothing else differs)."""
    import zabatch
    from zajit import noise
    loaded, leaf = leaf.endswith("+IR"), leaf.split("+")[0]
    if not zabatch.module_path(leaf).exists():
        pytest.skip(f"module for {leaf} not built")
    meta = zabatch.leaf_meta(leaf)
    n, frames = 6, 48000
    # (an impulse response in file slot 0: 0.25 s of decaying stereo noise -- the convolver's partitions then run, every 2048 frames)
    ir = (noise.white_noise([321], 12000)[0].T * np.exp(-np.arange(12000) / 3000.0)[:, None]).reshape(-1).astype(np.float64)
    if leaf in ("Contour", "Texture", "TextureXY"):
        n = 3             # (arenas of tens of millions of cells)
    if "gmem" in meta["features"]:
        n = 1             # instances of one engine share its gmem segment: what one reads depends on when the others wrote
    nch = int(meta["nch"])
    x = noise.white_noise(range(n), frames, channels=nch)
    x[:, :, 20000:26000] *= 0.01                                 # a quiet stretch: gates close, holds run out
    rows = np.tile(np.array(meta["default_sliders"], dtype=np.float64), (n, 1))
    for k, sd in meta["sliders"].items():                        # spread every continuous slider over the instances
        if not sd["is_choice"] and not sd["is_string"] and sd["max"] > sd["min"]:
            k = int(k)
            rows[:, k] = rows[:, k] + (np.arange(n) / n - 0.4) * 0.2 * (sd["max"] - sd["min"])
            rows[:, k] = np.clip(rows[:, k], sd["min"], sd["max"])
    cap = {"SOMA": 1 << 18, "Alias": 1 << 19, "PsychoConvolver": 1 << 22,
           "Contour": 1 << 24, "Texture": 1 << 25, "TextureXY": 1 << 25}.get(leaf, 1 << 16)
    cuts = [0, 1, 66, 66 + 63, 4096 + 129, 30000, frames]
    # (a script with @block sees where a launch starts -- every launch begins a block -- so both engines get the same launches)
    ref_cuts = cuts if meta["has"]["block"] else [0, frames]
    with zabatch.Engine(leaf, n, path=zabatch.ZAB_PATH_GENERIC, mem_cap=cap) as e:
        if loaded:
            e.file_slot_set(0, ir, channels=2, sample_rate=48000.0)
        e.set_sliders(rows); e.prepare()
        want = np.concatenate([e.process_host(x[:, :, a:b], block=512) for a, b in zip(ref_cuts[:-1], ref_cuts[1:])], axis=2)
        want_v = e.read_vars(); names = e.var_names()
        want_ck = e.checkpoint()                                 # (arena up to the write high-water mark, marks, rand() state)
    with zabatch.Engine(leaf, n, path=zabatch.ZAB_PATH_FAST, mem_cap=cap) as e:
        if loaded:
            e.file_slot_set(0, ir, channels=2, sample_rate=48000.0)
        e.set_sliders(rows); e.prepare()
        got, handed = [], 0
        for a, b in zip(cuts[:-1], cuts[1:]):
            got.append(e.process_host(x[:, :, a:b], block=512))
            handed += e.handback()[0]
        got = np.concatenate(got, axis=2)
        assert e.used_fast_path()
        # the catalog's leaves keep every chunk of this run on the time-parallel path (zab_handback_stats); fixtures written to
        # break the lowering's run-time conditions do not
        print(f"{leaf}: instance-launches handed to the serial tail: {handed}")
        assert handed == 0 or leaf.startswith("fx_") or (loaded and leaf in TPAR_R4_CATALOG), (leaf, handed)
        if leaf == "fx_ringabort":
            assert handed > 0
        got_v = e.read_vars()
        got_ck = e.checkpoint()
    assert np.array_equal(got_ck["mti"], want_ck["mti"]) and np.array_equal(got_ck["mt"], want_ck["mt"])     # rand() state
    assert np.array_equal(got_ck["mem_high"], want_ck["mem_high"])
    if want_ck["mem_data"].size:
        assert np.abs(got_ck["mem_data"] - want_ck["mem_data"]).max() <= SCALAR_EPS
    err = np.abs(got.astype(np.float64) - want.astype(np.float64)).max()
    print(f"{leaf}: tpar vs generic over {frames} frames: {dbfs(err):.1f} dBFS")
    assert err <= AUDIO_EPS
    for i in range(n):
        assert_state_close(names, got_v[i], want_v[i], what=f"{leaf} vars[{i}]", rel=True)


@pytest.mark.gpu
@pytest.mark.parametrize("leaf", ["fx_evtkat", "fx_evtkat2", "fx_guardkat", "fx_stft", "fx_convkat", "ERBTilt", "TSEQ"])
@pytest.mark.parametrize("block", [37, 64, 65, 1000])
def test_odd_block_sizes_and_short_launches(leaf, block):
    """Blocks that are no multiple of a chunk, shorter than one, or longer than the launch; launches of 1, 63 and 64 frames:
    segment ends, event frames and block boundaries fall on every lane position. Both kernels get the same launches and blocks
    (a script with @block sees them), audio within 1e-5, state within 1e-8."""
    import zabatch
    from zajit import noise
    if not zabatch.module_path(leaf).exists():
        pytest.skip(f"module for {leaf} not built")
    meta = zabatch.leaf_meta(leaf)
    n, frames = 3, 2300
    nch = int(meta["nch"])
    x = noise.white_noise(range(n), frames, channels=nch)
    cuts = [0, 1, 64, 127, 128 + 64, 1000, frames]
    res = {}
    for label, path in (("fast", zabatch.ZAB_PATH_FAST), ("generic", zabatch.ZAB_PATH_GENERIC)):
        with zabatch.Engine(leaf, n, path=path, max_block=block) as e:
            e.set_sliders(meta["default_sliders"]); e.prepare()
            y = np.concatenate([e.process_host(x[:, :, a:b], block=block) for a, b in zip(cuts[:-1], cuts[1:])], axis=2)
            res[label] = (y, e.read_vars(), e.var_names())
            if label == "fast":
                assert e.used_fast_path()
    err = np.abs(res["fast"][0].astype(np.float64) - res["generic"][0].astype(np.float64)).max()
    assert err <= AUDIO_EPS, (leaf, block, err)
    for i in range(n):
        assert_state_close(res["fast"][2], res["fast"][1][i], res["generic"][1][i], what=f"{leaf} block {block} vars[{i}]")


@pytest.mark.gpu
@pytest.mark.parametrize("period", [2, 3, 16, 17, 64, 65, 129])
def test_events_of_every_density(period):
    """fx_evtkat / fx_evtkat2 with their event every `period` frames: thicker than one frame in sixteen the kernel hands the rest of
    the launch to the serial tail, at 64 / 65 / 129 the event walks through the lanes of a chunk or sits on its first one."""
    import zabatch
    from zajit import noise
    for leaf in ("fx_evtkat", "fx_evtkat2"):
        if not zabatch.module_path(leaf).exists():
            pytest.skip(f"module for {leaf} not built")
        meta = zabatch.leaf_meta(leaf)
        n, frames = 3, 3000
        x = noise.white_noise(range(n), frames)
        rows = np.tile(np.array(meta["default_sliders"], dtype=np.float64), (n, 1))
        rows[:, 0] = period
        rows[1, 0] = period + 1                      # (instances of a batch need not agree on where their events fall)
        res = {}
        for label, path in (("fast", zabatch.ZAB_PATH_FAST), ("generic", zabatch.ZAB_PATH_GENERIC)):
            with zabatch.Engine(leaf, n, path=path) as e:
                e.set_sliders(rows); e.prepare()
                y = np.concatenate([e.process_host(x[:, :, :1700], block=500), e.process_host(x[:, :, 1700:], block=500)], axis=2)
                res[label] = (y, e.read_vars(), e.var_names(), e.read_mem(1000, 8))
        assert np.abs(res["fast"][0].astype(np.float64) - res["generic"][0]).max() <= AUDIO_EPS, (leaf, period)
        assert np.abs(res["fast"][3] - res["generic"][3]).max() <= SCALAR_EPS
        for i in range(n):
            assert_state_close(res["fast"][2], res["fast"][1][i], res["generic"][1][i], what=f"{leaf} period {period} vars[{i}]")


@pytest.mark.gpu
def test_slider_change_between_launches_runs_at_slider_before_the_tpar_kernel():
    import zabatch
    from zajit import noise
    leaf = "fx_dynkat"
    meta = zabatch.leaf_meta(leaf)
    n, frames = 3, 1500
    x = noise.white_noise(range(n), 2 * frames)
    rows = np.tile(np.array(meta["default_sliders"], dtype=np.float64), (n, 1))
    rows2 = rows.copy(); rows2[1, 0] = 0.9; rows2[2, 4] = 4000.0
    out = {}
    for label, path in (("tpar", zabatch.ZAB_PATH_FAST), ("generic", zabatch.ZAB_PATH_GENERIC)):
        with zabatch.Engine(leaf, n, path=path) as e:
            e.set_sliders(rows); e.prepare()
            a = e.process_host(x[:, :, :frames], block=256)
            e.set_sliders(rows2)
            b = e.process_host(x[:, :, frames:], block=256)
            out[label] = (np.concatenate([a, b], axis=2), e.read_vars())
    assert np.abs(out["tpar"][0].astype(np.float64) - out["generic"][0]).max() <= AUDIO_EPS
    assert np.abs(out["tpar"][1] - out["generic"][1]).max() <= SCALAR_EPS
    assert not np.array_equal(out["tpar"][0][0, :, frames:], out["tpar"][0][1, :, frames:])


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["fx_dynkat_default", "fx_dynkat_hot"])
def test_tpar_kernel_serial_fallback_of_switched_recurrences(case):
    """fx_dynkat_s1 = the same script built with an iteration budget of one (-DZT_SPEC_MAX=1): chunks whose condition pattern
    is not right at the first guess run the serial loop instead. Same golden vectors."""
    import zabatch
    leaf = "fx_dynkat_s1"
    if not zabatch.module_path(leaf).exists():
        pytest.skip(f"module for {leaf} not built")
    g = load_golden(case)
    x = np.repeat(golden_input(g)[None], 3, axis=0)
    with zabatch.Engine(leaf, 3, srate=float(g["srate"]), path=zabatch.ZAB_PATH_FAST) as e:
        e.set_sliders(g["sliders"]); e.prepare()
        names = e.var_names()
        y = e.process_host(x, block=int(g["block"]))
        v = e.read_vars()
    assert np.abs(y.astype(np.float64) - g["out"].astype(np.float64)[None]).max() <= AUDIO_EPS
    assert_state_close(names, v[2], g["vars"], what=f"{case} vars")


def test_generated_module_text_does_not_depend_on_the_process(tmp_path):
    """Two builds of one script must write the same module text: a time-parallel plan whose node order followed the iteration order
    of a set of state names came out differently in every process (Python hashes strings per process), so TextureXY, BedRock,
    EasyExpander and a fixture were recompiled by every build() -- and a hazard of the device compiler that depends on the exact text
    (DESIGN.md "Compiler hazards") could come and go between builds. Same text under different hash seeds."""
    import subprocess, sys
    prog = (
        "import sys, hashlib\n"
        f"sys.path[:0] = [{str(ROOT / 'zorakaudio-experimental-plugins_amd')!r}, {str(ROOT)!r}]\n"
        "from zajit import build as zb, program, codegen\n"
        "from pathlib import Path\n"
        "for fx in sys.argv[1:]:\n"
        "    p = program.analyse_file(Path(fx)); p.name = 'fx_' + Path(fx).stem\n"
        "    print(hashlib.sha1(zb.module_source(codegen.make_unit(p)).encode()).hexdigest())\n")
    files = [str(FIXTURES / f"{n}.jsfx") for n in ("statekat", "dynkat", "voicekat", "stft", "ringkat")]
    outs = []
    for seed in ("1", "2", "777"):
        import os
        r = subprocess.run([sys.executable, "-c", prog] + files, capture_output=True, text=True, env={**os.environ, "PYTHONHASHSEED": seed})
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(r.stdout.split())
    assert outs[0] == outs[1] == outs[2] and len(outs[0]) == len(files)
