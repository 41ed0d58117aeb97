"""zajit front end vs pins captured from the reference's own front end (tests/golden/frontend.json), plus parser
behaviours the reference's compile-smoke tests exercise (scripts/run_dsp-jsfx_*tests.py)."""
import hashlib
import json
from pathlib import Path

import numpy as np
import pytest

from conftest import GOLDEN

REF_PLUGINS = Path("/root/reference/plugins")
PINS = json.loads((GOLDEN / "frontend.json").read_text())


def _leaf_sources():
    return {p.parent.parent.name: p for p in sorted(REF_PLUGINS.glob("*/*/src/*.jsfx"))}


@pytest.mark.skipif(not REF_PLUGINS.exists(), reason="leaf sources live in the reference tree (dev container only)")
@pytest.mark.parametrize("leaf", sorted(PINS))
def test_front_end_matches_reference(leaf):
    from zajit import program
    src = _leaf_sources()[leaf]
    prog = program.analyse_file(src)
    pin = PINS[leaf]
    assert len(prog.vars) == pin["nvars"]
    sha = hashlib.sha1(json.dumps(sorted(prog.vars.items())).encode()).hexdigest()
    assert sha == pin["vars_sha1"], "vars[] name->index table differs from the reference's DSPJSFX_VARS"
    assert len(prog.fns) == pin["nfns"]
    assert hashlib.sha1(json.dumps(sorted(prog.fns.keys())).encode()).hexdigest() == pin["fns_sha1"]
    for k in ("inputs", "outputs", "process", "max_read", "max_write"):
        assert prog.io[k] == pin["io"][k], k
    assert prog.memtop == pin["memtop"]
    assert {k: prog.has(k) for k in ("init", "slider", "block", "sample")} == pin["sections"]


def test_built_modules_carry_the_pinned_var_tables():
    """The metadata that travels with the built modules must still agree with the reference pins."""
    import zabatch
    for leaf in ("DDT", "DPT", "ADS"):
        if not zabatch.module_path(leaf).exists():
            pytest.skip("modules not built")
        meta = zabatch.leaf_meta(leaf)
        assert meta["vars_sha1"] == PINS[leaf]["vars_sha1"]
        assert meta["nvars"] == max(1, PINS[leaf]["nvars"])


def _parse(code):
    from zajit import syntax
    return syntax.parse_section(code)


def test_parser_precedence_and_forms():
    from zajit import syntax as S
    (n,) = _parse("a = b + c * d ^ e;")
    assert isinstance(n, S.Assign) and n.value.op == "+" and n.value.r.op == "*" and n.value.r.r.op == "^"
    (n,) = _parse("x = c ? 1;")                      # implicit else 0
    assert isinstance(n.value, S.Cond) and n.value.els.value == 0.0
    (n,) = _parse("a | b || c & d == e")             # '|' binds like '||', '&' like '=='
    assert n.op == "||" and n.l.op == "|" and n.r.op == "=="
    (n,) = _parse("loop(4, a += 1; b += a;)")
    assert isinstance(n, S.Loop) and isinstance(n.body, S.Seq) and len(n.body.items) == 2
    (n,) = _parse("wrapped\n  || other\n")           # newline-led infix continuation
    assert isinstance(n, S.Binary) and n.op == "||"
    two = _parse("x = a\n+ b;")                      # '+' on a new line starts a new statement (unary)
    assert len(two) == 2
    (n,) = _parse("m[3] = u.next.bank;")
    assert isinstance(n.target, S.Index) and n.value.name == "u.next.bank"
    (f,) = _parse("function f(a b) local(t, u) instance(s) ( t = a; t*b );")
    assert f.params == ["a", "b"] and f.locals == ["t", "u"] and f.instances == ["s"]
    (n,) = _parse("slider(3) = 5;")
    assert isinstance(n.target, S.Call) and n.target.fn == "slider"
    with pytest.raises(SyntaxError):
        _parse("a + b = 3;")
    with pytest.raises(SyntaxError):
        _parse("x = 'unterminated")


def test_function_lowering_names_and_locals():
    from zajit import program
    text = """desc:t
@init
function lp(x) instance(s) local(tmp) ( tmp = x; s += 0.5*(tmp - s); s );
function two(x) ( this.a.lp(x) + this.b.lp(x) );
@sample
spl0 = f1.two(spl0);
spl1 = lp(spl1);
"""
    p = program.analyse(text)
    assert "__fn__sample__two__ns__f1" in p.fns
    assert "__fn__sample__lp__ns__f1_x2E_a" in p.fns and "__fn__sample__lp__ns__f1_x2E_b" in p.fns
    assert "__fn__sample__lp__ns__lp" in p.fns            # bare call of an instance() function: namespace = its name
    for v in ("f1.a.s", "f1.b.s", "lp.s", "__fnlocal__sample__lp__tmp"):
        assert v in p.vars, v
    assert p.io["inputs"] == 2 and p.io["outputs"] == 2


def test_named_constants_become_literals_outside_init():
    """A script's named constants (one unconditional top-level `name = <numbers>` in @init, no other store anywhere) are read as
    literals by @slider / @block / @sample and the functions specialised for them; @init and the variable table stay as they
    were (zajit/program.py named_constants)."""
    from zajit import program, syntax as S
    text = """desc:t
slider1:gain_db=0<-60,12,1>gain
@init
STRIDE = 8; N = STRIDE * 4 - 2; HALF = N / 4;
twice = 1; twice = 2;
cond ? guarded = 3;
compound = 1; compound += 1;
outarg = 0;
late = early_read + 1; early_read = 5;
gain_db = 1;
gfx_w = 640;
function scale(x, N) ( x * N * STRIDE );
function peek() ( file_var(0, outarg); N );
base = 100;
@slider
k = N;
@sample
i = 0; loop(N, base[i * STRIDE] = scale(spl0, 3) + HALF; i += 1;);
peek(); spl1 = twice + guarded + compound + outarg + early_read + gain_db + gfx_w;
"""
    p = program.analyse(text)
    prog, fns = p.sections, p.fns
    texts = {sec: repr(prog[sec]) for sec in prog}
    # established: STRIDE = 8, N = 30, base = 100, early_read = 5 (its read in @init comes first and sees 0, untouched)
    assert "Var(name='N')" not in texts["slider"] and "Num(value=30.0)" in texts["slider"]
    for nm in ("STRIDE", "N", "base", "early_read"):
        assert f"Var(name='{nm}')" not in texts["sample"], nm
        assert f"Var(name='{nm}')" in texts["init"], nm            # @init still stores (and reads) the table cells
        assert nm in p.vars
    assert "Num(value=100.0)" in texts["sample"] and "Num(value=5.0)" in texts["sample"]
    # only whole numbers become literals (addresses, counts, enumerations): HALF = 7.5 stays a variable the kernels keep in a register
    assert "Var(name='HALF')" in texts["sample"] and "Num(value=7.5)" not in texts["sample"]
    # not constants: two stores, a store under a condition, a compound assignment, a builtin's output argument, a slider alias,
    # a host variable
    for nm in ("twice", "guarded", "compound", "outarg", "gain_db", "gfx_w"):
        assert f"Var(name='{nm}')" in texts["sample"], nm
    # a parameter shadows the global of the same name inside its function; the global STRIDE next to it is folded
    body = repr(fns["__fn__sample__scale"].body)
    assert "Var(name='N')" in body and "Var(name='STRIDE')" not in body and "Num(value=8.0)" in body
    assert "Num(value=30.0)" in repr(fns["__fn__sample__peek"].body)
    # `late` was computed from a variable that was still 0 when @init read it: not derivable, stays a variable
    assert program.named_constants({"init": prog["init"], "sample": []}, {}, {}).get("late") is None


def test_slider_declarations_and_quantiser():
    from zajit import sliders
    text = "slider1:30<0,100,1:sqr>Distance\nslider5:2<0,4,1{Eco,Moderate,High}>Quality\nslider7:0<-12,12,0.1:log>Out\nslider3:thr=-40<-80,0,0.1>-Hidden\n"
    d = sliders.parse_slider_decls(text)
    assert d[0].vmax == 100.0 and d[0].shape == "sqr" and d[4].is_choice and d[4].choices[2] == "High"
    assert d[2].var_name == "thr" and d[2].hidden
    row = sliders.default_slider_values(d)
    assert row[0] == 30.0 and row[4] == 2.0
    # float32 step 0.1 => -12 + 120 * 0.100000001490116 (what the reference host pushes for "0")
    assert abs(row[6] - 1.7881393432617188e-07) < 1e-20
    assert d[0].to_slider_value(250.0) == 100.0 and d[0].to_slider_value(33.4) == 33.0


# (min, max, step, host value, is_choice) -> what hostParameterToJsfxSliderValue leaves in st.sliders[]. Expected values were
# worked out from src/JSFXJuceProcessor.cpp:5556-5596 with exact rational arithmetic and one IEEE rounding per C++ operation --
# independently of zajit/sliders.py: min / max / step are floats (`info.min` etc., parsed :706-931) widened to double; the host
# value arrives through `std::atomic<float>` (:5566, :9792); a choice is `min + llround(v) * step` (:5578-5582); then
# jlimit (:5584), `q = llround((v - min) / step); v = min + q * step` in double (:5591-5593) and jlimit again (:5594).
QUANTISER_TABLE = [
    ((-24, 24, 0.1, 3.14159, False), 3.1000004038214684),         # q = 271; 0.1f = 0.100000001490116
    ((0, 1, 0.001, 0.8249, False), 0.8250000391853973),
    ((0, 1, 0.01, 0.005, False), 0.009999999776482582),          # 0.005f / 0.01f = 0.50000001...: rounds up to one step
    ((0, 1, 0.01, 0.995, False), 0.9999999776482582),            # q = 100: 100 * 0.01f stays below the float max of 1: no clamp
    ((20, 20000, 1, 440.5, False), 441.0),                       # llround: halves away from zero
    ((-72, 0, 0.1, -36.04999, False), -35.9999994635582),
    ((0.1, 2000, 0.1, 8.05, False), 8.100000120699406),          # min itself is 0.1f
    ((0, 0.95, 0.001, 0.9504, False), 0.949999988079071),        # clamped to the float max first
    ((0, 1, 0, 0.3, False), 0.30000001192092896),                # no step: the host's float, widened
    ((-100, 100, 0.5, -0.25, False), 0.0),                       # (99.75 / 0.5) = 199.5 -> 200
    ((-100, 100, 0.5, 0.25, False), 0.5),
    ((0, 3, 1, 1.5, True), 2.0),                                 # choice: llround(1.5) = 2
    ((0, 3, 1, 7, True), 3.0),                                   # choice index past the end: clamped
    ((1, 9, 2, 2.49, True), 5.0),                                # choice: min + 2 * step
    ((0, 720, 0.1, 30, False), 30.000000447034836),              # 300 * 0.1f
    ((0.01, 1, 0.01, 0.25, False), 0.24999999441206455),
    ((-24, 24, 0.01, 1e9, False), 23.999998927116394),           # clamp, then 4800 * 0.01f - 24
    ((-24, 24, 0.01, -1e9, False), -24.0),
]


@pytest.mark.parametrize("row,want", QUANTISER_TABLE)
def test_slider_quantiser_against_hand_evaluated_reference(row, want):
    """the golden slider rows come from zajit/sliders.py, so the quantiser itself needs a pin that does
    not: this table."""
    from zajit.sliders import SliderDecl
    mn, mx, st, val, choice = row
    f32 = lambda v: float(np.float32(v))
    d = SliderDecl(index0=0, default=0.0, vmin=f32(mn), vmax=f32(mx), step=f32(st), is_choice=choice)
    assert d.to_slider_value(val) == want


def test_leaf_discovery_contract(tmp_path):
    """plugins/<Category>/<Key>/plugin.json, exactly two levels deep, entry -> source (scripts/pluginlib.py:105-240)."""
    from zajit import build
    leaf = tmp_path / "Dynamics" / "Foo"
    (leaf / "src").mkdir(parents=True)
    (leaf / "plugin.json").write_text(json.dumps({"name": "Foo", "pluginType": "jsfx", "entry": "src/Foo.jsfx"}))
    (leaf / "src" / "Foo.jsfx").write_text("desc:x\n@sample\nspl0*=0.5;\n")
    (tmp_path / "Dynamics" / "Deep" / "Er").mkdir(parents=True)
    (tmp_path / "Dynamics" / "Deep" / "Er" / "plugin.json").write_text("{}")
    found = build.discover(tmp_path)
    assert list(found) == ["Foo"] and found["Foo"]["entry"].name == "Foo.jsfx" and found["Foo"]["category"] == "Dynamics"


# ---- section validity of host-coupled builtins + the reference's own compile-smoke scripts ----------------------
REF_TESTS = Path("/root/reference/tests")


@pytest.mark.parametrize("code,msg", [
    ("@sample\nmsg_send(1, 2, 3);\n", "msg_send() is only valid in @block"),
    ("@sample\nx = gmem_put(0, 0, 4);\n", "gmem_put() is only valid in @block"),
    ("@init\np = 1;\n@sample\nspl0 += sample_export_mem(p, 1, 0, 0, 64);\n", "sample_export_mem() is only valid in @block"),
    ("@sample\ngmem_attach(\"x\");\n", "gmem_attach() is only valid in @init, @slider, or @block"),
    ("@sample\nid = instance_id();\n", "instance_id() is only valid in @init, @slider, or @block"),
])
def test_section_validity_messages(code, msg):
    """Same rule table and message text as the reference (dsp_jsfx_aot.py:1544-1605; pinned by its comm / sample-pool test
    drivers, scripts/run_dsp-jsfx_commtests.py:65-66)."""
    from zajit import program, syntax
    with pytest.raises(syntax.JsfxSyntaxError) as ei:
        program.analyse("desc:t\n" + code)
    assert msg in str(ei.value)


def test_block_placement_is_accepted_and_functions_are_not_walked():
    from zajit import program
    program.analyse("desc:t\n@block\nmsg_send(1,2,3); gmem_put(0,0,4);\n@sample\nspl0 = sample_read(1, 1, 0, 0);\n")
    # calls inside a user function are checked where the function is written, not where it is called (as in the reference)
    program.analyse("desc:t\n@init\nfunction f() ( gmem_put(0, 0, 1); );\n@sample\nf();\n")


@pytest.mark.skipif(not REF_TESTS.exists(), reason="reference test scripts not present on this box")
def test_reference_compile_smoke_scripts():
    """The reference's three test drivers (math / comm / sample-pool) only check that scripts compile or are
    rejected with a given message; the same scripts go through this translator, and the math one through g++ as well
    (every documented math builtin must exist in csrc/zart.h)."""
    from zajit import codegen, program, syntax
    ok = ["dsp-jsfx-math/math_builtins_all.jsfx", "dsp-jsfx-comm/sender.jsfx", "dsp-jsfx-comm/receiver.jsfx",
          "dsp-jsfx-comm/gmem_writer.jsfx", "dsp-jsfx-comm/gmem_reader.jsfx", "dsp-jsfx-comm/ipc_probe.jsfx",
          "dsp-jsfx-sample-pool/sample_pool_probe.jsfx"]
    for rel in ok:
        codegen.make_unit(program.analyse_file(REF_TESTS / rel))
    bad = {"dsp-jsfx-comm/invalid_msg_sample.jsfx": "msg_send() is only valid in @block",
           "dsp-jsfx-comm/invalid_gmem_put_sample.jsfx": "gmem_put() is only valid in @block",
           "dsp-jsfx-sample-pool/invalid_export_sample.jsfx": "sample_export_mem() is only valid in @block"}
    for rel, msg in bad.items():
        with pytest.raises(syntax.JsfxSyntaxError) as ei:
            program.analyse_file(REF_TESTS / rel)
        assert msg in str(ei.value)
    from oracle import port
    so = port.build_port(REF_TESTS / "dsp-jsfx-math/math_builtins_all.jsfx", "reftest_math")
    p = port.Port("reftest_math", 48000.0)
    p.prepare()
    y = p.process(np.zeros((2, 64), np.float32) + 0.25, 64)
    assert np.isfinite(y).all()


COOP_CASES = [
    # (body of @sample, expected number of loops emitted in the replica-lane form)
    ("s = 0; i = 0; loop(64, s += mem[100 + i] * mem[i]; i += 1;); spl0 = s;", 1),                     # FIR
    ("a = 0; b = 0; i = 0; loop(32, c = mem[i]; a += c * spl0; b -= c; i += 2;); spl0 = a + b;", 1),       # two sums, temp, step 2
    ("function tap(k) local(j) ( j = (k + 3) & 63; mem[j]; ); s = 0; i = 0; loop(64, s += tap(i); i += 1;); spl0 = s;", 1),
    ("s = 0; i = 0; loop(64, mem[i] = s; s += 1; i += 1;);", 1),                                         # arena store: a map loop, two counters
    ("s = 0; i = 0; loop(64, s += mem[i] * s; i += 1;);", 0),                                            # sum read in the trip
    ("s = 0; i = 0; y = 0; loop(64, s += y; y = mem[i]; i += 1;);", 0),                                  # y carried between trips
    ("s = 0; i = 0; loop(64, s += mem[i]; i += 0.5;);", 0),                                              # non-integer step
    ("s = 0; i = 0; loop(64, s += mem[i]; mem[i] > 0 ? i += 1;);", 0),                                   # conditional counter
    ("s = 0; i = 0; loop(64, s += rand(1); i += 1;);", 0),                                               # impure builtin
    ("s = 0; i = 0; loop(8, j = 0; loop(8, s += mem[i + j]; j += 1;); i += 1;);", 1),                    # only the inner loop
]


COOP_CASES += [
    # elementwise ("map") loops: _map_plan. Whether the trips are independent is decided at run time (za_map_ok).
    ("i = 0; loop(64, mem[200 + i] = mem[i] * 2; i += 1;);", 1),
    ("i = 0; loop(64, mem[i + 1] = mem[i]; i += 1;);", 1),                                               # (the guard refuses it when it runs)
    ("i = 0; loop(64, x = mem[i]; x < 0 ? x = 0; mem[i] = x; i += 1;);", 1),                             # conditional write after a definition
    ("i = 0; while (i < 64) ( mem[2 * i] += mem[2 * i + 1]; i += 1; );", 1),                             # while form
    ("b = 64; function add(d, r, n) local(i) ( i = 0; while (i < n) ( d[i] += r[i]; i += 1; ); ); add(0, b, 32); add(b, 0, 32);", 1),
    ("i = 0; p = 0; loop(64, t = mem[i]; mem[i] = t + p; p = t; i += 1;);", 0),                          # p carried between trips
    ("i = 0; loop(64, mem[i] > 0 ? y = 1; mem[i] = y; i += 1;);", 0),                                    # y only written conditionally
    ("i = 0; loop(64, mem[i] = rand(1); i += 1;);", 0),                                                  # impure builtin
    ("i = 0; loop(64, mem[mem[i]] = 1; i += 1;);", 0),                                                   # address not affine in the counter
    ("i = 0; n = 64; while (i < n) ( mem[i] = 0; n -= 1; i += 1; );", 0),                                # the bound moves
]


@pytest.mark.parametrize("body,expected", COOP_CASES)
def test_accumulation_loop_recognition(body, expected):
    """zajit/emit.py _coop_plan: which loops may run their trips on the replica lanes of an instance."""
    from zajit import codegen, program
    unit = codegen.make_unit(program.analyse("desc:t\n@sample\n" + body + "\n"))
    assert unit.code.count("ZA_COOP_ON(s)") == expected, unit.code
    assert bool({"coop", "coopmap"} & set(unit.features)) == (expected > 0)
