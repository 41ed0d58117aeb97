"""Time-parallel lowering of a leaf's @sample section: ONE WAVEFRONT PER INSTANCE, lane = frame.

The generic kernels (csrc/zab_generic.hip.h) run a script the way jsfx_process_block does (dsp_jsfx_aot.py:5814-5899): one
frame after the other, one lane per instance. Most of a dynamics / filter script is not serial in time at all, though. This
module proves which parts are, per leaf, and emits a second kernel `zab_<leaf>_tpar` that processes 64 consecutive frames
of one instance at once:

  1. @sample (user functions inlined, conditionals if-converted) becomes a DAG over the values of ONE frame: inputs
     spl0.., invariants (variables / sliders that @sample never writes: they change only in @slider / @init), constants,
     and `state-in` nodes -- the value a variable written by @sample had at the end of the PREVIOUS frame.
  2. The cross-frame edges out(v)[t-1] -> state-in(v)[t] close cycles. Strongly connected components of that graph are
     the true recurrences; everything else is feed-forward in time and runs one lane per frame.
       * no cycle through state-in(v) ......... v is a delayed signal: a one-lane shift of out(v) (DPP wave_shr),
       * a cycle that is AFFINE in its states .. y[t] = A[t] y[t-1] + b[t] with A, b free of y (one-poles, leaky
         integrators, counters, sample-and-hold `c ? y = x`, biquads as 2x2): a weighted prefix scan over the wavefront
         with DPP row_shr / row_bcast moves (the scheme of the hand-written DDT kernel, csrc/kernels/ddt_fast.hip.h:98-108),
       * anything else (attack/release smoothers whose coefficient depends on the state, hold counters, ...): the
         minimal cycle runs as a uniform 64-step loop, inputs broadcast with v_readlane, independent cycles that are ready
         at the same time share one loop (instruction-level parallelism instead of lanes).
     In every case the component only has to deliver state-in(v) per lane; all nodes of the frame, the components' own
     included, are then evaluated one lane per frame with the script's own expressions.
  3. Values that depend on invariants only are computed once per launch.

The state a launch leaves in vars[] / spl[] is what the serial path leaves: every variable @sample writes holds its value
at the last frame. Affine components differ from the serial order of operations by re-association only (O(1e-16)
relative), checked by the same reference-VM fixtures as the generic path (tests/test_tpar.py, tests/test_catalog_gpu.py).

`Plan.simulate` is a numpy restatement of the staged algorithm (one array element per lane) used by the CPU tests to pin
the analysis itself -- classification, coefficients, carries, partial chunks -- without a GPU.
"""
from __future__ import annotations

import math
import os
from typing import Dict, List, Optional, Tuple

import numpy as np

from . import syntax as S
from .emit import NOOP_CALLS, PURE_MATH1, PURE_MATH2, c_double
from .program import Program, is_slider_name, is_spl_name

WAVE = 64
RNG_INDEX = "rand#index"      # hidden state: MT19937 outputs consumed since the start of the launch


class Unsupported(Exception):
    """@sample uses a construct the time-parallel lowering does not handle; the leaf keeps the generic kernel only."""


class N:
    __slots__ = ("i", "kind", "op", "args", "val", "name", "uniform")

    def __init__(self, i, kind, op=None, args=(), val=None, name=None):
        self.i, self.kind, self.op, self.args, self.val, self.name = i, kind, op, tuple(args), val, name
        self.uniform = False

    def __repr__(self):
        if self.kind == "const":
            return f"#{self.i}:{self.val!r}"
        if self.kind in ("var", "inv", "st", "in"):
            return f"#{self.i}:{self.kind}({self.name})"
        return f"#{self.i}:{self.op}(" + ",".join(str(a.i) for a in self.args) + ")"


BIN_OPS = {"+", "-", "*", "/", "<", "<=", ">", ">=", "==", "!=", "^", "|", "&", "~", "<<", ">>", "%"}
CALL1 = set(PURE_MATH1) | {"sqr", "sign", "invsqrt"}
MT_N, MT_M = 624, 397
CALL2 = set(PURE_MATH2) | {"min", "max"}


# ----------------------------------------------------------------------------------------------------------------------
# 1. one frame of @sample as a DAG
# ----------------------------------------------------------------------------------------------------------------------
class FrameGraph:
    def __init__(self, prog: Program, nch: int):
        self.p, self.nch = prog, nch
        self.nodes: List[N] = []
        self.memo: Dict[tuple, N] = {}
        self.env: Dict[str, N] = {}
        self.varnodes: Dict[str, N] = {}
        self.written: List[str] = []
        self.scope: List[Dict[str, str]] = []
        self.depth = 0
        self.rand_sites = 0
        self.ZERO, self.ONE = self.const(0.0), self.const(1.0)

    # -- node construction -------------------------------------------------------------------------------------------------
    def mk(self, kind, op=None, args=(), val=None, name=None) -> N:
        key = (kind, op, tuple(a.i for a in args), repr(val), name)
        n = self.memo.get(key)
        if n is None:
            n = N(len(self.nodes), kind, op, args, val, name)
            self.nodes.append(n)
            self.memo[key] = n
        return n

    def const(self, v: float) -> N:
        return self.mk("const", val=float(v))

    def op(self, op, *args) -> N:
        return self.mk("op", op=op, args=args)

    def sel(self, c: N, a: N, b: N) -> N:
        return a if a is b else self.mk("op", op="sel", args=(c, a, b))

    # -- names ------------------------------------------------------------------------------------------------------------
    def _canon(self, name: str) -> str:
        if self.scope and name in self.scope[-1]:
            return self.scope[-1][name]
        return name

    def read(self, name: str) -> N:
        key = self._canon(name)
        if key in self.env:
            return self.env[key]
        if key.startswith("%"):
            raise Unsupported(f"parameter {name} read before it was bound")
        if key in self.varnodes:
            return self.varnodes[key]
        k = is_spl_name(key)
        if k is not None:
            if not 0 <= k < 64:
                raise Unsupported("spl index out of range")
            n = self.mk("in", name=key, val=k) if k < self.nch else self.mk("var", name=key)
        elif key in ("mem", "gmem"):
            raise Unsupported("mem/gmem used as a value")
        elif key == "samplesblock":
            raise Unsupported("@sample reads samplesblock (per host block)")
        else:
            if (is_slider_name(key) is None and key not in ("srate", "midi_bus", "ext_midi_bus", RNG_INDEX) and key not in self.p.vars):
                raise Unsupported(f"unknown variable {key}")
            n = self.mk("var", name=key)
        self.varnodes[key] = n
        return n

    def write(self, name: str, node: N):
        key = self._canon(name)
        if not key.startswith("%"):
            if is_slider_name(key) is not None:
                raise Unsupported("@sample writes a slider")
            if key in ("srate", "samplesblock", "mem", "gmem", "midi_bus", "ext_midi_bus"):
                raise Unsupported(f"@sample writes {key}")
            k = is_spl_name(key)
            if k is not None and not 0 <= k < 64:
                raise Unsupported("spl index out of range")
            if k is None and key not in self.p.vars and key != RNG_INDEX:
                raise Unsupported(f"unknown variable {key}")
            if key not in self.written:
                self.written.append(key)
        self.env[key] = node

    # -- evaluation in the reference emitter's order (dsp_jsfx_aot.py:4263-5590; zajit/emit.py) -------------------------------
    def ev(self, n) -> N:
        return getattr(self, "v_" + type(n).__name__)(n)

    def v_Num(self, n):
        return self.const(n.value)

    def v_Str(self, n):
        raise Unsupported("string literal in @sample")

    def v_Var(self, n):
        nm = n.name
        if not (self.scope and nm in self.scope[-1]):
            if nm == "$pi":
                return self.const(math.pi)
            if nm == "$phi":
                return self.const((1.0 + math.sqrt(5.0)) * 0.5)
            if nm == "$e":
                return self.const(math.e)
            if nm.startswith("$x") and len(nm) > 2:
                try:
                    return self.const(float(int(nm[2:], 16)))
                except ValueError:
                    pass
            if nm == "mem":
                return self.ZERO
        return self.read(nm)

    def v_Index(self, n):
        raise Unsupported("mem[] / gmem[] access in @sample")

    def v_Loop(self, n):
        raise Unsupported("loop() in @sample")

    def v_While(self, n):
        raise Unsupported("while in @sample")

    def v_FuncDef(self, n):
        raise Unsupported("nested function definition")

    def v_Unary(self, n):
        a = self.ev(n.a)
        if n.op == "+":
            return a
        if n.op == "-":
            return self.op("neg", a)
        if n.op == "!":
            return self.op("not", a)
        raise Unsupported(f"unary {n.op}")

    def v_Binary(self, n):
        if n.op in ("&&", "||"):
            l = self.ev(n.l)
            env0 = self.env
            self.env = dict(env0)
            r = self.ev(n.r)
            env_r = self.env
            if all(env_r.get(k) is v for k, v in env0.items()) and len(env_r) == len(env0):
                self.env = env0                      # right operand has no effects: both sides evaluated, plain logic
                return self.op("land" if n.op == "&&" else "lor", l, r)
            # short circuit with effects on the right: they happen iff the left operand lets the right one run
            rb = self.op("truth", r)
            if n.op == "&&":
                self.env = self._merge(l, env_r, env0, env0)
                return self.sel(l, rb, self.ZERO)
            self.env = self._merge(l, env0, env_r, env0)
            return self.sel(l, self.ONE, rb)
        if n.op not in BIN_OPS:
            raise Unsupported(f"binary {n.op}")
        l = self.ev(n.l)
        r = self.ev(n.r)
        return self.op(n.op, l, r)

    def _lookup_incoming(self, key: str, env0) -> N:
        if key in env0:
            return env0[key]
        if key.startswith("%"):
            return self.ZERO
        save, self.env = self.env, env0
        try:
            return self.read(key)
        finally:
            self.env = save

    def _merge(self, c: N, env_t, env_e, env0) -> Dict[str, N]:
        out = dict(env0)
        for key in list(env_t.keys()) + [k for k in env_e if k not in env_t]:
            a = env_t[key] if key in env_t else self._lookup_incoming(key, env0)
            b = env_e[key] if key in env_e else self._lookup_incoming(key, env0)
            out[key] = self.sel(c, a, b)
        return out

    def _branch(self, cond_ast, then_ast, else_ast) -> Tuple[N, N]:
        c = self.ev(cond_ast)
        env0 = self.env
        scope_keys = None
        self.env = dict(env0)
        vt = self.ev(then_ast) if then_ast is not None else self.ZERO
        env_t = self.env
        self.env = dict(env0)
        ve = self.ev(else_ast) if else_ast is not None else self.ZERO
        env_e = self.env
        self.env = self._merge(c, env_t, env_e, env0)
        return c, self.sel(c, vt, ve)

    def v_Cond(self, n):
        return self._branch(n.cond, n.then, n.els)[1]

    def v_If(self, n):
        self._branch(n.cond, n.then, n.els)
        return self.ZERO

    def v_Seq(self, n):
        v = self.ZERO
        for it in n.items:
            v = self.ev(it)
        if n.items and isinstance(n.items[-1], (S.If, S.While)):
            return self.ZERO
        return v

    def v_Assign(self, n):
        tgt = n.target
        rhs = self.ev(n.value)
        if not isinstance(tgt, S.Var):
            raise Unsupported("assignment to mem[] / slider() / spl() in @sample")
        if n.op == "=":
            val = rhs
        else:
            bop = n.op[:-1]
            if bop not in BIN_OPS:
                raise Unsupported(f"assignment operator {n.op}")
            val = self.op(bop, self.read(tgt.name), rhs)
        self.write(tgt.name, val)
        return val

    def v_Call(self, n):
        fn = n.fn
        if fn in self.p.fns:
            f = self.p.fns[fn]
            if len(n.args) != len(f.params):
                raise Unsupported(f"{fn}: arity")
            if self.depth > 32:
                raise Unsupported("call depth")
            args = [self.ev(a) for a in n.args]
            self.depth += 1
            frame = {p: f"%{self.depth}_{len(self.scope)}_{p}" for p in f.params}
            self.scope.append(frame)
            for p, a in zip(f.params, args):
                self.env[frame[p]] = a
            v = self.ev(f.body)
            self.scope.pop()
            for k in frame.values():
                self.env.pop(k, None)
            self.depth -= 1
            return v
        if fn.startswith("gfx_") or fn in NOOP_CALLS:
            for a in n.args:
                self.ev(a)
            return self.ZERO
        if fn == "abs":
            fn = "fabs"
        if fn in CALL1:
            if len(n.args) != 1:
                raise Unsupported(f"{fn}: arity")
            return self.op(fn, self.ev(n.args[0]))
        if fn in CALL2:
            if len(n.args) != 2:
                raise Unsupported(f"{fn}: arity")
            a = self.ev(n.args[0])
            b = self.ev(n.args[1])
            return self.op(fn, a, b)
        if fn == "__memtop" and not n.args:
            return self.const(float(self.p.memtop))
        if fn == "rand" and len(n.args) <= 1:
            # za_rand (csrc/zart.h): ((double)next_word * (1 / 4294967295)) * max(1, floor(arg)). The generator's position is a
            # state like any other: the hidden counter RNG_INDEX = outputs consumed so far in this launch, stepped by every call
            # that executes (if-conversion makes the step conditional); the word itself is a pure function of the position.
            arg = self.ev(n.args[0]) if n.args else self.ONE
            idx = self.read(RNG_INDEX)
            self.write(RNG_INDEX, self.op("+", idx, self.ONE))
            self.rand_sites += 1
            fl = self.op("floor", arg)
            m = self.sel(self.op("<", fl, self.ONE), self.ONE, fl)
            return self.op("*", self.op("*", self.op("mtout", idx), self.const(1.0 / 4294967295.0)), m)
        raise Unsupported(f"builtin {fn} in @sample")


# ----------------------------------------------------------------------------------------------------------------------
# 2. analysis: recurrences, affine forms, schedule
# ----------------------------------------------------------------------------------------------------------------------
class Component:
    """One recurrence: the state variables whose state-in nodes lie on a common cycle."""

    def __init__(self, names, members):
        self.names: List[str] = names            # state variables, order = order of first write in the frame
        self.members: List[N] = members          # nodes on the cycle(s), topological order within the frame
        self.kind = "serial"                     # "scan": affine, at most 2 states; "spec": affine once its switches are fixed
        self.A: List[List[N]] = []               # scan / spec: y[t] = A y[t-1] + b  (nodes free of the component's states)
        self.b: List[N] = []
        self.ext: List[N] = []                   # non-member operands of the members (varying ones are broadcast per step)
        # spec: switches = conditions (of ?:, min, max, abs) that depend on the component's own states. With every switch
        # fixed the recurrence is affine, so a guessed switch pattern gives the states by one scan; the states give the
        # pattern back; a pattern that reproduces itself is the serial solution (induction over the frames).
        self.conds: List[N] = []                 # condition node of each switch (member or synthetic compare of members)
        self.gnodes: List[N] = []                # its placeholder ("guess") in A / b
        self.gdep: List[N] = []                  # nodes of A / b that depend on a placeholder, topological order
        self.slice: List[N] = []                 # nodes needed to evaluate the conditions from the states, topological order
        self.inputs: List[N] = []                # everything outside that the unit reads


class Plan:
    def __init__(self):
        self.g: FrameGraph = None
        self.nch = 0
        self.outs: Dict[str, N] = {}             # variable (or splK) -> node holding its value at the end of a frame
        self.spl_out: List[N] = []               # per processed channel
        self.st: Dict[str, N] = {}               # state variable -> its state-in node
        self.items: List[tuple] = []             # schedule of one chunk
        self.uniform: List[N] = []               # per-launch nodes, topological order
        self.invariants: List[N] = []
        self.inputs: List[N] = []
        self.stats: Dict[str, int] = {}
        self.uses_rand = False

    # ------------------------------------------------------------------------------------------------------------------
    # numpy restatement of the staged algorithm (tests)
    # ------------------------------------------------------------------------------------------------------------------
    def _sim_serial(self, comp: "Component", val, carry, tn):
        cur = {nm: carry[nm] for nm in comp.names}
        caps = {nm: np.zeros(WAVE) for nm in comp.names}
        for t in range(tn):
            loc: Dict[int, np.float64] = {}
            for nm in comp.names:
                caps[nm][t] = cur[nm]
                loc[self.st[nm].i] = cur[nm]
            for m in comp.members:
                if m.kind == "st":
                    continue
                ops = []
                for a in m.args:
                    if a.i in loc:
                        ops.append(loc[a.i])
                    else:
                        v = val[a.i]
                        ops.append(v if np.ndim(v) == 0 else v[t])
                loc[m.i] = np.float64(_np_op(m.op, ops))
            for nm in comp.names:
                cur[nm] = loc[self.outs[nm].i]
        for nm in comp.names:
            caps[nm][tn:] = cur[nm]
            val[self.st[nm].i] = caps[nm]

    def simulate(self, vars0: Dict[str, float], x: np.ndarray, sliders=None, srate=48000.0, spl0=None, mt=None):
        """x: [nch, frames] float32. vars0: name -> value before the launch (missing names are 0). mt: (randMT[624], randIndex)
        before the launch for scripts that call rand(); self.mt_after holds the pair after it.
        Returns (y float32 [nch, frames], vars after {name: value}, spl after {k: value})."""
        stream = MtStream(*(mt if mt is not None else (None, 0))) if self.uses_rand else None
        _MT_CTX[0] = stream
        x = np.asarray(x, dtype=np.float32)
        frames = x.shape[1]
        sliders = np.zeros(64) if sliders is None else np.asarray(sliders, dtype=np.float64)
        spl_state = dict(spl0 or {})

        def inv_value(name):
            k = is_slider_name(name)
            if k is not None:
                return float(sliders[k - 1])
            if name == "srate":
                return float(srate)
            if name in ("midi_bus", "ext_midi_bus", RNG_INDEX):
                return 0.0
            k = is_spl_name(name)
            if k is not None:
                return float(spl_state.get(k, 0.0))
            return float(vars0.get(name, 0.0))

        val: Dict[int, np.ndarray] = {}
        self.spec_log = []                     # (states, iterations, converged) per switched recurrence and chunk
        with np.errstate(all="ignore"):
            for n in self.uniform:
                if n.kind == "const":
                    val[n.i] = np.float64(n.val)
                elif n.kind == "inv":
                    val[n.i] = np.float64(inv_value(n.name))
                else:
                    val[n.i] = _np_op(n.op, [val[a.i] for a in n.args])
            carry = {name: np.float64(inv_value(name)) for name in self.st}
            y = np.zeros_like(x)
            final_vals: Dict[int, float] = {}
            lane = np.arange(WAVE)
            for f0 in range(0, max(frames, 0), WAVE):
                tn = min(WAVE, frames - f0)
                last = tn - 1
                for n in self.inputs:
                    col = np.zeros(WAVE)
                    col[:tn] = x[int(n.val), f0:f0 + tn].astype(np.float64)
                    val[n.i] = col
                for it in self.items:
                    kind = it[0]
                    if kind == "par":
                        n = it[1]
                        val[n.i] = np.broadcast_to(_np_op(n.op, [val[a.i] for a in n.args]), (WAVE,)).astype(np.float64)
                    elif kind == "shift":
                        name = it[1]
                        src = np.broadcast_to(val[self.outs[name].i], (WAVE,))
                        sh = np.empty(WAVE)
                        sh[0] = carry[name]
                        sh[1:] = src[:-1]
                        val[self.st[name].i] = sh
                    elif kind == "scan":
                        comp: Component = it[1]
                        d = len(comp.names)
                        A = np.stack([np.stack([np.broadcast_to(val[comp.A[r][c].i], (WAVE,)) for c in range(d)]) for r in range(d)])
                        b = np.stack([np.broadcast_to(val[comp.b[r].i], (WAVE,)) for r in range(d)])
                        A, b = A.astype(np.float64).copy(), b.astype(np.float64).copy()          # [d,d,64], [d,64]
                        states = _scan_exclusive(A, b, np.array([carry[nm] for nm in comp.names]))
                        for r, nm in enumerate(comp.names):
                            val[self.st[nm].i] = states[r]
                    elif kind == "serial":
                        for comp in it[1]:
                            self._sim_serial(comp, val, carry, tn)
                    elif kind == "spec":
                        for comp in it[1]:
                            d = len(comp.names)

                            def conds_from(states):
                                loc = {self.st[nm].i: states[r] for r, nm in enumerate(comp.names)}
                                for m in comp.slice:
                                    loc[m.i] = np.broadcast_to(_np_op(m.op, [loc[a.i] if a.i in loc else val[a.i] for a in m.args]), (WAVE,))
                                return [_truthy(np.broadcast_to(loc[c.i] if c.i in loc else val[c.i], (WAVE,))) for c in comp.conds]

                            prev = [np.full(WAVE, carry[nm]) for nm in comp.names]
                            gs = conds_from(prev)
                            converged, iters = False, 0
                            while iters < SPEC_MAX:
                                iters += 1
                                loc = {gn.i: np.where(gs[k], 1.0, 0.0) for k, gn in enumerate(comp.gnodes)}
                                for n in comp.gdep:
                                    loc[n.i] = _np_op(n.op, [loc[a.i] if a.i in loc else val[a.i] for a in n.args])
                                gv = lambda n: np.broadcast_to(loc[n.i] if n.i in loc else val[n.i], (WAVE,)).astype(np.float64)
                                A = np.stack([np.stack([gv(comp.A[r][c]) for c in range(d)]) for r in range(d)]).copy()
                                b = np.stack([gv(comp.b[r]) for r in range(d)]).copy()
                                states = _scan_exclusive(A, b, np.array([carry[nm] for nm in comp.names]))
                                ng = conds_from(states)
                                changed = any(bool(np.any(x[:tn] != y[:tn])) for x, y in zip(ng, gs))
                                # a pattern that only still moves where both of its branches agree (a smoother sitting on its
                                # target, a value on its clamp) leaves the states where they were: that is converged too
                                moved = any(bool(np.any(np.abs(a[:tn] - b[:tn]) > SPEC_TOL * np.maximum(np.abs(a[:tn]), np.abs(b[:tn]))))
                                            for a, b in zip(states, prev))
                                gs, prev = ng, states
                                if not (changed and moved):
                                    converged = True
                                    break
                            self.spec_log.append((tuple(comp.names), iters, converged))
                            if converged:
                                for r, nm in enumerate(comp.names):
                                    val[self.st[nm].i] = states[r]
                            else:
                                self._sim_serial(comp, val, carry, tn)
                    else:
                        raise AssertionError(kind)
                for ch in range(self.nch):
                    v = np.broadcast_to(val[self.spl_out[ch].i], (WAVE,))
                    y[ch, f0:f0 + tn] = v[:tn].astype(np.float32)
                for name in self.st:
                    v = val[self.outs[name].i]
                    carry[name] = np.float64(v if np.ndim(v) == 0 else v[last])
                if stream is not None:
                    stream.end_chunk(int(carry[RNG_INDEX]))
                if f0 + WAVE >= frames:
                    for name, o in list(self.outs.items()) + [(f"spl{ch}", self.spl_out[ch]) for ch in range(self.nch)]:
                        v = val[o.i]
                        final_vals[name] = float(v if np.ndim(v) == 0 else v[last])
        vars_after = dict(vars0)
        spl_after = dict(spl_state)
        self.mt_after = stream.state(int(final_vals.get(RNG_INDEX, 0))) if stream is not None else mt
        final_vals.pop(RNG_INDEX, None)
        for name, v in final_vals.items():
            k = is_spl_name(name)
            if k is not None:
                spl_after[k] = v
            else:
                vars_after[name] = v
        return y, vars_after, spl_after


class MtStream:
    """MT19937 as za_mt_next (csrc/zart.h) runs it, in the form the kernels use: two generations side by side, the next one
    produced from the current one in three lane-parallel phases (element k of a new generation needs new[k - 227] from
    k = 227 on, so [0, 227), [227, 454) and [454, 623) are each parallel inside; element 623 closes the ring)."""

    def __init__(self, table=None, mti: int = 0):
        self.seeded_here = mti == 0
        if mti == 0:                       # first use: seed, position at the end -> the first word comes from the next generation
            t = np.zeros(MT_N, dtype=np.uint64)
            prev = 0x4141F00D
            t[0] = prev
            for k in range(1, MT_N):
                prev = (1812433253 * (prev ^ (prev >> 30)) + k) & 0xFFFFFFFF
                t[k] = prev
            self.cur, self.pos0 = t, MT_N
        else:
            self.cur, self.pos0 = np.asarray(table, dtype=np.uint64).copy(), int(mti)
        self.orig = (None if table is None else np.asarray(table).copy(), int(mti))
        self.nxt = self.twist(self.cur)

    @staticmethod
    def twist(cur):
        nxt = np.zeros(MT_N, dtype=np.uint64)

        def tw(a, b):
            y = (a & 0x80000000) | (b & 0x7FFFFFFF)
            return (y >> 1) ^ np.where(y & 1, 0x9908B0DF, 0).astype(np.uint64)

        k = np.arange(0, 227)
        nxt[k] = cur[k + MT_M] ^ tw(cur[k], cur[k + 1])
        k = np.arange(227, 454)
        nxt[k] = nxt[k - 227] ^ tw(cur[k], cur[k + 1])
        k = np.arange(454, 623)
        nxt[k] = nxt[k - 227] ^ tw(cur[k], cur[k + 1])
        nxt[623] = nxt[396] ^ tw(cur[623:624], nxt[0:1])[0]
        return nxt

    def word(self, idx):
        pos = self.pos0 + np.asarray(idx, dtype=np.int64)
        pos = np.clip(pos, 0, 2 * MT_N - 1)
        y = np.where(pos < MT_N, self.cur[np.minimum(pos, MT_N - 1)], self.nxt[np.maximum(pos - MT_N, 0)]).astype(np.uint64)
        y ^= y >> 11
        y ^= (y << 7) & 0x9D2C5680
        y ^= (y << 15) & 0xEFC60000
        y ^= y >> 18
        return (y & 0xFFFFFFFF).astype(np.float64)

    def end_chunk(self, total: int):
        """`total` words consumed so far: retire a generation once the last consumed word lies in the next one."""
        if self.pos0 + total > MT_N:
            self.cur = self.nxt
            self.nxt = self.twist(self.cur)
            self.pos0 -= MT_N

    def state(self, total: int):
        if total <= 0:
            return self.orig
        return self.cur.astype(np.uint32), self.pos0 + total


_MT_CTX: List[Optional[MtStream]] = [None]

SPEC_TOL = 1.0e-13    # relative change of a state between two iterations below which it counts as settled (ZT_SPEC_TOL)
SPEC_MAX = 8          # iterations of a switched recurrence before the chunk falls back to its serial loop (ZT_SPEC_MAX)


def _scan_exclusive(A, b, c0):
    """Kogge-Stone over the lanes, element = the map y -> A y + b, combined as (current o earlier); returns the state BEFORE
    each frame given the state c0 before the chunk. A: [d, d, 64], b: [d, 64]."""
    A, b = A.copy(), b.copy()
    s = 1
    while s < WAVE:
        A2, b2 = A.copy(), b.copy()
        for t in range(s, WAVE):
            A2[:, :, t] = A[:, :, t] @ A[:, :, t - s]
            b2[:, t] = A[:, :, t] @ b[:, t - s] + b[:, t]
        A, b = A2, b2
        s *= 2
    yinc = np.einsum("rct,c->rt", A, c0) + b
    out = np.empty_like(yinc)
    out[:, 0] = c0
    out[:, 1:] = yinc[:, :-1]
    return out


def _truthy(a):
    return (a < 0.0) | (a > 0.0)


def _i32(a):
    """za_i32 of csrc/zart.h, element by element (tests only)."""
    a = np.asarray(a, dtype=np.float64)
    flat = a.reshape(-1)
    out = np.zeros(flat.shape, dtype=np.int64)
    for k, v in enumerate(flat):
        w = -(1 << 63) if not (-9.2233720368547758e18 < v < 9.2233720368547758e18) else int(v)
        w &= 0xFFFFFFFF
        out[k] = w - (1 << 32) if w >= (1 << 31) else w
    return out.reshape(a.shape)


def _np_op(op, a):
    if op == "+":
        return a[0] + a[1]
    if op == "-":
        return a[0] - a[1]
    if op == "*":
        return a[0] * a[1]
    if op == "/":
        return np.divide(a[0], a[1])
    if op == "neg":
        return 0.0 - a[0]
    if op == "not":
        return np.where(a[0] == 0.0, 1.0, 0.0)
    if op == "truth":
        return np.where(_truthy(a[0]), 1.0, 0.0)
    if op in ("<", "<=", ">", ">=", "=="):
        f = {"<": np.less, "<=": np.less_equal, ">": np.greater, ">=": np.greater_equal, "==": np.equal}[op]
        return np.where(f(a[0], a[1]), 1.0, 0.0)
    if op == "!=":
        return np.where((a[0] < a[1]) | (a[0] > a[1]), 1.0, 0.0)
    if op == "land":
        return np.where(_truthy(a[0]) & _truthy(a[1]), 1.0, 0.0)
    if op == "lor":
        return np.where(_truthy(a[0]) | _truthy(a[1]), 1.0, 0.0)
    if op == "sel":
        return np.where(_truthy(a[0]), a[1], a[2])
    if op in ("^", "pow"):
        return np.power(np.asarray(a[0], dtype=np.float64), a[1])
    if op in ("|", "&", "~", "<<", ">>", "%"):
        l, r = _i32(a[0]), _i32(a[1])
        if op == "|":
            v = l | r
        elif op == "&":
            v = l & r
        elif op == "~":
            v = l ^ r
        elif op == "<<":
            v = ((l & 0xFFFFFFFF) << (r & 31)) & 0xFFFFFFFF
            v = np.where(v >= 2 ** 31, v - 2 ** 32, v)
        elif op == ">>":
            v = l >> (r & 31)
        else:
            bad = (r == 0) | ((l == -2 ** 31) & (r == -1))
            rr = np.where(bad, 1, r)
            v = np.where(bad, 0, np.fmod(l, rr))           # C remainder: sign of the dividend
        return np.asarray(v, dtype=np.float64)
    if op == "min":
        return np.where(a[0] < a[1], a[0], a[1])
    if op == "max":
        return np.where(a[0] > a[1], a[0], a[1])
    if op == "sqr":
        return a[0] * a[0]
    if op == "sign":
        return np.where(a[0] > 0.0, 1.0, np.where(a[0] < 0.0, -1.0, 0.0))
    if op == "invsqrt":
        f = np.asarray(a[0], dtype=np.float64).astype(np.float32)
        bits = np.atleast_1d(f).view(np.int32)
        bits = (np.int32(0x5f3759df) - (bits >> 1)).astype(np.int32)
        y0 = bits.view(np.float32).astype(np.float64).reshape(np.shape(f))
        return y0 * (1.5 - (0.5 * a[0]) * (y0 * y0))
    if op == "atan2":
        return np.arctan2(a[0], a[1])
    if op == "mtout":
        return _MT_CTX[0].word(np.asarray(a[0]))
    if op in PURE_MATH1:
        f = {"sin": np.sin, "cos": np.cos, "sqrt": np.sqrt, "fabs": np.fabs, "floor": np.floor, "ceil": np.ceil, "asin": np.arcsin,
             "acos": np.arccos, "atan": np.arctan, "exp": np.exp, "log": np.log, "tan": np.tan, "log10": np.log10}[op]
        return f(np.asarray(a[0], dtype=np.float64))
    raise AssertionError(op)


def _sccs(n_nodes: int, succ: List[List[int]]) -> List[List[int]]:
    """Tarjan, iterative. Returns the components in reverse topological order."""
    index = [-1] * n_nodes
    low = [0] * n_nodes
    on = [False] * n_nodes
    stack: List[int] = []
    out: List[List[int]] = []
    counter = 0
    for root in range(n_nodes):
        if index[root] != -1:
            continue
        work = [(root, 0)]
        while work:
            v, pi = work.pop()
            if pi == 0:
                index[v] = low[v] = counter
                counter += 1
                stack.append(v)
                on[v] = True
            recurse = False
            for k in range(pi, len(succ[v])):
                w = succ[v][k]
                if index[w] == -1:
                    work.append((v, k + 1))
                    work.append((w, 0))
                    recurse = True
                    break
                if on[w]:
                    low[v] = min(low[v], index[w])
            if recurse:
                continue
            if low[v] == index[v]:
                comp = []
                while True:
                    w = stack.pop()
                    on[w] = False
                    comp.append(w)
                    if w == v:
                        break
                out.append(comp)
            if work:
                u = work[-1][0]
                low[u] = min(low[u], low[v])
    return out


def build_plan(prog: Program, nch: int) -> Plan:
    """Raises Unsupported when the leaf cannot take the time-parallel kernel."""
    if not prog.has("sample") or nch <= 0:
        raise Unsupported("no audio @sample")
    if prog.has("block"):
        raise Unsupported("@block present")
    g = FrameGraph(prog, nch)
    for st in prog.sections["sample"]:
        g.ev(st)
    if g.scope:
        raise AssertionError("scope leak")
    if g.rand_sites * WAVE > MT_N:
        raise Unsupported("more rand() calls per chunk than one generation of the generator holds")
    plan = Plan()
    plan.g, plan.nch = g, nch
    written = list(g.written)
    # variables @sample leaves as they were (x = x) are not state
    for name in list(written):
        vn = g.varnodes.get(name)
        if vn is not None and g.env.get(name) is vn:
            written.remove(name)
    wset = set(written)
    for name, vn in g.varnodes.items():
        if vn.kind == "var":
            vn.kind = "st" if name in wset else "inv"
    plan.outs = {name: g.env[name] for name in written}
    plan.spl_out = [g.env.get(f"spl{ch}", None) or g.read(f"spl{ch}") for ch in range(nch)]
    plan.st = {name: vn for name, vn in g.varnodes.items() if vn.kind == "st"}

    # live nodes
    live: Dict[int, N] = {}
    todo = list(plan.outs.values()) + list(plan.spl_out)
    while todo:
        n = todo.pop()
        if n.i in live:
            continue
        live[n.i] = n
        todo.extend(n.args)
        if n.kind == "st":
            todo.append(plan.outs[n.name])
    order = sorted(live)                     # creation order is a topological order of the in-frame edges
    pos = {i: k for k, i in enumerate(order)}
    succ: List[List[int]] = [[] for _ in order]
    for i in order:
        n = live[i]
        for a in n.args:
            succ[pos[a.i]].append(pos[i])
        if n.kind == "st":
            succ[pos[plan.outs[n.name].i]].append(pos[i])
    comps_raw = _sccs(len(order), succ)
    comp_of: Dict[int, int] = {}
    components: List[Component] = []
    for comp in comps_raw:
        ids = [order[k] for k in comp]
        cyclic = len(ids) > 1 or any(pos[ids[0]] in succ[pos[ids[0]]] for _ in (0,))
        if not cyclic:
            continue
        members = [live[i] for i in sorted(ids)]
        names = [m.name for m in members if m.kind == "st"]
        names.sort(key=lambda nm: written.index(nm))
        c = Component(names, members)
        for m in members:
            comp_of[m.i] = len(components)
        components.append(c)

    # uniform (per launch) nodes
    for i in order:
        n = live[i]
        if n.kind in ("const", "inv"):
            n.uniform = True
        elif n.kind in ("st", "in"):
            n.uniform = False
        else:
            n.uniform = all(a.uniform for a in n.args) and n.i not in comp_of
    # affine forms
    import os
    for ci, c in enumerate(components):
        _classify(g, plan, c, comp_of, ci)
        if c.kind == "spec" and os.environ.get("ZA_TPAR_NO_SPEC"):
            c.kind = "serial"
    # nodes created by the affine analysis: liveness / uniformity of the new coefficient nodes. Placeholder-dependent nodes
    # and the synthetic compares live inside their unit only.
    inside = {x.i for c in components if c.kind == "spec" for x in c.gdep + c.gnodes + c.slice}
    extra: Dict[int, N] = {}
    todo = [x for c in components if c.kind in ("scan", "spec") for row in c.A for x in row]
    todo += [x for c in components if c.kind in ("scan", "spec") for x in c.b]
    todo += [a for c in components if c.kind == "spec" for x in c.gdep + c.slice for a in x.args]
    while todo:
        n = todo.pop()
        if n.i in live or n.i in extra or n.i in inside:
            continue
        extra[n.i] = n
        todo.extend(n.args)
    for i in sorted(extra):
        n = extra[i]
        live[i] = n
        n.uniform = n.kind in ("const", "inv") or (n.kind == "op" and all(a.uniform for a in n.args))
    order = sorted(live)

    # ---- schedule of one chunk --------------------------------------------------------------------------------------------
    plan.uniform = [live[i] for i in order if live[i].uniform]
    plan.invariants = [n for n in plan.uniform if n.kind == "inv"]
    plan.inputs = [live[i] for i in order if live[i].kind == "in"]
    done = {n.i for n in plan.uniform} | {n.i for n in plan.inputs}
    pending_nodes = [live[i] for i in order if i not in done]
    comp_done = [False] * len(components)
    items: List[tuple] = []

    def comp_inputs(c: Component) -> List[N]:
        return c.inputs

    for c in components:
        mem = {m.i for m in c.members}
        ext, seen = [], set()
        for m in c.members:
            for a in m.args:
                if a.i not in mem and a.i not in seen:
                    seen.add(a.i)
                    ext.append(a)
        c.ext = ext
        if c.kind == "scan":
            c.inputs = [x for row in c.A for x in row] + list(c.b)
        elif c.kind == "spec":
            own = {x.i for x in c.gdep + c.gnodes + c.slice} | mem
            ins, seen = list(ext), {x.i for x in ext}
            for x in [y for row in c.A for y in row] + list(c.b) + [a for y in c.gdep + c.slice for a in y.args]:
                if x.i not in own and x.i not in seen:
                    seen.add(x.i)
                    ins.append(x)
            c.inputs = ins
        else:
            c.inputs = ext

    remaining = list(pending_nodes)
    guard = 0
    while remaining:
        guard += 1
        if guard > 10 * len(order) + 100:
            raise AssertionError("scheduler made no progress")
        progressed = False
        nxt = []
        for n in remaining:
            if n.kind == "st":
                ci = comp_of.get(n.i)
                if ci is None:                              # delayed signal
                    if plan.outs[n.name].i in done:
                        items.append(("shift", n.name))
                        done.add(n.i)
                        progressed = True
                    else:
                        nxt.append(n)
                elif comp_done[ci]:
                    done.add(n.i)
                    progressed = True
                else:
                    nxt.append(n)
                continue
            if all(a.i in done for a in n.args):
                items.append(("par", n))
                done.add(n.i)
                progressed = True
            else:
                nxt.append(n)
        remaining = nxt
        # scans as soon as their coefficients exist (they are lane-parallel work too)
        for ci, c in enumerate(components):
            if not comp_done[ci] and c.kind == "scan" and all(x.i in done for x in comp_inputs(c)):
                items.append(("scan", c))
                comp_done[ci] = True
                progressed = True
        if progressed:
            continue
        # only switched / serial recurrences can move now: every one of a kind that is ready shares one loop
        for kind in ("spec", "serial"):
            ready = [ci for ci, c in enumerate(components) if not comp_done[ci] and c.kind == kind and all(x.i in done for x in c.inputs)]
            if ready:
                break
        if not ready:
            raise AssertionError("dependency cycle outside the recurrences")
        items.append((kind, [components[ci] for ci in ready]))
        for ci in ready:
            comp_done[ci] = True
    plan.items = items
    plan.uses_rand = RNG_INDEX in plan.outs
    plan.stats = {
        "nodes": len(order), "uniform": len(plan.uniform), "par": sum(1 for it in items if it[0] == "par"),
        "shift": sum(1 for it in items if it[0] == "shift"),
        "scan1": sum(1 for it in items if it[0] == "scan" and len(it[1].names) == 1),
        "scan2": sum(1 for it in items if it[0] == "scan" and len(it[1].names) == 2),
        "spec_loops": sum(1 for it in items if it[0] == "spec"),
        "spec_chains": sum(len(it[1]) for it in items if it[0] == "spec"),
        "spec_switches": sum(len(c.conds) for it in items if it[0] == "spec" for c in it[1]),
        "serial_loops": sum(1 for it in items if it[0] == "serial"),
        "serial_chains": sum(len(it[1]) for it in items if it[0] == "serial"),
        "serial_ops": sum(len([m for m in c.members if m.kind != "st"]) for it in items if it[0] == "serial" for c in it[1]),
        "states": len(plan.st), "written": len(plan.outs), "rand_sites": g.rand_sites,
    }
    return plan


def _classify(g: FrameGraph, plan: Plan, c: Component, comp_of: Dict[int, int], ci: int = 0):
    """Affine in the component's own states, with coefficients that do not depend on them? -> "scan".
    Affine once the state-dependent conditions (switches) are fixed? -> "spec". Otherwise it stays "serial"."""
    mem = {m.i for m in c.members}
    names = c.names
    d = len(names)
    if d > 2:
        return

    def add(a: N, b: N) -> N:
        if a is g.ZERO:
            return b
        if b is g.ZERO:
            return a
        return g.op("+", a, b)

    def sub(a: N, b: N) -> N:
        if b is g.ZERO:
            return a
        if a is g.ZERO:
            return g.op("neg", b)
        return g.op("-", a, b)

    def mul(a: N, b: N) -> N:
        if a is g.ZERO or b is g.ZERO:
            return g.ZERO
        if a is g.ONE:
            return b
        if b is g.ONE:
            return a
        return g.op("*", a, b)

    def attempt(allow_guess: bool):
        memo: Dict[int, Optional[tuple]] = {}
        conds: List[N] = []
        gnodes: List[N] = []

        def guess_for(cond: N) -> N:
            for k, x in enumerate(conds):
                if x is cond:
                    return gnodes[k]
            conds.append(cond)
            gn = g.mk("guess", name=f"{ci}", val=len(gnodes))
            gnodes.append(gn)
            return gn

        def pick(cnd: N, a, b):
            co = {k: g.sel(cnd, a[0].get(k, g.ZERO), b[0].get(k, g.ZERO)) for k in set(a[0]) | set(b[0])}
            return (co, g.sel(cnd, a[1], b[1]))

        def aff(n: N):
            if n.i not in mem:
                return ({}, n)
            if n.i in memo:
                return memo[n.i]
            r = None
            if n.kind == "st":
                r = ({n.name: g.ONE}, g.ZERO)
            elif n.kind == "op":
                op = n.op
                if op in ("+", "-"):
                    a, b = aff(n.args[0]), aff(n.args[1])
                    if a and b:
                        f = add if op == "+" else sub
                        co = {k: f(a[0].get(k, g.ZERO), b[0].get(k, g.ZERO)) for k in set(a[0]) | set(b[0])}
                        r = (co, f(a[1], b[1]))
                elif op == "neg":
                    a = aff(n.args[0])
                    if a:
                        r = ({k: sub(g.ZERO, v) for k, v in a[0].items()}, sub(g.ZERO, a[1]))
                elif op == "*":
                    a, b = aff(n.args[0]), aff(n.args[1])
                    if a and b:
                        if not a[0]:
                            r = ({k: mul(a[1], v) for k, v in b[0].items()}, mul(a[1], b[1]))
                        elif not b[0]:
                            r = ({k: mul(v, b[1]) for k, v in a[0].items()}, mul(a[1], b[1]))
                elif op == "/":
                    a, b = aff(n.args[0]), aff(n.args[1])
                    if a and b and not b[0]:
                        r = ({k: g.op("/", v, b[1]) for k, v in a[0].items()}, g.op("/", a[1], b[1]) if a[1] is not g.ZERO else g.ZERO)
                elif op == "sel":
                    cnd = n.args[0]
                    if cnd.i not in mem or allow_guess:
                        a, b = aff(n.args[1]), aff(n.args[2])
                        if a and b:
                            r = pick(cnd if cnd.i not in mem else guess_for(cnd), a, b)
                elif op in ("min", "max") and allow_guess:      # za_min(a, b) = a < b ? a : b,  za_max(a, b) = a > b ? a : b
                    a, b = aff(n.args[0]), aff(n.args[1])
                    if a and b:
                        r = pick(guess_for(g.op("<" if op == "min" else ">", n.args[0], n.args[1])), a, b)
                elif op == "fabs" and allow_guess:                # |x| = x < 0 ? -x : x
                    a = aff(n.args[0])
                    if a:
                        neg = ({k: sub(g.ZERO, v) for k, v in a[0].items()}, sub(g.ZERO, a[1]))
                        r = pick(guess_for(g.op("<", n.args[0], g.ZERO)), neg, a)
            memo[n.i] = r
            return r

        rows = []
        for nm in names:
            r = aff(plan.outs[nm])
            if r is None:
                return None
            rows.append(r)
        return rows, conds, gnodes

    if d == 1 and _persistent_rounding(g, plan, c, mem):
        return                                    # stays "serial": see _persistent_rounding
    res = attempt(False)
    if res is not None:
        c.kind = "scan"
    else:
        res = attempt(True)
        if res is None:
            return
        c.kind = "spec"
    rows, c.conds, c.gnodes = res
    c.A = [[rows[r][0].get(names[k], g.ZERO) for k in range(d)] for r in range(d)]
    c.b = [rows[r][1] for r in range(d)]
    if c.kind == "spec":
        # coefficient nodes that depend on a placeholder (evaluated inside the iteration), topological = creation order
        dep: Dict[int, bool] = {}

        def gd(n: N) -> bool:
            if n.i in dep:
                return dep[n.i]
            r = n.kind == "guess" or any(gd(x) for x in n.args)
            dep[n.i] = r
            return r

        seen: Dict[int, N] = {}
        todo = [x for row in c.A for x in row] + list(c.b)
        while todo:
            n = todo.pop()
            if n.i in seen or not gd(n):
                continue
            seen[n.i] = n
            todo.extend(n.args)
        c.gdep = [seen[i] for i in sorted(seen) if seen[i].kind != "guess"]
        # nodes needed to evaluate the conditions from the states: members (and the synthetic compares) only
        sl: Dict[int, N] = {}
        todo = list(c.conds)
        synth = {x.i for x in c.conds if x.i not in mem}
        while todo:
            n = todo.pop()
            if n.i in sl or (n.i not in mem and n.i not in synth):
                continue
            sl[n.i] = n
            todo.extend(n.args)
        c.slice = [sl[i] for i in sorted(sl) if sl[i].kind != "st"]


def _const_value(n: N) -> Optional[float]:
    """Value of a node built from constants only."""
    if n.kind == "const":
        return float(n.val)
    if n.kind == "op" and n.op in ("+", "-", "*", "neg") and n.args:
        v = [_const_value(a) for a in n.args]
        if any(x is None for x in v):
            return None
        return {"+": lambda: v[0] + v[1], "-": lambda: v[0] - v[1], "*": lambda: v[0] * v[1], "neg": lambda: -v[0]}[n.op]()
    return None


def _persistent_rounding(g: FrameGraph, plan: Plan, c: Component, mem) -> bool:
    """A recurrence y = y + b with a fractional step keeps every rounding error it ever made (coefficient exactly 1: nothing
    decays), and scripts put thresholds exactly where such sums are meant to land -- `pos += 1 / N; pos < 1 ? ...` reaches
    1 after N steps only up to rounding, so the frame at which the test flips depends on the ORDER of the additions. A scan
    re-associates them. Such components therefore keep their serial loop (exact order); integer-valued steps (counters,
    hold timers) are exact in any order and stay scans, and |a| < 1 forgets its rounding, so thresholds on it are generic.
    Decided on the branch-wise affine forms of the new state: (coefficient on itself, constant term) per path through ?: /
    min / max; any path with coefficient 1 and a constant term that is not an integer literal marks the component."""
    nm = c.names[0]
    limit = 256

    def forms(n: N):
        if n.i not in mem:
            return [(g.ZERO, n)]
        if n.kind == "st":
            return [(g.ONE, g.ZERO)]
        if n.kind != "op":
            return None
        if n.op == "sel":
            a, b = forms(n.args[1]), forms(n.args[2])
            return None if a is None or b is None or len(a) + len(b) > limit else a + b
        if n.op in ("min", "max"):
            a, b = forms(n.args[0]), forms(n.args[1])
            return None if a is None or b is None or len(a) + len(b) > limit else a + b
        if n.op == "fabs":
            a = forms(n.args[0])
            return None if a is None else a + [(g.op("neg", k), g.op("neg", v)) for k, v in a]
        if n.op in ("+", "-"):
            a, b = forms(n.args[0]), forms(n.args[1])
            if a is None or b is None or len(a) * len(b) > limit:
                return None
            return [(g.op(n.op, ka, kb), g.op(n.op, va, vb)) for ka, va in a for kb, vb in b]
        if n.op == "neg":
            a = forms(n.args[0])
            return None if a is None else [(g.op("neg", k), g.op("neg", v)) for k, v in a]
        if n.op == "*":
            a, b = forms(n.args[0]), forms(n.args[1])
            if a is None or b is None or len(a) * len(b) > limit:
                return None
            out = []
            for ka, va in a:
                for kb, vb in b:
                    if _const_value(ka) == 0.0:
                        out.append((g.op("*", va, kb), g.op("*", va, vb)))
                    elif _const_value(kb) == 0.0:
                        out.append((g.op("*", ka, vb), g.op("*", va, vb)))
                    else:
                        return None
            return out
        if n.op == "/":
            a, b = forms(n.args[0]), forms(n.args[1])
            if a is None or b is None or any(_const_value(kb) != 0.0 for kb, _ in b) or len(a) * len(b) > limit:
                return None
            return [(g.op("/", ka, vb), g.op("/", va, vb)) for ka, va in a for _, vb in b]
        return None

    fs = forms(plan.outs[nm])
    if fs is None:
        return False                              # not affine even branch-wise: the classification below decides
    for k, v in fs:
        if _const_value(k) == 1.0:
            cv = _const_value(v)
            if cv is None or cv != math.floor(cv):
                return True
    return False


def try_plan(prog: Program, nch: int) -> Tuple[Optional[Plan], str]:
    try:
        return build_plan(prog, nch), ""
    except Unsupported as ex:
        return None, str(ex)


# ----------------------------------------------------------------------------------------------------------------------
# 3. HIP emission (csrc/zart_tpar.h holds the wavefront primitives)
# ----------------------------------------------------------------------------------------------------------------------
_INFIX = {"+": "+", "-": "-", "*": "*", "/": "/"}
_CMP = {"<": "<", "<=": "<=", ">": ">", ">=": ">=", "==": "=="}
_FN2 = {"^": "pow", "|": "za_or", "&": "za_and", "~": "za_xor", "<<": "za_shl", ">>": "za_shr", "%": "za_mod", "!=": "za_ne",
        "min": "za_min", "max": "za_max", "pow": "pow", "atan2": "atan2"}
_POW_BASE = {"10.0": "exp10", "2.0": "exp2", c_double(math.e): "exp"}
_FN1 = {"neg": "za_neg", "not": "za_not", "sqr": "za_sqr", "sign": "za_sign", "invsqrt": "za_invsqrt"}


def _expr(op: str, a: List[str]) -> str:
    """Same C++ spelling as zajit/emit.py gives the construct, so both kernels share zart.h's semantics."""
    if op in _INFIX:
        return f"({a[0]} {_INFIX[op]} {a[1]})"
    if op in _CMP:
        return f"za_b({a[0]} {_CMP[op]} {a[1]})"
    if op in ("^", "pow") and a[0] in _POW_BASE and not os.environ.get("ZA_TPAR_PLAIN_POW"):
        # constant base: the dedicated exponential (68 instructions on gfx950) instead of the general pow (240); both are
        # accurate to the last bits, so results agree to ~4e-16 relative -- 10^(dB/20) is the commonest libm call in the catalog
        return f"{_POW_BASE[a[0]]}({a[1]})"
    if op in _FN2:
        return f"{_FN2[op]}({a[0]}, {a[1]})"
    if op in _FN1:
        return f"{_FN1[op]}({a[0]})"
    if op == "truth":
        return f"za_b(za_truthy({a[0]}))"
    if op == "land":
        return f"za_b(za_truthy({a[0]}) && za_truthy({a[1]}))"
    if op == "lor":
        return f"za_b(za_truthy({a[0]}) || za_truthy({a[1]}))"
    if op == "sel":
        return f"(za_truthy({a[0]}) ? {a[1]} : {a[2]})"
    if op in PURE_MATH1:
        return f"{PURE_MATH1[op]}({a[0]})"
    if op == "mtout":
        return f"zt_mt_word(zt_mt, zt_pos0, {a[0]})"
    raise AssertionError(op)


def emit_hip(plan: Plan, prog: Program, kernel_macro: str = "ZA_KERNEL(tpar)") -> str:
    """Kernel + launcher text, appended to a leaf module after zab_generic.hip.h (which defines ZabBatch / ZabAudio)."""
    g = plan.g
    L: List[str] = []

    def ref(n: N) -> str:
        if n.kind == "const":
            return c_double(n.val)
        return (f"u{n.i}" if n.uniform else f"n{n.i}")

    def inv_src(name: str) -> str:
        k = is_slider_name(name)
        if k is not None:
            return f"b.sliders[{k - 1} * b.sl_se + inst * b.sl_si]"
        if name == "srate":
            return "b.srate"
        if name in ("midi_bus", "ext_midi_bus", RNG_INDEX):
            return "0.0"
        k = is_spl_name(name)
        if k is not None:
            return f"b.spl[{k} * b.sl_se + inst * b.sl_si]"
        return f"b.vars[{prog.vars[name]} * b.var_se + inst * b.var_si]"

    def dst(name: str) -> str:
        k = is_spl_name(name)
        if k is not None:
            return f"b.spl[{k} * b.sl_se + inst * b.sl_si]"
        return f"b.vars[{prog.vars[name]} * b.var_se + inst * b.var_si]"

    cname = {name: f"c{k}" for k, name in enumerate(plan.st)}
    L.append("// ---- time-parallel kernel: one wavefront per instance, lane = frame (generated by zajit/tpar.py) ----")
    L.append(f"// schedule: {plan.stats}")
    L.append("#ifndef ZT_SPEC_MAX")
    L.append(f"#define ZT_SPEC_MAX {SPEC_MAX}")
    L.append("#endif")
    L.append(f"#define ZT_SPEC_TOL {SPEC_TOL!r}")
    L.append("#ifndef ZT_UNI")
    L.append("#define ZT_UNI(x) zt_uniform(x)")
    L.append("#endif")
    L.append(f'extern "C" __global__ void __launch_bounds__(64) {kernel_macro}(ZabBatch b, ZabAudio a) {{')
    L.append("  const int lane = threadIdx.x;")
    L.append("  const int64_t inst = blockIdx.x;")
    L.append("  const int64_t frames = a.frames;")
    L.append("  if (frames <= 0 || inst >= b.n_inst) return;")
    if plan.uses_rand:
        L.append("  __shared__ uint32_t zt_mt[2 * ZT_MT_N];      // rand(): current and next generation of the instance's MT19937")
        L.append("  uint32_t* const zt_gmt = b.mt + inst * b.mt_si;")
        L.append("  int zt_pos0 = zt_mt_begin(zt_mt, zt_gmt, b.mt_se, b.mti[inst], lane);")
    L.append("  // per launch: invariants and everything that depends on them only")
    for n in plan.uniform:
        if n.kind == "const":
            continue
        if n.kind == "inv":
            L.append(f"  const double u{n.i} = {inv_src(n.name)};   // {n.name}")
        else:
            L.append(f"  const double u{n.i} = ZT_UNI({_expr(n.op, [ref(x) for x in n.args])});")
    # recurrences whose coefficient is constant over the launch: one LDS row of per-lane weights per distinct coefficient
    inv_coefs: List[N] = []
    for it in plan.items:
        if it[0] == "scan" and len(it[1].names) == 1:
            a = it[1].A[0][0]
            if a.uniform and a.kind != "const" and a not in inv_coefs and not os.environ.get("ZA_TPAR_NO_INVSCAN"):
                inv_coefs.append(a)
    # coupled pairs with a launch-constant matrix (biquads): one table per distinct matrix, within an LDS budget that still
    # lets four wavefronts share a CU (one per SIMD, the 1024-instance case)
    inv_mats: List[tuple] = []
    budget = 36 * 1024 - len(inv_coefs) * (64 + 4) * 8 - (2 * 624 * 4 if plan.uses_rand else 0)
    for it in plan.items:
        if it[0] == "scan" and len(it[1].names) == 2 and not os.environ.get("ZA_TPAR_NO_INVSCAN"):
            key = tuple(x for row in it[1].A for x in row)
            if all(x.uniform or x.kind == "const" for x in key) and key not in inv_mats and (len(inv_mats) + 1) * (12 + 8 * 64) * 8 <= budget:
                inv_mats.append(key)
    if inv_mats:
        L.append(f"  __shared__ double zt_m[{len(inv_mats)} * ZT_MAT_TABLE_DOUBLES];      // per launch-constant 2 x 2 matrix: powers and per-lane weights")
        for k, key in enumerate(inv_mats):
            L.append(f"  {{ const ZtMat2 am = {{{ref(key[0])}, {ref(key[1])}, {ref(key[2])}, {ref(key[3])}}}; zt_mat_table(zt_m + {k} * ZT_MAT_TABLE_DOUBLES, am, lane); }}")
        if not inv_coefs:
            L.append("  __syncthreads();")
    if inv_coefs:
        L.append(f"  __shared__ double zt_w[{len(inv_coefs)} * 64];      // a^((lane & 15) + 1) per launch-constant coefficient")
        L.append(f"  __shared__ double zt_q[{len(inv_coefs)} * 4];       // a^2, a^4, a^8, a^16")
        for k, a in enumerate(inv_coefs):
            L.append(f"  zt_w[{k} * 64 + lane] = zt_pow_row({ref(a)}, lane);")
            L.append(f"  if (lane == 0) {{ const double p2 = {ref(a)} * {ref(a)}, p4 = p2 * p2, p8 = p4 * p4; zt_q[{k} * 4] = p2; zt_q[{k} * 4 + 1] = p4; zt_q[{k} * 4 + 2] = p8; zt_q[{k} * 4 + 3] = p8 * p8; }}")
        L.append("  __syncthreads();")
    L.append("  // state carried from frame to frame (wave-uniform)")
    for name, c in cname.items():
        L.append(f"  double {c} = {inv_src(name)};   // {name}")
    L.append(f"  const float* const in_ = a.in + inst * {plan.nch} * a.frame_stride;")
    L.append(f"  float* const out_ = a.out + inst * {plan.nch} * a.frame_stride;")
    L.append("  // the audio of a chunk is read one iteration ahead, so that its HBM latency is hidden behind the previous chunk's work")
    for n in plan.inputs:
        L.append(f"  float x{n.i} = lane < frames ? in_[{int(n.val)} * a.frame_stride + lane] : 0.0f;")
    L.append("  for (int64_t f0 = 0; f0 < frames; f0 += 64) {")
    L.append("    const int tn = (int)(frames - f0 < 64 ? frames - f0 : 64);")
    L.append("    const int last = tn - 1;")
    L.append("    const bool valid = lane < tn;")
    for n in plan.inputs:
        L.append(f"    const double n{n.i} = (double)x{n.i};")
    L.append("    if (f0 + 64 + lane < frames) {")
    for n in plan.inputs:
        L.append(f"      x{n.i} = in_[{int(n.val)} * a.frame_stride + f0 + 64 + lane];")
    L.append("    } else {")
    for n in plan.inputs:
        L.append(f"      x{n.i} = 0.0f;")
    L.append("    }")

    def serial_loop(comps: List[Component], ind: str):
        """64 uniform steps; leaves the state before each frame in k<st> of that frame's lane."""
        for c in comps:
            for nm in c.names:
                s = plan.st[nm].i
                L.append(f"{ind}double y{s} = {cname[nm]}, k{s} = {cname[nm]};")
        L.append(f"{ind}for (int t = 0; t < tn; ++t) {{")
        L.append(f"{ind}  const bool me = lane == t;")
        seen_ext = set()
        for c in comps:
            mem = {m.i for m in c.members}
            for nm in c.names:
                s = plan.st[nm].i
                L.append(f"{ind}  k{s} = me ? y{s} : k{s};")
            for x in c.ext:
                if not x.uniform and x.kind != "const" and x.i not in seen_ext:
                    seen_ext.add(x.i)
                    L.append(f"{ind}  const double e{x.i} = zt_readlane(n{x.i}, t);")

            def sref(x: N, mem=mem) -> str:
                if x.kind == "st" and x.i in mem:
                    return f"y{x.i}"
                if x.i in mem:
                    return f"m{x.i}"
                if x.kind == "const" or x.uniform:
                    return ref(x)
                return f"e{x.i}"

            for m in c.members:
                if m.kind == "st":
                    continue
                L.append(f"{ind}  const double m{m.i} = {_expr(m.op, [sref(x) for x in m.args])};")
            for nm in c.names:            # all new states are computed from the old ones before any is replaced
                L.append(f"{ind}  const double q{plan.st[nm].i} = {sref(plan.outs[nm])};")
            for nm in c.names:
                L.append(f"{ind}  y{plan.st[nm].i} = q{plan.st[nm].i};")
        L.append(f"{ind}}}")

    # Values leave the registers as early as possible: a state's carry is taken (v_readlane at the chunk's last frame) as soon
    # as both its recurrence and its new value exist, and the values a launch must leave in vars[] -- needed in the launch's
    # last chunk only -- are stored in small conditional batches right after they are computed, instead of all living to the
    # end of the chunk body (144 written variables would be 288 registers per lane there).
    if inv_coefs or inv_mats:
        L.append("    int zo; asm volatile(\"s_mov_b32 %0, 0\" : \"=s\"(zo));   // opaque 0: keeps the table reads inside the iteration")
    L.append("    const bool fin = f0 + 64 >= frames;   // the launch's last chunk: its last frame leaves every written variable as the script would")
    avail = {n.i for n in plan.inputs}
    unit_done: set = set()
    carried: set = set()
    stored: set = set()
    finals = [(name, o) for name, o in plan.outs.items() if name != RNG_INDEX]
    finals += [(f"spl{ch}", plan.spl_out[ch]) for ch in range(plan.nch) if f"spl{ch}" not in plan.outs]
    pending: List[tuple] = []

    def ready(o: N) -> bool:
        return o.uniform or o.kind == "const" or o.i in avail

    def retire(final: bool = False):
        for name, c in cname.items():
            o = plan.outs[name]
            if name not in carried and name in unit_done and ready(o):
                carried.add(name)
                L.append(f"    {c} = {ref(o) if (o.uniform or o.kind == 'const') else f'zt_readlane(n{o.i}, last)'};")
        for name, o in finals:
            if name not in stored and ready(o) and (final or not (o.uniform or o.kind == "const")):
                stored.add(name)
                pending.append((name, o))
        if pending and (final or len(pending) >= 12):
            L.append("    if (fin && lane == last) {")
            for name, o in pending:
                L.append(f"      {dst(name)} = {ref(o)};")
            L.append("    }")
            pending.clear()

    for gid, it in enumerate(plan.items):
        kind = it[0]
        if kind == "par":
            avail.add(it[1].i)
        elif kind == "shift":
            avail.add(plan.st[it[1]].i)
            unit_done.add(it[1])
        elif kind == "scan":
            for nm in it[1].names:
                avail.add(plan.st[nm].i)
                unit_done.add(nm)
        else:
            for c in it[1]:
                for nm in c.names:
                    avail.add(plan.st[nm].i)
                    unit_done.add(nm)
        if gid:
            pass
        if kind == "par":
            n = it[1]
            L.append(f"    const double n{n.i} = {_expr(n.op, [ref(x) for x in n.args])};")
        elif kind == "shift":
            name = it[1]
            L.append(f"    const double n{plan.st[name].i} = zt_shift1({ref(plan.outs[name])}, {cname[name]});   // {name}[t-1]")
        elif kind == "scan":
            c: Component = it[1]
            if len(c.names) == 1 and c.A[0][0].kind == "const" and c.A[0][0].val == 1.0:
                nm = c.names[0]
                s = plan.st[nm].i
                L.append(f"    const double n{s} = zt_shift1(zt_scan1_sum({ref(c.b[0])}, {cname[nm]}, lane), {cname[nm]});   // {nm}: running sum")
            elif len(c.names) == 1 and c.A[0][0] in inv_coefs:
                nm = c.names[0]
                s = plan.st[nm].i
                k = inv_coefs.index(c.A[0][0])
                L.append(f"    const ZtPow sq{s} = {{zt_q[{k} * 4 + zo], zt_q[{k} * 4 + 1 + zo], zt_q[{k} * 4 + 2 + zo], zt_q[{k} * 4 + 3 + zo]}};   // {nm}: constant-coefficient recurrence")
                L.append(f"    const double n{s} = zt_shift1(zt_scan1_inv({ref(c.b[0])}, {ref(c.A[0][0])}, sq{s}, zt_w[{k} * 64 + lane + zo], {cname[nm]}, lane), {cname[nm]});")
            elif len(c.names) == 1:
                nm = c.names[0]
                s = plan.st[nm].i
                L.append(f"    double sa{s} = {ref(c.A[0][0])}, sb{s} = {ref(c.b[0])};   // {nm}: affine recurrence")
                L.append(f"    zt_scan1(sa{s}, sb{s});")
                L.append(f"    const double n{s} = zt_shift1(__builtin_fma(sa{s}, {cname[nm]}, sb{s}), {cname[nm]});")
            elif len(c.names) == 2 and tuple(x for row in c.A for x in row) in inv_mats:
                n0, n1 = c.names
                s0, s1 = plan.st[n0].i, plan.st[n1].i
                k = inv_mats.index(tuple(x for row in c.A for x in row))
                L.append(f"    double sb{s0} = {ref(c.b[0])}, sb{s1} = {ref(c.b[1])};   // {n0}, {n1}: coupled pair, launch-constant matrix")
                L.append(f"    {{ const ZtMat2 am = {{{ref(c.A[0][0])}, {ref(c.A[0][1])}, {ref(c.A[1][0])}, {ref(c.A[1][1])}}};")
                L.append(f"      zt_scan2_inv(sb{s0}, sb{s1}, am, zt_m + {k} * ZT_MAT_TABLE_DOUBLES, zo, {cname[n0]}, {cname[n1]}, lane); }}")
                L.append(f"    const double n{s0} = zt_shift1(sb{s0}, {cname[n0]});")
                L.append(f"    const double n{s1} = zt_shift1(sb{s1}, {cname[n1]});")
            else:
                n0, n1 = c.names
                s0, s1 = plan.st[n0].i, plan.st[n1].i
                L.append(f"    ZtMap2 sm{s0} = {{{ref(c.A[0][0])}, {ref(c.A[0][1])}, {ref(c.A[1][0])}, {ref(c.A[1][1])}, {ref(c.b[0])}, {ref(c.b[1])}}};   // {n0}, {n1}: coupled affine pair")
                L.append(f"    zt_scan2(sm{s0});")
                L.append(f"    const double n{s0} = zt_shift1(__builtin_fma(sm{s0}.a00, {cname[n0]}, __builtin_fma(sm{s0}.a01, {cname[n1]}, sm{s0}.b0)), {cname[n0]});")
                L.append(f"    const double n{s1} = zt_shift1(__builtin_fma(sm{s0}.a10, {cname[n0]}, __builtin_fma(sm{s0}.a11, {cname[n1]}, sm{s0}.b1)), {cname[n1]});")
        elif kind == "serial":
            names = [nm for c in it[1] for nm in c.names]
            L.append(f"    // serial recurrences sharing one loop: {', '.join(names)}")
            serial_loop(it[1], "    ")
            for nm in names:
                s = plan.st[nm].i
                L.append(f"    const double n{s} = k{s};")
        elif kind == "spec":
            comps: List[Component] = it[1]
            names = [nm for c in comps for nm in c.names]
            L.append(f"    // switched recurrences (affine once their state-dependent conditions are fixed), solved by iterating the")
            L.append(f"    // condition pattern to its fixed point: {', '.join(names)}")
            for nm in names:
                L.append(f"    double s{plan.st[nm].i} = {cname[nm]}, p{plan.st[nm].i} = {cname[nm]};")
            gname = {}
            for c in comps:
                for k, gn in enumerate(c.gnodes):
                    gname[gn.i] = f"g{gn.name}_{k}"
                    L.append(f"    bool {gname[gn.i]};")

            def xref(x: N, loc: Dict[int, str]) -> str:
                if x.i in loc:
                    return loc[x.i]
                if x.kind == "guess":
                    return f"({gname[x.i]} ? 1.0 : 0.0)"
                return ref(x)

            def slice_eval(c: Component, ind: str, out_prefix: str):
                loc = {plan.st[nm].i: f"s{plan.st[nm].i}" for nm in c.names}
                for m in c.slice:
                    loc[m.i] = f"v{m.i}"
                    L.append(f"{ind}const double v{m.i} = {_expr(m.op, [xref(x, loc) for x in m.args])};")
                for k, (cn, gn) in enumerate(zip(c.conds, c.gnodes)):
                    L.append(f"{ind}{out_prefix}{gname[gn.i]} = za_truthy({xref(cn, loc)});")

            L.append("    {   // first pattern: the states taken to stay at their carried values")
            for c in comps:
                slice_eval(c, "      ", "")
            L.append("    }")
            L.append(f"    bool sch{gid}; int sit{gid} = 0;")
            L.append("    do {")
            for c in comps:
                loc: Dict[int, str] = {}
                for n in c.gdep:
                    loc[n.i] = f"d{n.i}"
                    L.append(f"      const double d{n.i} = {_expr(n.op, [xref(x, loc) for x in n.args])};")
                if len(c.names) == 1:
                    nm = c.names[0]
                    s = plan.st[nm].i
                    L.append(f"      double sa{s} = {xref(c.A[0][0], loc)}, sb{s} = {xref(c.b[0], loc)};")
                    L.append(f"      zt_scan1(sa{s}, sb{s});")
                    L.append(f"      s{s} = zt_shift1(__builtin_fma(sa{s}, {cname[nm]}, sb{s}), {cname[nm]});")
                else:
                    n0, n1 = c.names
                    s0, s1 = plan.st[n0].i, plan.st[n1].i
                    L.append(f"      ZtMap2 sm{s0} = {{{xref(c.A[0][0], loc)}, {xref(c.A[0][1], loc)}, {xref(c.A[1][0], loc)}, {xref(c.A[1][1], loc)}, {xref(c.b[0], loc)}, {xref(c.b[1], loc)}}};")
                    L.append(f"      zt_scan2(sm{s0});")
                    L.append(f"      s{s0} = zt_shift1(__builtin_fma(sm{s0}.a00, {cname[n0]}, __builtin_fma(sm{s0}.a01, {cname[n1]}, sm{s0}.b0)), {cname[n0]});")
                    L.append(f"      s{s1} = zt_shift1(__builtin_fma(sm{s0}.a10, {cname[n0]}, __builtin_fma(sm{s0}.a11, {cname[n1]}, sm{s0}.b1)), {cname[n1]});")
            L.append("      // the pattern these states imply")
            for c in comps:
                slice_eval(c, "      ", "const bool h")
            diffs = " || ".join(f"(h{gname[gn.i]} != {gname[gn.i]})" for c in comps for gn in c.gnodes)
            moved = " || ".join(f"(fabs(s{plan.st[nm].i} - p{plan.st[nm].i}) > ZT_SPEC_TOL * fmax(fabs(s{plan.st[nm].i}), fabs(p{plan.st[nm].i})))" for nm in names)
            L.append("      // settled = the pattern reproduced itself, or it only still moves where its branches agree (states unchanged)")
            L.append(f"      sch{gid} = (__ballot(valid && ({diffs})) != 0ull) && (__ballot(valid && ({moved})) != 0ull);")
            for c in comps:
                for gn in c.gnodes:
                    L.append(f"      {gname[gn.i]} = h{gname[gn.i]};")
            for nm in names:
                L.append(f"      p{plan.st[nm].i} = s{plan.st[nm].i};")
            L.append(f"    }} while (sch{gid} && ++sit{gid} < ZT_SPEC_MAX);")
            L.append(f"    if (sch{gid}) {{   // no fixed point within the budget (a pattern that keeps moving along the chunk): the serial loop")
            serial_loop(comps, "      ")
            for nm in names:
                s = plan.st[nm].i
                L.append(f"      s{s} = k{s};")
            L.append("    }")
            for nm in names:
                s = plan.st[nm].i
                L.append(f"    const double n{s} = s{s};")
        else:
            raise AssertionError(kind)
        retire()
    retire(final=True)
    L.append("    if (valid) {")
    for ch in range(plan.nch):
        L.append(f"      out_[{ch} * a.frame_stride + f0 + lane] = (float){ref(plan.spl_out[ch])};")
    L.append("    }")
    if plan.uses_rand:
        L.append(f"    zt_mt_retire(zt_mt, zt_pos0, (int){cname[RNG_INDEX]}, lane);")
    L.append("  }")
    if plan.uses_rand:
        L.append(f"  zt_mt_end(zt_mt, zt_pos0, (int){cname[RNG_INDEX]}, zt_gmt, b.mt_se, b.mti + inst, lane);")
    L.append("}")
    L.append("static int32_t za_fast_applies(const ZabBatch* b, const ZabAudio* a) { (void)b; return a->frames > 0 ? 1 : 0; }")
    L.append("static hipError_t za_launch_fast(const ZabBatch* b, const ZabAudio* a, hipStream_t st) {")
    L.append(f"  hipLaunchKernelGGL({kernel_macro}, dim3(b->n_inst), dim3(64), 0, st, *b, *a);")
    L.append("  return hipGetLastError();")
    L.append("}")
    return "\n".join(L) + "\n"
