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
    __slots__ = ("i", "kind", "op", "args", "val", "name", "uniform", "su", "extra")

    def __init__(self, i, kind, op=None, args=(), val=None, name=None):
        self.i, self.kind, self.op, self.args, self.val, self.name = i, kind, op, tuple(args), val, name
        self.uniform = False
        self.su = False          # built from constants and variables @sample never assigns: constant over a launch (known while walking)
        self.extra = ()          # ld: nodes this load must wait for besides its address (the stores it may have to forward from)

    def __repr__(self):
        if self.kind == "const":
            return f"#{self.i}:{self.val!r}"
        if self.kind in ("var", "inv", "st", "in"):
            return f"#{self.i}:{self.kind}({self.name})"
        if self.kind == "ld":
            return f"#{self.i}:ld({self.args[0].i})"
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
        # variables assigned anywhere in @sample (or a function it can reach): everything else is constant over a launch, which
        # lets the walk tell launch-constant addresses (mem[] cells used as named state) from moving ones (delay lines)
        self.wsyn = set()
        todo = list(prog.sections.get("sample", [])) + [f.body for f in prog.fns.values()]
        while todo:
            x = todo.pop()
            if isinstance(x, S.Assign) and isinstance(x.target, S.Var):
                self.wsyn.add(x.target.name)
            todo.extend(S.children(x))
        self.pred: Optional[N] = None            # path condition of the statement being walked (None: unconditional)
        self.mem_seq = 0                         # program order of the memory operations of a frame
        self.cells: Dict[str, N] = {}            # "mem@<id>" -> its (launch-constant) address node
        self.loads: List[N] = []                 # moving-address loads
        self.stores: List["StoreSite"] = []      # moving-address stores
        self.ZERO, self.ONE = self.const(0.0), self.const(1.0)

    # -- node construction -------------------------------------------------------------------------------------------------
    def mk(self, kind, op=None, args=(), val=None, name=None) -> N:
        key = (kind, op, tuple(a.i for a in args), repr(val), name)
        n = self.memo.get(key)
        if n is None:
            n = N(len(self.nodes), kind, op, args, val, name)
            if kind == "const":
                n.su = True
            elif kind == "var":
                n.su = name not in self.wsyn and name != RNG_INDEX and not name.startswith("mem@") and not name.startswith("memw@")
            elif kind == "op":
                n.su = op != "mtout" and all(a.su for a in args)
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
            if (is_slider_name(key) is None and key not in ("srate", "midi_bus", "ext_midi_bus", RNG_INDEX) and key not in self.p.vars
                    and key not in self.cells and not key.startswith("memw@")):
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
            if k is None and key not in self.p.vars and key != RNG_INDEX and key not in self.cells and not key.startswith("memw@"):
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

    # -- mem[] ---------------------------------------------------------------------------------------------------------------
    def _address(self, n) -> N:
        if isinstance(n.base, S.Var) and n.base.name == "gmem":
            raise Unsupported("gmem[] access in @sample")
        b = self.ev(n.base)
        i = self.ev(n.index)
        return self.op("addr", b, i)              # za_addr(base, index) of csrc/zart.h, as a double

    def _region(self, a: N) -> tuple:
        """Launch-constant terms of base + index: accesses that differ in them are taken to address different buffers (checked
        at run time, chunk by chunk: a load that falls into another buffer's freshly written span aborts the fast path)."""
        terms, todo = [], list(a.args)
        while todo:
            x = todo.pop()
            if x.kind == "op" and x.op == "+":
                todo.extend(x.args)
            elif x.su and not (x.kind == "const" and x.val == 0.0):
                terms.append(x.i)
        return tuple(sorted(terms))

    def _cell(self, a: N) -> str:
        key = f"mem@{a.i}"
        self.cells[key] = a
        return key

    def _load(self, a: N) -> N:
        if a.su:                                   # a cell: mem[] used as a named state variable
            return self.read(self._cell(a))
        self.mem_seq += 1
        ld = self.mk("ld", args=(a,), val=self.mem_seq)
        self.loads.append(ld)
        return ld

    def _store(self, a: N, v: N):
        if a.su:
            key = self._cell(a)
            self.write(key, v)
            # "has this cell been stored to in this launch": the write high-water mark of the arena moves only for executed
            # stores, and a store under a condition may never run. (An ordinary state: its updates merge like any variable's.)
            self.write("memw@" + key[4:], self.ONE)
            return
        self.mem_seq += 1
        self.stores.append(StoreSite(len(self.stores), a, v, self.pred, self.mem_seq, self._region(a)))

    def v_Index(self, n):
        return self._load(self._address(n))

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
            env0, pred0 = self.env, self.pred
            self.env = dict(env0)
            gate = l if n.op == "&&" else self.op("not", l)          # the right operand runs iff ...
            self.pred = gate if pred0 is None else self.op("land", pred0, gate)
            r = self.ev(n.r)
            self.pred = pred0
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
        env0, pred0 = self.env, self.pred
        self.env = dict(env0)
        self.pred = c if pred0 is None else self.op("land", pred0, c)
        vt = self.ev(then_ast) if then_ast is not None else self.ZERO
        env_t = self.env
        self.env = dict(env0)
        nc = self.op("not", c)
        self.pred = nc if pred0 is None else self.op("land", pred0, nc)
        ve = self.ev(else_ast) if else_ast is not None else self.ZERO
        env_e = self.env
        self.pred = pred0
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
        if isinstance(tgt, S.Index):               # value first, then base and index (zajit/emit.py e_Assign)
            a = self._address(tgt)
            if n.op == "=":
                val = rhs
            else:
                bop = n.op[:-1]
                if bop not in BIN_OPS:
                    raise Unsupported(f"assignment operator {n.op}")
                val = self.op(bop, self._load(a), rhs)
            self._store(a, val)
            return val
        if not isinstance(tgt, S.Var):
            raise Unsupported("assignment to slider() / spl() in @sample")
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
class StoreSite:
    """One moving-address store of the frame (a delay line's write)."""

    def __init__(self, j, addr, value, pred, seq, region):
        self.j, self.addr, self.value, self.pred, self.seq, self.region = j, addr, value, pred, seq, region


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
        self.cells: Dict[str, N] = {}            # "mem@<id>" -> launch-constant address node (mem[] used as named state)
        self.stores: List[StoreSite] = []        # delay-line writes (moving addresses), program order
        self.loads: List[N] = []                 # delay-line reads

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

    def simulate(self, vars0: Dict[str, float], x: np.ndarray, sliders=None, srate=48000.0, spl0=None, mt=None, mem=None):
        """x: [nch, frames] float32. vars0: name -> value before the launch (missing names are 0). mt: (randMT[624], randIndex)
        before the launch for scripts that call rand(); self.mt_after holds the pair after it. mem: the arena before the launch
        (numpy doubles) for scripts that touch mem[]; self.mem_after / self.mem_high_after hold it after.
        Returns (y float32 [nch, frames], vars after {name: value}, spl after {k: value}). Raises TparAbort when a chunk breaks
        one of the run-time conditions of the delay-line handling (the kernel hands such a launch to the generic kernel)."""
        memv = np.zeros(1 << 16) if mem is None else np.array(mem, dtype=np.float64)
        mem_high = 0
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
            if name in ("midi_bus", "ext_midi_bus", RNG_INDEX) or name.startswith("memw@"):
                return 0.0
            if name in self.cells:
                a = int(val[self.cells[name].i])
                return float(memv[a]) if a < len(memv) else 0.0
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
            cell_addr = {name: int(val[a.i]) for name, a in self.cells.items()}
            if len(set(cell_addr.values())) != len(cell_addr) or any(a >= len(memv) for a in cell_addr.values()):
                raise TparAbort(0, "mem[] cells alias each other or lie past the arena")
            carry = {name: np.float64(inv_value(name)) for name in self.st}
            sites: Dict[int, dict] = {}
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
                sites.clear()
                for it in self.items:
                    kind = it[0]
                    if kind == "site":
                        st_: StoreSite = it[1]
                        A = np.broadcast_to(val[st_.addr.i], (WAVE,)).astype(np.int64)
                        d = np.diff(A[:tn])
                        brk = np.flatnonzero(d != 1)
                        if len(brk) > 1 or A[:tn].min() < 0 or A[:tn].max() >= len(memv):
                            raise TparAbort(f0, "a delay-line write does not advance by one cell per frame (or leaves the arena)")
                        k = int(brk[0]) + 1 if len(brk) else tn
                        sites[st_.j] = {"A": A, "a0": int(A[0]), "k": k, "ak": int(A[k]) if k < tn else 0}
                        if any(lo <= a <= hi for a in cell_addr.values() for lo, hi in ((A[:tn].min(), A[:tn].max()),)):
                            raise TparAbort(f0, "a delay line runs over a mem[] cell")
                    elif kind == "par" and it[1].kind == "ld":
                        n = it[1]
                        B = np.broadcast_to(val[n.args[0].i], (WAVE,)).astype(np.int64)
                        out = np.where(B < len(memv), memv[np.minimum(B, len(memv) - 1)], 0.0)
                        best = np.full(WAVE, -1)
                        for st_ in self.stores:
                            si = sites[st_.j]
                            tw = np.full(WAVE, -1)
                            d0 = B - si["a0"]
                            tw = np.where((d0 >= 0) & (d0 < si["k"]), d0, tw)
                            d1 = B - si["ak"]
                            tw = np.where((d1 >= 0) & (d1 < tn - si["k"]), si["k"] + d1, tw)
                            if ",".join(map(str, st_.region)) != n.name:
                                if np.any(tw[:tn] >= 0):
                                    raise TparAbort(f0, "a delay-line read falls into another buffer's freshly written span")
                                continue
                            vis = (tw >= 0) & ((tw < lane) | ((tw == lane) & (st_.seq < n.val))) & (tw >= best)
                            V = np.broadcast_to(val[st_.value.i], (WAVE,)).astype(np.float64)
                            out = np.where(vis, V[np.clip(tw, 0, WAVE - 1)], out)
                            best = np.where(vis, tw, best)
                        if any(np.any(B[:tn] == a) for a in cell_addr.values()):
                            raise TparAbort(f0, "a delay-line read hits a mem[] cell")
                        val[n.i] = out
                    elif kind == "par":
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
                for st_ in self.stores:                    # the chunk's writes land after all of its reads are resolved
                    si = sites[st_.j]
                    memv[si["A"][:tn]] = np.broadcast_to(val[st_.value.i], (WAVE,))[:tn]
                    mem_high = max(mem_high, int(si["A"][:tn].max()) + 1)
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
            if name.startswith("memw@"):
                continue
            if name in self.cells:
                if final_vals.get("memw@" + name[4:], 0.0) != 0.0:       # stored to at least once in this launch
                    memv[cell_addr[name]] = v
                    mem_high = max(mem_high, cell_addr[name] + 1)
            elif k is not None:
                spl_after[k] = v
            else:
                vars_after[name] = v
        self.mem_after, self.mem_high_after = memv, mem_high
        return y, vars_after, spl_after


class TparAbort(Exception):
    """A chunk broke a run-time condition of the delay-line handling at frame `f0`; the kernel stops there and the generic
    kernel finishes the launch."""

    def __init__(self, f0, why):
        super().__init__(f"frame {f0}: {why}")
        self.f0, self.why = f0, why


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

ULDS_THRESHOLD = 64   # launch-constant values beyond which they live in LDS rather than in (spilled) scalar registers
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
    if op == "addr":          # za_addr: trunc(base + index + 1e-5), negatives (and NaN) to 0
        x = np.asarray(a[0], dtype=np.float64) + a[1] + 1.0e-5
        return np.where(x > 0.0, np.trunc(np.where(x > 0.0, x, 0.0)), 0.0)
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
    plan.cells = dict(g.cells)
    plan.stores, plan.loads = list(g.stores), list(g.loads)
    for st_ in plan.stores:
        if st_.pred is not None:
            raise Unsupported("conditional store to a moving mem[] address")
    if plan.stores and g.rand_sites:
        raise Unsupported("rand() together with delay lines")
    for ld in plan.loads:
        # a load may have to take its value from a store of this chunk: it waits for every store of its own buffer (address
        # and value) and, for the aliasing check, for the addresses of all the others
        reg = g._region(ld.args[0])
        ld.name = ",".join(map(str, reg))
        ld.extra = tuple(x for st_ in plan.stores for x in ((st_.addr, st_.value) if st_.region == reg else (st_.addr,)))

    # live nodes
    live: Dict[int, N] = {}
    todo = list(plan.outs.values()) + list(plan.spl_out) + [x for st_ in plan.stores for x in (st_.addr, st_.value)]
    todo += [a for a in plan.cells.values()]
    while todo:
        n = todo.pop()
        if n.i in live:
            continue
        live[n.i] = n
        todo.extend(n.args)
        todo.extend(n.extra)
        if n.kind == "st":
            todo.append(plan.outs[n.name])
    order = sorted(live)                     # creation order is a topological order of the in-frame edges
    pos = {i: k for k, i in enumerate(order)}
    succ: List[List[int]] = [[] for _ in order]
    for i in order:
        n = live[i]
        for a in n.args + n.extra:
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
        if any(m.kind == "ld" for m in members):
            raise Unsupported("feedback through a delay line (a stored value depends on a load of the same buffer)")
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
        elif n.kind in ("st", "in", "ld"):
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

    site_done: set = set()
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
            if all(a.i in done for a in n.args + n.extra):
                if n.kind == "ld":
                    for st_ in plan.stores:              # every write's span is known before the first read is resolved
                        if st_.j not in site_done:
                            site_done.add(st_.j)
                            items.append(("site", st_))
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
    for st_ in plan.stores:
        if st_.j not in site_done:
            items.append(("site", st_))
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
        "mem_cells": len(plan.cells), "delay_writes": len(plan.stores), "delay_reads": len(plan.loads),
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
    if op == "addr":
        return f"(double)za_addr({a[0]}, {a[1]})"
    raise AssertionError(op)


def emit_hip(plan: Plan, prog: Program, kernel_macro: str = "ZA_KERNEL(tpar)") -> str:
    """Kernel + launcher text, appended to a leaf module after zab_generic.hip.h (which defines ZabBatch / ZabAudio)."""
    g = plan.g
    L: List[str] = []

    # Launch-constant values: a few dozen fit the scalar registers (ZT_UNI); past that the compiler spills them into lanes of
    # vector registers and every use costs two v_readlane. Large scripts keep them in LDS instead: one broadcast ds_read_b64 per
    # use, the `zo` offset (an opaque 0 set per chunk) keeping the reads inside the iteration.
    n_uni = sum(1 for n in plan.uniform if n.kind != "const")
    mode = os.environ.get("ZA_TPAR_ULDS", "auto")
    ulds = mode == "1" or (mode == "auto" and n_uni > ULDS_THRESHOLD)
    uslot = {n.i: k for k, n in enumerate(x for x in plan.uniform if x.kind != "const")}
    in_loop = [False]

    def ref(n: N) -> str:
        if n.kind == "const":
            return c_double(n.val)
        if n.uniform and ulds and in_loop[0]:
            return f"zt_u[{uslot[n.i]} + zo]"
        return (f"u{n.i}" if n.uniform else f"n{n.i}")

    def inv_src(name: str) -> str:
        k = is_slider_name(name)
        if k is not None:
            return f"b.sliders[{k - 1} * b.sl_se + inst * b.sl_si]"
        if name == "srate":
            return "b.srate"
        if name in ("midi_bus", "ext_midi_bus", RNG_INDEX) or name.startswith("memw@"):
            return "0.0"
        if name in plan.cells:
            return f"(ca{plan.cells[name].i} < mcap ? memp[ca{plan.cells[name].i} * mse] : 0.0)"
        k = is_spl_name(name)
        if k is not None:
            return f"b.spl[{k} * b.sl_se + inst * b.sl_si]"
        return f"b.vars[{prog.vars[name]} * b.var_se + inst * b.var_si]"

    def dst(name: str) -> str:
        if name in plan.cells:
            return f"memp[ca{plan.cells[name].i} * mse]"
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
    if ulds:
        L.append(f"  __shared__ double zt_u[{max(1, n_uni)}];")
        L.append("  if (lane == 0) {")
        for n in plan.uniform:
            if n.kind != "const":
                L.append(f"    zt_u[{uslot[n.i]}] = u{n.i};")
        L.append("  }")
        L.append("  __syncthreads();")
    has_mem = bool(plan.cells or plan.stores or plan.loads)
    has_streams = bool(plan.stores)
    cell_addrs: List[N] = []
    for a in plan.cells.values():
        if a not in cell_addrs:
            cell_addrs.append(a)
    if has_mem:
        L.append("  // mem[]: launch-constant addresses are cells (named state kept in registers), moving ones are delay lines")
        L.append("  double* const memp = b.mem + inst * b.mem_si;")
        L.append("  const int64_t mse = b.mem_se, mcap = b.mem_cap;")
        L.append("  int64_t zt_high = b.mem_high[inst], zt_hc = 0;")
        for a in cell_addrs:
            L.append(f"  const int64_t ca{a.i} = (int64_t){ref(a)};")
        if cell_addrs:
            clash = " || ".join([f"ca{a.i} >= mcap" for a in cell_addrs] +
                                [f"ca{a.i} == ca{b_.i}" for i_, a in enumerate(cell_addrs) for b_ in cell_addrs[i_ + 1:]])
            lo = cell_addrs[0]
            L.append(f"  int64_t cmin = ca{lo.i}, cmax = ca{lo.i};")
            for a in cell_addrs[1:]:
                L.append(f"  cmin = ca{a.i} < cmin ? ca{a.i} : cmin; cmax = ca{a.i} > cmax ? ca{a.i} : cmax;")
            L.append(f"  if ({clash}) {{   // cells that alias each other (or lie past the arena): not a case for this kernel")
            L.append("    if (lane == 0) b.resume[inst] = 0;")
            L.append("    return;")
            L.append("  }")
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
    if has_streams:
        L.append(f"  __shared__ double zt_snap[{max(1, len(cname))}];")
    L.append("  // state carried from frame to frame (wave-uniform)")
    for name, c in cname.items():
        L.append(f"  double {c} = {inv_src(name)};   // {name}")
    L.append(f"  const float* const in_ = a.in + inst * {plan.nch} * a.frame_stride;")
    L.append(f"  float* const out_ = a.out + inst * {plan.nch} * a.frame_stride;")
    L.append("  // the audio of a chunk is read one iteration ahead, so that its HBM latency is hidden behind the previous chunk's work")
    for n in plan.inputs:
        L.append(f"  float x{n.i} = lane < frames ? in_[{int(n.val)} * a.frame_stride + lane] : 0.0f;")
    # ZT_PIN: an empty asm that takes the prefetched registers, i.e. the point where the compiler waits for their loads. It
    # sits before the loop and, in the loop, before the chunk's stores: the loads have had the whole chunk to land, and no
    # path reaches the top of the loop with them pending -- there the wait would be a full vmcnt(0), taken right after the
    # NEXT chunk's loads were issued (every chunk would pay an HBM round trip).
    pin = ", ".join(f'"+v"(x{n.i})' for n in plan.inputs)
    if pin:
        L.append(f"  asm volatile(\"\" : {pin});")
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
    in_loop[0] = True
    if inv_coefs or inv_mats or ulds:
        L.append("    int zo; asm volatile(\"s_mov_b32 %0, 0\" : \"=s\"(zo));   // opaque 0: keeps the table reads inside the iteration")
    if has_streams:
        L.append(f"    if (lane == 0) {{   // the states as they stand before this chunk, in case it has to be handed to the generic kernel")
        for k, (name, c) in enumerate(cname.items()):
            L.append(f"      zt_snap[{k}] = {c};")
        L.append("    }")
        L.append("    bool zt_bad = false, zt_badl = false;")
    L.append("    const bool fin = f0 + 64 >= frames;   // the launch's last chunk: its last frame leaves every written variable as the script would")
    avail = {n.i for n in plan.inputs}
    raw_issued: set = set()
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
            if (name in plan.cells or name.startswith("memw@")) and not final:
                continue                                   # (a cell needs its "stored to" flag beside it: both go out at the end)
            if name not in stored and ready(o) and (final or not (o.uniform or o.kind == "const")):
                stored.add(name)
                pending.append((name, o))
        if pending and (final or len(pending) >= 12):
            L.append("    if (fin && lane == last) {")
            for name, o in pending:
                if name.startswith("memw@"):
                    continue
                if name in plan.cells:                     # a cell is written back only if the launch stored to it at all
                    flag = plan.outs.get("memw@" + name[4:])
                    if flag is not None:
                        L.append(f"      if ({ref(flag)} != 0.0) {{ {dst(name)} = {ref(o)}; zt_hc = zt_hc > ca{plan.cells[name].i} + 1 ? zt_hc : ca{plan.cells[name].i} + 1; }}")
                    continue
                L.append(f"      {dst(name)} = {ref(o)};")
            L.append("    }")
            pending.clear()

    for gid, it in enumerate(plan.items):
        kind = it[0]
        if kind == "par":
            avail.add(it[1].i)
        elif kind == "site":
            pass
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
        if kind == "site":
            st_: StoreSite = it[1]
            j, an = st_.j, ref(st_.addr)
            L.append(f"    // delay-line write {j}: must advance by one cell per frame (at most one wrap inside the chunk)")
            L.append(f"    const double sp{j} = zt_shift1({an}, {an} - 1.0);")
            L.append(f"    const uint64_t sm{j} = __ballot(valid && lane > 0 && ({an} - sp{j} != 1.0));")
            L.append(f"    const int sk{j} = sm{j} ? (int)__ffsll((long long)sm{j}) - 1 : tn;")
            L.append(f"    const int64_t s0{j} = (int64_t)zt_readlane({an}, 0), s1{j} = sk{j} < tn ? (int64_t)zt_readlane({an}, sk{j}) : 0;")
            L.append(f"    zt_bad |= __popcll(sm{j}) > 1 || s0{j} + sk{j} > mcap || (sk{j} < tn && s1{j} + (tn - sk{j}) > mcap);")
            if cell_addrs:
                L.append(f"    zt_bad |= (s0{j} <= cmax && s0{j} + sk{j} > cmin) || (sk{j} < tn && s1{j} <= cmax && s1{j} + (tn - sk{j}) > cmin);")
            continue
        if kind == "par" and it[1].kind == "ld":
            n = it[1]
            L.append(f"    double n{n.i};   // delay-line read: memory as it was before this chunk, or the value an earlier frame of the chunk writes")
            L.append("    {")
            if n.i in raw_issued:
                L.append(f"      const int64_t B = B{n.i};")
                L.append(f"      double v = raw{n.i};")
            else:
                L.append(f"      const int64_t B = (int64_t){ref(n.args[0])};")
                L.append("      double v = B < mcap ? memp[B * mse] : 0.0;")
            L.append("      int best = -1;")
            for st_ in plan.stores:
                j = st_.j
                L.append(f"      {{ int tw = -1; const int64_t d0 = B - s0{j}, d1 = B - s1{j};")
                L.append(f"        if ((uint64_t)d0 < (uint64_t)sk{j}) tw = (int)d0;")
                L.append(f"        if ((uint64_t)d1 < (uint64_t)(tn - sk{j})) tw = sk{j} + (int)d1;")
                if ",".join(map(str, st_.region)) != n.name:
                    L.append("        zt_badl |= tw >= 0; }")
                else:
                    before = "true" if st_.seq < n.val else "false"
                    L.append(f"        const bool vis = valid && tw >= 0 && (tw < lane || (tw == lane && {before})) && tw >= best;")
                    L.append(f"        if (__ballot(vis)) {{ const double fw = zt_bperm({ref(st_.value)}, tw); v = vis ? fw : v; best = vis ? tw : best; }} }}")
            if cell_addrs:
                L.append("      zt_badl |= B >= cmin && B <= cmax;")
            L.append(f"      n{n.i} = v;")
            L.append("    }")
            retire()
            continue
        if kind == "par":
            n = it[1]
            L.append(f"    const double n{n.i} = {_expr(n.op, [ref(x) for x in n.args])};")
            for ld in plan.loads:            # the reads of this address go out now: their latency overlaps everything up to their use
                if ld.args[0] is n and ld.i not in raw_issued:
                    raw_issued.add(ld.i)
                    L.append(f"    const int64_t B{ld.i} = (int64_t)n{n.i};")
                    L.append(f"    const double raw{ld.i} = B{ld.i} < mcap ? memp[B{ld.i} * mse] : 0.0;")
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
    if has_streams:
        L.append("    if (zt_bad || __ballot(valid && zt_badl)) {")
        L.append("      // a condition of the delay-line handling does not hold in this chunk: put the states back as they were before it")
        L.append("      // and leave the rest of the launch to the generic kernel (za_launch_fast runs it right behind this one)")
        L.append("      if (lane == 0) {")
        for k, name in enumerate(cname):
            if name.startswith("memw@") or name == RNG_INDEX:
                continue
            if name in plan.cells:
                flag = "memw@" + name[4:]
                if flag in cname:
                    fk = list(cname).index(flag)
                    L.append(f"        if (zt_snap[{fk}] != 0.0) {{ {dst(name)} = zt_snap[{k}]; zt_high = zt_high > ca{plan.cells[name].i} + 1 ? zt_high : ca{plan.cells[name].i} + 1; }}")
                continue
            L.append(f"        {dst(name)} = zt_snap[{k}];")
        L.append("        b.mem_high[inst] = zt_high;")
        L.append("        b.resume[inst] = f0;")
        L.append("      }")
        L.append("      return;")
        L.append("    }")
        L.append("    // the chunk's writes land after all of its reads are resolved")
        for st_ in plan.stores:
            L.append(f"    if (valid) memp[(int64_t){ref(st_.addr)} * mse] = {ref(st_.value)};")
            L.append(f"    {{ const int64_t h0 = s0{st_.j} + sk{st_.j}, h1 = sk{st_.j} < tn ? s1{st_.j} + (tn - sk{st_.j}) : 0; zt_high = h0 > zt_high ? h0 : zt_high; zt_high = h1 > zt_high ? h1 : zt_high; }}")
    retire(final=True)
    if has_mem:
        L.append("    if (fin && lane == last) { b.mem_high[inst] = zt_high > zt_hc ? zt_high : zt_hc; b.resume[inst] = frames; }")
    if pin:
        L.append(f"    asm volatile(\"\" : {pin});   // the next chunk's audio has landed; its wait comes before this chunk's stores")
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
    if has_mem:
        # the generic code of the leaf, one lane per instance, from wherever the kernel above stopped (normally: nowhere)
        L.append("// instances the time-parallel kernel handed back (b.resume[i] < frames) finish the launch here, frame by frame, with the")
        L.append("// generic section code -- the exact serial semantics; every other lane leaves at once")
        L.append(f'extern "C" __global__ void __launch_bounds__(64) {kernel_macro[:-1]}_tail)(ZabBatch b, ZabAudio a) {{')
        L.append("  ZA_KERNEL_ENTRY();")
        L.append("  const int inst = blockIdx.x * 64 + threadIdx.x;")
        L.append("  if (inst >= b.n_inst) return;")
        L.append("  const int64_t from = b.resume[inst];")
        L.append("  if (from >= a.frames) return;")
        L.append("  ZaS s;")
        L.append("  za_state_load(s, b, inst);")
        L.append(f"  const float* in = a.in + (int64_t)inst * {plan.nch} * a.frame_stride;")
        L.append(f"  float* out = a.out + (int64_t)inst * {plan.nch} * a.frame_stride;")
        L.append("  for (int64_t t = from; t < a.frames; ++t) {")
        for ch in range(plan.nch):
            L.append(f"    s.spl[{ch}] = (double)in[{ch} * a.frame_stride + t];")
        L.append("    za_section_sample(s);")
        for ch in range(plan.nch):
            L.append(f"    out[{ch} * a.frame_stride + t] = (float)s.spl[{ch}];")
        L.append("  }")
        L.append("  za_state_store(s, b, inst);")
        L.append("  b.resume[inst] = a.frames;")
        L.append("}")
    L.append("static int32_t za_fast_applies(const ZabBatch* b, const ZabAudio* a) { (void)b; return a->frames > 0 ? 1 : 0; }")
    L.append("static hipError_t za_launch_fast(const ZabBatch* b, const ZabAudio* a, hipStream_t st) {")
    L.append(f"  hipLaunchKernelGGL({kernel_macro}, dim3(b->n_inst), dim3(64), 0, st, *b, *a);")
    if has_mem:
        L.append(f"  hipLaunchKernelGGL({kernel_macro[:-1]}_tail), dim3((b->n_inst + 63) / 64), dim3(64), 0, st, *b, *a);")
    L.append("  return hipGetLastError();")
    L.append("}")
    return "\n".join(L) + "\n"
