"""Time-parallel lowering of a leaf's @sample section: ONE WAVEFRONT PER INSTANCE, lane = frame.

The generic kernels (csrc/zab_generic.hip.h) run a script the way jsfx_process_block does (dsp_jsfx_aot.py:5785-5899): host
block after host block -- @block, the pending-mask @slider check -- and inside a block one frame after the other, one lane
per instance. Most of a dynamics / filter script is not serial in time at all, though. This module proves which parts are,
per leaf, and emits a second kernel `zab_<leaf>_tpar` that keeps the block structure (the wavefront runs @block and @slider
itself between the blocks, with the leaf's ordinary section code) and processes the frames of a block 64 at a time:

  1. @sample (user functions inlined, conditionals if-converted) becomes a DAG over the values of ONE frame: inputs
     spl0.., invariants (variables / sliders that @sample never writes: they change only in @block / @slider / @init, i.e.
     between blocks), constants, and `state-in` nodes -- the value a variable written by @sample had at the end of the
     PREVIOUS frame. A variable whose incoming value no path of @sample can observe (every read follows a write of the same
     frame) has no state-in at all: where a frame leaves it alone it carries a HOLD marker, and only its last written value
     is tracked.
  2. The cross-frame edges out(v)[t-1] -> state-in(v)[t] close cycles. Strongly connected components of that graph are
     the true recurrences; everything else is feed-forward in time and runs one lane per frame.
       * no cycle through state-in(v) ......... v is a delayed signal: a one-lane shift of out(v) (DPP wave_shr),
       * a cycle that is AFFINE in its states .. y[t] = A[t] y[t-1] + b[t] with A, b free of y (one-poles, leaky
         integrators, counters, sample-and-hold `c ? y = x`, biquads as 2x2): a weighted prefix scan over the wavefront
         with DPP row_shr / row_bcast moves (the scheme of the hand-written DDT kernel, csrc/kernels/ddt_fast.hip.h:98-108),
       * affine once its state-dependent conditions are fixed (attack/release smoothers, holds): fixed-point iteration of
         the condition pattern,
       * anything else: the minimal cycle runs as a uniform 64-step loop, inputs broadcast with v_readlane.
  3. loop() / while whose trip count is the same in every frame run as UNIFORM loops: trip k of all 64 frames together, then
     trip k + 1. Counters are wave-uniform, variables handed from trip to trip are per-lane values, and mem[] at addresses
     that depend on the counters only are PER-TRIP CELLS -- band k's filter state -- whose recurrence over the frames is
     classified and solved inside the trip exactly like a top-level one (the band loops of ERBTilt, SpectralStabilizer,
     EasyExpander, CMD). Reads at moving addresses inside such a loop are gathers (FIR taps into a ring: TSEQ, DOT).
  4. Values that depend on invariants only are computed once per host block.

The state a launch leaves in vars[] / spl[] / mem[] is what the serial path leaves. Affine components differ from the serial
order of operations by re-association only (O(1e-16) relative), checked by the same reference-VM fixtures as the generic
path (tests/test_tpar.py, tests/test_catalog_gpu.py). Everything the lowering assumes but cannot prove (distinct address
expressions address distinct cells, delay lines advance by one cell per frame, rare-event guards stay false) is checked at
run time; a launch that breaks an assumption is handed, from that point on, to the serial section code
(`zab_<leaf>_tpar_tail`).

`Plan.simulate` is a numpy restatement of the staged algorithm (one array element per lane) used by the CPU tests to pin
the analysis itself -- classification, coefficients, carries, partial chunks, loops -- without a GPU.
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
HOLD_BITS = 0x7FF800005A5A5A5A   # "this frame left the variable alone": a quiet NaN no arithmetic produces (payload in the low
HOLD = np.array([HOLD_BITS], dtype=np.uint64).view(np.float64)[0]   # word, which a float -> double conversion leaves zero)


class Unsupported(Exception):
    """@sample uses a construct the time-parallel lowering does not handle; the leaf keeps the generic kernel only."""


class N:
    __slots__ = ("i", "kind", "op", "args", "val", "name", "uniform", "su", "extra", "loop", "ctx", "fb")

    def __init__(self, i, kind, op=None, args=(), val=None, name=None):
        self.i, self.kind, self.op, self.args, self.val, self.name = i, kind, op, tuple(args), val, name
        self.uniform = False
        self.su = False          # wave-uniform, known while walking: built from constants, variables @sample never assigns and
        #                          uniform loop counters
        self.ctx = ()            # ld: the statements around it that could run as events instead (FrameGraph.ctx)
        self.fb = frozenset()    # ld in a feedback loop through its delay line: the writes (StoreSite.j) whose values depend on it --
        #                          never forwarded from, the chunk is cut short instead
        self.extra = ()          # ld: nodes this load must wait for besides its address (the stores it may have to forward from)
        self.loop = None         # innermost uniform loop this node's value changes in (None: once per frame)

    def __repr__(self):
        if self.kind == "const":
            return f"#{self.i}:{self.val!r}"
        if self.kind in ("var", "inv", "st", "in", "hold", "phi", "lout", "lcin"):
            return f"#{self.i}:{self.kind}({self.name})"
        if self.kind == "ld":
            return f"#{self.i}:ld({self.args[0].i})"
        return f"#{self.i}:{self.op}(" + ",".join(str(a.i) for a in self.args) + ")"


BIN_OPS = {"+", "-", "*", "/", "<", "<=", ">", ">=", "==", "!=", "^", "|", "&", "~", "<<", ">>", "%"}
CALL1 = set(PURE_MATH1) | {"sqr", "sign", "invsqrt"}
MT_N, MT_M = 624, 397
CALL2 = set(PURE_MATH2) | {"min", "max"}


class LoopInfo:
    """One uniform loop of the frame: every frame runs the same number of trips, so the wavefront runs trip k of all of its
    frames together."""

    def __init__(self, lid, parent):
        self.id = lid
        self.parent: Optional["LoopInfo"] = parent
        self.depth = 1 + (parent.depth if parent is not None else 0)
        self.count: Optional[N] = None           # loop(n): the count node (evaluated once, before the first trip)
        self.cond: Optional[N] = None            # while: the condition, in terms of this loop's phis
        self.order: List[str] = []               # loop-carried names, order of first appearance
        self.phis: Dict[str, N] = {}
        self.init: Dict[str, N] = {}
        self.next: Dict[str, N] = {}
        self.louts: Dict[str, N] = {}
        self.cells: Dict[str, N] = {}            # "lmem@<id>" -> address node (wave-uniform, changes from trip to trip)
        self.cin: Dict[str, N] = {}              # -> the cell's value before this frame
        self.cell_out: Dict[str, N] = {}         # -> the value this frame stores (cells the loop never stores to: absent)
        self.cell_flag: Dict[str, N] = {}        # -> "a store to this cell ran in this frame" (the arena's high-water mark)
        self.children: List["LoopInfo"] = []
        self.guards: List[N] = []                # conditions (wave-uniform per trip) of statements of the body that became events
        self.ctx: tuple = ()                     # the statements around the loop (itself included) that could run as events instead
        self.entry_pred: Optional[N] = None      # path condition the loop statement stands under (None: runs in every frame)

    def inside(self, other: Optional["LoopInfo"]) -> bool:
        """self is `other` or nested in it (other None: the frame itself)."""
        x = self
        while x is not None:
            if x is other:
                return True
            x = x.parent
        return other is None


def _deeper(a: Optional[LoopInfo], b: Optional[LoopInfo]) -> Optional[LoopInfo]:
    if a is None:
        return b
    if b is None:
        return a
    return a if a.depth >= b.depth else b


# ----------------------------------------------------------------------------------------------------------------------
# 0. passes over the syntax tree: rare-event guards, variables whose incoming value is observable
# ----------------------------------------------------------------------------------------------------------------------
def _reachable_fns(prog: Program, roots) -> List[str]:
    seen, todo = [], list(roots)
    while todo:
        x = todo.pop()
        if isinstance(x, S.Call) and x.fn in prog.fns and x.fn not in seen:
            seen.append(x.fn)
            todo.append(prog.fns[x.fn].body)
        todo.extend(S.children(x))
    return seen


def _event_head(x):
    """What still runs in every frame of a statement the lowering has turned into an event (FrameGraph.event_ids): its
    condition / count / left operand. The rest is the event's body, which only the serial section code ever runs."""
    if isinstance(x, (S.Cond, S.If, S.While)):
        return x.cond
    if isinstance(x, S.Loop):
        return x.count
    if isinstance(x, S.Binary):
        return x.l
    raise AssertionError(type(x))


def _assigned_names(prog: Program, roots, shadow=(), skip=frozenset()) -> set:
    """Variables assigned by `roots` or a function they can reach (parameters of those functions are their own). Statements
    whose id is in `skip` are events: only their heads count."""
    out = set()
    seen = set()

    def walk(x, sh):
        if id(x) in skip:
            walk(_event_head(x), sh)
            return
        if isinstance(x, S.Assign) and isinstance(x.target, S.Var) and x.target.name not in sh:
            out.add(x.target.name)
        if isinstance(x, S.Call) and x.fn in prog.fns and x.fn not in seen:
            seen.add(x.fn)
            walk(prog.fns[x.fn].body, frozenset(prog.fns[x.fn].params))
        for c in S.children(x):
            walk(c, sh)

    for r in roots:
        walk(r, frozenset(shadow))
    return out


def _outarg_names(prog: Program, roots) -> set:
    """Variables handed to a runtime builtin as plain arguments: the builtin may assign them (msg_recv's outputs, file_var,
    file_riff, midirecv, slider_next_chg ...)."""
    out, seen, todo = set(), set(), list(roots)
    while todo:
        x = todo.pop()
        if isinstance(x, S.Call):
            if x.fn in prog.fns:
                if x.fn not in seen:
                    seen.add(x.fn)
                    todo.append(prog.fns[x.fn].body)
            elif ("fabs" if x.fn == "abs" else x.fn) not in (CALL1 | CALL2) and x.fn != EVENT:
                out |= {a.name for a in x.args if isinstance(a, S.Var)}
        todo.extend(S.children(x))
    return out


def _read_names(prog: Program, roots) -> set:
    out, seen, todo = set(), set(), list(roots)
    while todo:
        x = todo.pop()
        if isinstance(x, S.Var):
            out.add(x.name)
        if isinstance(x, S.Call) and x.fn in prog.fns and x.fn not in seen:
            seen.add(x.fn)
            todo.append(prog.fns[x.fn].body)
        todo.extend(S.children(x))
    return out


def _pure_scalar(prog: Program, x) -> bool:
    """An expression over variables and constants only: no memory, no calls with effects, no assignment."""
    if isinstance(x, (S.Num, S.Var)):
        return True
    if isinstance(x, (S.Unary, S.Binary)):
        return all(_pure_scalar(prog, c) for c in S.children(x))
    if isinstance(x, S.Call):
        fn = "fabs" if x.fn == "abs" else x.fn
        return fn not in prog.fns and fn in (CALL1 | CALL2) and all(_pure_scalar(prog, a) for a in x.args)
    return False


def split_guards(prog: Program):
    """Top-level statements of @sample of the form `G ? ( ... )` where G reads only variables that nothing else in @sample
    writes and the body assigns one of them -- the "rebuild when the sample rate changed / a table is dirty" idiom: running the
    body clears the condition. The lowering takes G to be false (the statements are dropped, so everything they assign stays
    an invariant); the kernel evaluates every G at the start of each block and hands a launch whose G holds to the serial code.
    Returns (remaining statements, guard conditions)."""
    items = list(prog.sections.get("sample", []))
    if len(items) == 1 and isinstance(items[0], S.Seq):
        items = list(items[0].items)
    cand = {k for k, st in enumerate(items)
            if isinstance(st, (S.Cond, S.If)) and (st.els is None or isinstance(st.els, S.Num)) and st.then is not None
            and _pure_scalar(prog, st.cond)
            and (_assigned_names(prog, [st.then]) & _read_names(prog, [st.cond]))}
    while cand:
        rest = [st for k, st in enumerate(items) if k not in cand]
        wrest = _assigned_names(prog, rest)
        bad = {k for k in cand if any(is_spl_name(nm) is not None or nm in wrest for nm in _read_names(prog, [items[k].cond]))}
        if not bad:
            break
        cand -= bad
    return [st for k, st in enumerate(items) if k not in cand], [items[k].cond for k in sorted(cand)]


def _heavy(prog: Program, x, seen=None) -> bool:
    """Something the per-frame dataflow cannot take, or should not pay for in every frame: a loop, or a builtin with effects
    (memcpy, fft, convolve_c, file and string calls ...), here or in a function called from here."""
    seen = set() if seen is None else seen
    if isinstance(x, (S.Loop, S.While)):
        return True
    if isinstance(x, S.Call):
        fn = "fabs" if x.fn == "abs" else x.fn
        if fn in prog.fns:
            if fn not in seen:
                seen.add(fn)
                if _heavy(prog, prog.fns[fn].body, seen):
                    return True
        elif not (fn in (CALL1 | CALL2) or fn in NOOP_CALLS or fn.startswith("gfx_") or fn in ("rand", "__memtop")):
            return True
    return any(_heavy(prog, c, seen) for c in S.children(x))


EVENT = "__event"


def split_events(prog: Program, stmts, keep=frozenset(), origin=None, cache=None):
    """Statements `C ? ( ... )` (no else) of @sample -- at the top or inside other conditionals, not inside loops or functions --
    whose body is heavy (_heavy) and whose condition is a plain expression: the "every hop: run the FFT" / "buffer full: convolve
    a block" idiom. The lowering replaces each by a marker that keeps the condition; the kernel evaluates the conditions first in
    every chunk, lets the frames before the first one that holds take the parallel path, runs that one frame with the serial
    section code (zt_frame) and starts over behind it. Returns (rewritten statements, bodies dropped)."""
    dropped = []

    def rw(x, stmt: bool):
        # (what holds no event keeps its identity: FrameGraph.event_ids names statements by it, from one build of the plan to the next)
        if isinstance(x, S.Seq):
            n = len(x.items)
            items = [rw(it, stmt or k + 1 < n) for k, it in enumerate(x.items)]
            if all(a is b for a, b in zip(items, x.items)):
                return x
            return S.Seq(items, line=x.line, col=x.col)
        if isinstance(x, (S.Cond, S.If)):
            st = stmt or isinstance(x, S.If)
            if (st and x.then is not None and (x.els is None or isinstance(x.els, S.Num)) and _pure_scalar(prog, x.cond)
                    and _heavy(prog, x.then) and id(x) not in keep):
                dropped.append(x)
                mark = S.Call(EVENT, [x.cond], line=x.line, col=x.col)
                if origin is not None:
                    origin[id(mark)] = x
                return mark
            th = rw(x.then, st) if x.then is not None else None
            el = rw(x.els, st) if x.els is not None else None
            if th is x.then and el is x.els:
                return x
            return type(x)(x.cond, th, el, line=x.line, col=x.col)
        return x

    key = (tuple(id(st) for st in stmts), frozenset(keep))
    if cache is not None and cache.get("key") == key:
        if origin is not None:
            origin.update(cache["origin"])
        return cache["out"], cache["dropped"]
    out = [rw(st, True) for st in stmts]
    if cache is not None:
        cache.update(key=key, out=out, dropped=dropped, origin=dict(origin or {}))
    return out, dropped


def exposed_vars(prog: Program, stmts, skip=frozenset()) -> set:
    """Variables whose value from the previous frame some path of the frame can read (a read not preceded, on that path, by a
    write of the same frame). Mirrors FrameGraph's order of evaluation; conservative (a loop body may run zero times, the
    right operand of && / || may not run)."""
    exposed: set = set()
    depth = [0]

    def ev(x, d: set, sh: frozenset):
        if id(x) in skip:
            ev(_event_head(x), d, sh)
            return
        if isinstance(x, (S.Num, S.Str)):
            return
        if isinstance(x, S.Var):
            if x.name not in sh and x.name not in d:
                exposed.add(x.name)
            return
        if isinstance(x, S.Assign):
            ev(x.value, d, sh)
            t = x.target
            if isinstance(t, S.Var):
                if x.op != "=" and t.name not in sh and t.name not in d:
                    exposed.add(t.name)
                if t.name not in sh:
                    d.add(t.name)
            else:
                for c in S.children(t):
                    ev(c, d, sh)
            return
        if isinstance(x, S.Binary) and x.op in ("&&", "||"):
            ev(x.l, d, sh)
            ev(x.r, set(d), sh)
            return
        if isinstance(x, (S.Cond, S.If)):
            ev(x.cond, d, sh)
            a, b = set(d), set(d)
            if x.then is not None:
                ev(x.then, a, sh)
            if x.els is not None:
                ev(x.els, b, sh)
            d |= (a & b)
            return
        if isinstance(x, S.Loop):
            ev(x.count, d, sh)
            ev(x.body, set(d), sh)
            return
        if isinstance(x, S.While):
            ev(x.cond, d, sh)
            ev(x.body, set(d), sh)
            return
        if isinstance(x, S.Call):
            for a in x.args:
                ev(a, d, sh)
            if x.fn in prog.fns and depth[0] < 40:
                depth[0] += 1
                ev(prog.fns[x.fn].body, d, frozenset(prog.fns[x.fn].params))
                depth[0] -= 1
            return
        for c in S.children(x):
            ev(c, d, sh)

    d: set = set()
    for st in stmts:
        ev(st, d, frozenset())
    return exposed


# ----------------------------------------------------------------------------------------------------------------------
# 1. one frame of @sample as a DAG
# ----------------------------------------------------------------------------------------------------------------------
class FrameGraph:
    def __init__(self, prog: Program, nch: int, stmts=None, event_ids=None, no_event=None):
        self.p, self.nch = prog, nch
        # statements (by id of their syntax node, anywhere @sample reaches) that run as EVENTS: their condition is part of the
        # frame, their body is not -- the frame one falls on runs with the serial section code. split_events picks the obvious
        # ones up front; the walk adds every conditional / loop whose body it cannot lower (`_or_event`), and the plan is then
        # built again with the set known from the start (what @sample assigns, and so what is an invariant, depends on it).
        self.event_ids: set = set() if event_ids is None else event_ids
        self.no_event: set = set() if no_event is None else no_event      # ... whose condition proved unusable: never again
        self.new_events = 0
        self.reasons: Dict[int, str] = {}         # id of a statement -> why an earlier build of the plan made it an event
        self.event_why: Dict[int, str] = {}
        self.event_src: Dict[int, int] = {}      # event condition node -> id of the statement it came from
        self.ctx: List = []                      # statements being lowered that could run as events instead, outermost first
        self.nodes: List[N] = []
        self.memo: Dict[tuple, N] = {}
        self.env: Dict[str, N] = {}
        self.varnodes: Dict[str, N] = {}
        self.written: List[str] = []
        self.scope: List[Dict[str, str]] = []
        self.depth = 0
        self.rand_sites = 0
        self.stmts = list(prog.sections.get("sample", [])) if stmts is None else list(stmts)
        # variables assigned anywhere in @sample (or a function it can reach): everything else is constant over a block, which
        # lets the walk tell block-constant addresses (mem[] cells used as named state) from moving ones (delay lines)
        skip = frozenset(self.event_ids)
        # variables no section ever assigns (and no builtin can: out-arguments) are 0 for good. A script that keeps a buffer's base
        # in one (Alias: `instance(buf, pos)` with buf never set) has all such buffers at mem[0]: such names do not tell regions apart
        everything = [st for sec in prog.sections.values() for st in sec]
        self.never_assigned = set(prog.vars) - _assigned_names(prog, everything) - _outarg_names(prog, everything) - set(prog.aliases.values())
        self.wsyn = _assigned_names(prog, self.stmts, skip=skip)
        self.exposed = set(prog.vars) if os.environ.get("ZA_TPAR_NO_HOLD") else exposed_vars(prog, self.stmts, skip=skip)
        self.pred: Optional[N] = None            # path condition of the statement being walked (None: unconditional)
        self.mem_seq = 0                         # program order of the memory operations of a frame
        self.cells: Dict[str, N] = {}            # "mem@<id>" -> its (block-constant) address node
        self.loads: List[N] = []                 # moving-address loads
        self.stores: List["StoreSite"] = []      # moving-address stores
        self.loops: List[LoopInfo] = []          # every uniform loop of the frame, outer before inner
        self.loop_stack: List[LoopInfo] = []
        self.loop_ids = 0
        self.lcell_addr: Dict[str, N] = {}       # "lmem@<id>" -> address node (cells of uniform loops)
        self.holds: Dict[str, N] = {}
        self.events: List[N] = []                # path condition && condition of every event marker (split_events)
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
            for a in args:
                n.loop = _deeper(n.loop, a.loop)
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

    def _lcell_owner(self, key: str) -> LoopInfo:
        return self.lcell_addr["lmem@" + key.split("@", 1)[1]].loop

    def read(self, name: str) -> N:
        key = self._canon(name)
        if key in self.env:
            return self.env[key]
        if key.startswith("%"):
            raise Unsupported(f"parameter {name} read before it was bound")
        if key.startswith("lmemw@"):
            return self.ZERO                                   # no store to this cell yet in this trip
        if key.startswith("lmem@"):
            owner = self._lcell_owner(key)
            n = owner.cin.get(key)
            if n is None:
                n = self.mk("lcin", name=key, val=owner.id)
                n.loop = owner
                owner.cin[key] = n
                owner.cells[key] = self.lcell_addr[key]
            return n
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
            n = self.mk("var", name=key)                        # (constant over a block)
        else:
            if (is_slider_name(key) is None and key not in ("srate", "midi_bus", "ext_midi_bus", RNG_INDEX) and key not in self.p.vars
                    and key not in self.cells and not key.startswith("memw@")):
                raise Unsupported(f"unknown variable {key}")
            if key in self.wsyn and key not in self.exposed and key in self.p.vars:
                # no path of the frame can observe this variable's incoming value: where the frame leaves it alone it carries
                # the HOLD marker instead of a state-in node (which would make every conditional temporary a recurrence)
                n = self.holds.get(key)
                if n is None:
                    n = self.holds[key] = self.mk("hold", name=key)
                return n
            n = self.mk("var", name=key)
        self.varnodes[key] = n
        return n

    def write(self, name: str, node: N):
        key = self._canon(name)
        if key.startswith("lmem@") or key.startswith("lmemw@"):
            self.env[key] = node
            return
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
        """Block-constant terms of base + index: accesses that differ in them are taken to address different buffers (checked
        at run time, chunk by chunk: a load that falls into another buffer's freshly written span aborts the fast path)."""
        terms, todo = [], list(a.args)
        while todo:
            x = todo.pop()
            if x.kind == "op" and x.op == "+":
                todo.extend(x.args)
            elif (x.su and x.loop is None and not (x.kind == "const" and x.val == 0.0)
                  and not (x.kind in ("var", "inv") and x.name in self.never_assigned)):
                terms.append(x.i)
        return tuple(sorted(terms))

    def _cell(self, a: N) -> str:
        if a.loop is not None:                     # changes from trip to trip of a uniform loop: a cell per trip
            key = f"lmem@{a.i}"
            self.lcell_addr[key] = a
            return key
        key = f"mem@{a.i}"
        self.cells[key] = a
        return key

    def _load(self, a: N) -> N:
        if a.su:                                   # a cell: mem[] used as a named state variable
            return self.read(self._cell(a))
        self.mem_seq += 1
        ld = self.mk("ld", args=(a,), val=self.mem_seq)
        ld.ctx = tuple(self.ctx)
        self.loads.append(ld)
        return ld

    def _store(self, a: N, v: N):
        if a.su:
            key = self._cell(a)
            self.write(key, v)
            # "has this cell been stored to in this launch": the write high-water mark of the arena moves only for executed
            # stores, and a store under a condition may never run. (An ordinary state: its updates merge like any variable's.)
            self.write(("lmemw@" if key.startswith("lmem@") else "memw@") + key.split("@", 1)[1], self.ONE)
            return
        if self.loop_stack:
            raise Unsupported("store to a moving mem[] address inside a loop")
        self.mem_seq += 1
        self.stores.append(StoreSite(len(self.stores), a, v, self.pred, self.mem_seq, self._region(a), tuple(self.ctx)))

    def v_Index(self, n):
        return self._load(self._address(n))

    # -- uniform loops ----------------------------------------------------------------------------------------------------------
    def _snapshot(self):
        return (dict(self.env), list(self.written), list(self.loads), list(self.stores), dict(self.cells), self.mem_seq,
                self.rand_sites, dict(self.varnodes), list(self.loops), dict(self.lcell_addr), dict(self.holds))

    def _restore(self, s):
        (self.env, self.written, self.loads, self.stores, self.cells, self.mem_seq, self.rand_sites, self.varnodes, self.loops,
         self.lcell_addr, self.holds) = (dict(s[0]), list(s[1]), list(s[2]), list(s[3]), dict(s[4]), s[5], s[6], dict(s[7]), list(s[8]),
                                        dict(s[9]), dict(s[10]))

    def _loop(self, body_ast, count_ast, cond_ast) -> N:
        """loop(count, body) / while (cond) body as a UNIFORM loop: the trip count must come out the same in every frame of a
        block (count / cond built from invariants and uniform counters only); variables the body assigns are handed from trip
        to trip (phi nodes), wave-uniform where their first value and their update are."""
        count = self.ev(count_ast) if count_ast is not None else None
        if count is not None and not count.su:
            raise Unsupported("loop() count differs from frame to frame")
        roots = [x for x in (body_ast, cond_ast) if x is not None]
        # (parameters of the enclosing function are locals of this call: their canonical names)
        carried = sorted({self._canon(nm) for nm in _assigned_names(self.p, roots, skip=frozenset(self.event_ids))})
        uniform = set(carried)
        parent = self.loop_stack[-1] if self.loop_stack else None
        for _attempt in range(64):
            snap = self._snapshot()
            self.loop_ids += 1
            L = LoopInfo(self.loop_ids, parent)
            L.ctx = tuple(self.ctx)
            L.entry_pred = self.pred
            L.count = count
            env0 = self.env
            self.env = dict(env0)
            for v in carried:
                init = env0[v] if v in env0 else (self.ZERO if v.startswith("%") else self.read(v))
                phi = self.mk("phi", name=v, val=L.id)
                phi.su = v in uniform and init.su
                phi.loop = L
                L.order.append(v)
                L.phis[v], L.init[v] = phi, init
                self.env[v] = phi
            self.loop_stack.append(L)
            try:
                if cond_ast is not None:
                    before = dict(self.env)
                    L.cond = self.ev(cond_ast)
                    if len(self.env) != len(before) or any(self.env.get(k) is not v for k, v in before.items()):
                        raise Unsupported("while condition with side effects")
                    if not L.cond.su:
                        raise Unsupported("while condition differs from frame to frame")
                self.ev(body_ast)
            finally:
                self.loop_stack.pop()
            # cells of this loop are per trip: they leave the environment here
            for key in [k for k in self.env if (k.startswith("lmem@") or k.startswith("lmemw@")) and self._lcell_owner(k) is L]:
                node = self.env.pop(key)
                # (conditions INSIDE the body reached these values through the branches' merges; the path condition the loop
                #  itself stands under -- `c ? ( loop(...) )` -- did not: the cells leave the environment here, before that
                #  conditional merges what its arms assigned. A frame whose condition is false leaves every cell as it was.)
                if key.startswith("lmemw@"):
                    L.cell_flag["lmem@" + key[6:]] = node if self.pred is None else self.op("land", self.pred, node)
                else:
                    if self.pred is not None:
                        node = self.sel(self.pred, node, self.read(key))
                    L.cell_out[key] = node
                    L.cells[key] = self.lcell_addr[key]
            # names the body wrote that were not known as carried (cells of the frame or of an outer loop, the generator's
            # position): walk again with them
            extra = [k for k, v in self.env.items() if k not in carried and env0.get(k) is not v]
            lost = [v for v in uniform if v in L.phis and not (L.init[v].su and self.env[v].su)]
            if extra or lost:
                carried = carried + extra
                uniform = (uniform | set(extra)) - set(lost)
                self._restore(snap)
                continue
            for gc in L.guards:
                if not self._trip_uniform(gc, L, {}):
                    raise Unsupported("event inside a loop with a condition that differs from frame to frame")
            for v in carried:
                L.next[v] = self.env[v]
            env1 = dict(env0)
            for v in carried:
                if L.next[v] is L.phis[v]:
                    continue                           # (assigned on no path that was walked)
                lo = self.mk("lout", name=v, val=L.id)      # (never `su`: it exists only once the loop has run, and what is built
                lo.loop = parent                            #  from su values is computed in a block's prologue)
                L.louts[v] = lo
                env1[v] = lo
                if not (v.startswith("%") or v.startswith("lmem")) and v not in self.written:
                    self.written.append(v)
            self.env = env1
            self.loops.append(L)
            if parent is not None:
                parent.children.append(L)
            return self.ZERO
        raise Unsupported("loop analysis did not settle")

    def v_Loop(self, n):
        def lower():
            r0 = self.rand_sites
            v = self._loop(n.body, n.count, None)
            if self.rand_sites != r0:
                raise Unsupported("rand() inside a loop")
            return v

        # as an event: the frames in which the loop runs at all (za_loopcount(count) >= 1)
        return self._or_event(n, lower, lambda: self.op(">=", self.ev(n.count), self.ONE), self.ZERO)

    def v_While(self, n):
        def lower():
            r0 = self.rand_sites
            v = self._loop(n.body, None, n.cond)
            if self.rand_sites != r0:
                raise Unsupported("rand() inside a loop")
            return v

        return self._or_event(n, lower, lambda: self.ev(n.cond), self.ZERO)

    # -- statements the lowering cannot take become events ------------------------------------------------------------------------
    def _checkpoint(self):
        return (self._snapshot(), list(self.events), dict(self.event_src), list(self.scope), self.depth, self.pred,
                list(self.loop_stack), [list(L.guards) for L in self.loop_stack])

    def _rollback(self, cp):
        self._restore(cp[0])
        self.events, self.event_src = list(cp[1]), dict(cp[2])
        self.scope, self.depth, self.pred = list(cp[3]), cp[4], cp[5]
        self.loop_stack = list(cp[6])
        for L, gs in zip(self.loop_stack, cp[7]):
            L.guards = list(gs)

    def _touches_memory(self, n: N, memo: Dict[int, bool]) -> bool:
        if n.i in memo:
            return memo[n.i]
        memo[n.i] = False
        r = n.kind in ("ld", "lout", "lcin", "phi") or n.loop is not None or any(self._touches_memory(a, memo) for a in n.args)
        memo[n.i] = r
        return r

    def _trip_uniform(self, n: N, L: LoopInfo, memo: Dict[int, bool]) -> bool:
        """n is the same for every frame of a block in a given trip of L (whose walk is complete: its stored cells are known)."""
        if n.i in memo:
            return memo[n.i]
        memo[n.i] = False
        if n.kind == "const":
            r = True
        elif n.kind in ("var", "inv", "phi"):
            r = n.su
        elif n.kind == "lcin":
            r = n.name not in L.cell_out and self._trip_uniform(self.lcell_addr[n.name], L, memo)
        elif n.kind == "op":
            r = n.op != "mtout" and all(self._trip_uniform(a, L, memo) for a in n.args)
        else:
            r = False
        memo[n.i] = r
        return r

    def _event_at(self, ast, c: N):
        """The statement `ast` runs as an event under condition c (and the path's): at the top of the frame it joins the
        conditions every chunk evaluates first; inside a uniform loop it must be wave-uniform per trip and is tested, trip by
        trip, before a segment starts (LoopInfo.guards)."""
        full = c if self.pred is None else self.op("land", self.pred, c)
        if full.kind == "const" and full.val == 0.0:
            return
        if _fold(c) not in (None, 0.0) and (self.pred is None or _fold(self.pred) not in (None, 0.0)):
            raise Unsupported("a statement that runs in every frame is no event")
        if self.loop_stack:
            if len(self.loop_stack) > 1:
                raise Unsupported("event in a nested loop")
            self.loop_stack[-1].guards.append(full)
            return
        if self._touches_memory(full, {}):
            raise Unsupported("event condition reads memory or a loop's result")
        self.events.append(full)
        self.event_src[full.i] = id(ast)

    def _or_event(self, ast, lower, cond, idle: N) -> N:
        """lower() -- or, where the lowering refuses what `ast` holds (Unsupported), the statement as an event: cond() is
        evaluated in every frame, the body only by the serial code in the frames whose condition holds; `idle` is the
        statement's value in all the others."""
        if id(ast) in self.event_ids:
            try:
                self._event_at(ast, cond())
            except Unsupported as ex:
                w = self.reasons.get(id(ast))        # (found after an earlier walk: that is the reason to report)
                raise Unsupported(w.split(": ", 1)[-1]) if isinstance(w, str) else ex
            return idle
        if id(ast) in self.no_event or os.environ.get("ZA_TPAR_NO_DYN_EVENTS"):
            return lower()
        cp = self._checkpoint()
        self.ctx.append(ast)
        try:
            return lower()
        except Unsupported as ex:
            self._rollback(cp)
            try:
                self._event_at(ast, cond())          # (no place / condition for an event: the next statement out gets its
            except Unsupported:                      #  chance, with the reason the body gave)
                raise ex
            self.event_ids.add(id(ast))
            self.event_why[id(ast)] = f"line {getattr(ast, 'line', '?')}: {ex}"
            self.new_events += 1
            return idle
        finally:
            self.ctx.pop()

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
            # (`a && ( heavy )` / `a || ( heavy )`: the right operand as an event under a / !a; the idle value is the left's verdict)
            gate = (lambda: self.ev(n.l)) if n.op == "&&" else (lambda: self.op("not", self.ev(n.l)))
            return self._or_event(n, lambda: self._short_circuit(n), gate, self.ZERO if n.op == "&&" else self.ONE)
        if n.op not in BIN_OPS:
            raise Unsupported(f"binary {n.op}")
        l = self.ev(n.l)
        r = self.ev(n.r)
        return self.op(n.op, l, r)

    def _short_circuit(self, n):
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

    def _cond_stmt(self, n) -> N:
        if n.then is None or not (n.els is None or isinstance(n.els, S.Num)):
            return self._branch(n.cond, n.then, n.els)[1]
        idle = self.ZERO if n.els is None else self.const(n.els.value)
        return self._or_event(n, lambda: self._branch(n.cond, n.then, n.els)[1], lambda: self.ev(n.cond), idle)

    def v_Cond(self, n):
        return self._cond_stmt(n)

    def v_If(self, n):
        self._cond_stmt(n)
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
        if fn == EVENT:
            orig = getattr(self, "event_origin", {}).get(id(n))
            try:
                self._event_at(n if orig is None else orig, self.ev(n.args[0]))
            except Unsupported:
                if orig is None:
                    raise
                self.no_event.add(id(orig))          # no event after all: the statement itself is walked next time
                raise _Replan()
            return self.ZERO
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

    def __init__(self, j, addr, value, pred, seq, region, ctx=()):
        self.j, self.addr, self.value, self.pred, self.seq, self.region = j, addr, value, pred, seq, region
        self.ctx = ctx                            # the statements around it that could run as events instead
        # "late":   the chunk's writes land after all of its reads; a read takes the value an earlier frame of the chunk
        #           writes from that frame's lane (store-to-load forwarding),
        # "early":  written before the reads (which then come from memory): buffers that loops gather from,
        # "sparse": under a per-frame condition, into a buffer @sample never reads (decimated histories for the UI).
        self.mode = "late"


class RingGroup:
    """Gathers of one uniform loop that read ONE ring relative to a per-lane position: address = S + ((P +- U) & mask) with S and
    mask block-constant, P the same for every trip (one ring position per frame, consecutive frames one cell apart) and U
    wave-uniform per trip (a tap's lag). All the cells such a loop reads lie in the window [P(frame 0) + min offset, P(frame 63)
    + max offset] of the ring, which the kernel stages in LDS once per chunk: a tap then is one conflict-free LDS read instead of
    a 512-byte gather that misses L2 (TSEQ: 3466 taps per frame; DOT; the DDT-class fixture)."""

    def __init__(self, idx, loop, S, P, mask):
        self.idx, self.loop, self.S, self.P, self.mask = idx, loop, S, P, mask
        self.loads: List[tuple] = []             # (ld node, U node, sign)
        self.site: Optional[StoreSite] = None    # this chunk's (early) write into the same ring, if any
        self.region = ""


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
        self.modk: Optional[N] = None            # "modc": y' = (y + modk) % modn
        self.modn: Optional[N] = None
        self.modmask = False                     # ... only with the start inside [0, modn): written as a mask, or a step under a condition
        self.modpow2 = False                     # ... written as (y + modk) & (modn - 1): modn must be a power of two
        self.reg: "Region" = None


class Region:
    """The frame itself (loop None) or the body of one uniform loop: its nodes, its recurrences, its schedule."""

    def __init__(self, loop: Optional[LoopInfo]):
        self.loop = loop
        self.nodes: List[N] = []
        self.outs: Dict[str, N] = {}             # state name -> node holding its value at the end of a frame
        self.st: Dict[str, N] = {}               # state name -> its state-in node
        self.items: List[tuple] = []
        self.comps: List[Component] = []
        self.subs: List["Region"] = []
        self.ext: List[N] = []                   # loop: everything outside that the body (nested loops included) reads


class Plan:
    def __init__(self):
        self.events: List[N] = []
        self.node_guard: Dict[int, tuple] = {}
        self.full_exposed: set = set()
        self.event_exposed: List[str] = []
        self.g: FrameGraph = None
        self.nch = 0
        self.outs: Dict[str, N] = {}             # variable (or splK) -> node holding its value at the end of a frame
        self.spl_out: List[N] = []               # per processed channel
        self.st: Dict[str, N] = {}               # state variable -> its state-in node
        self.items: List[tuple] = []             # schedule of one chunk
        self.uniform: List[N] = []               # per-block nodes, topological order
        self.invariants: List[N] = []
        self.inputs: List[N] = []
        self.stats: Dict[str, int] = {}
        self.uses_rand = False
        self.cells: Dict[str, N] = {}            # "mem@<id>" -> block-constant address node (mem[] used as named state)
        self.stores: List[StoreSite] = []        # delay-line writes (moving addresses), program order
        self.loads: List[N] = []                 # delay-line reads
        self.top: Region = None
        self.regions: Dict[int, Region] = {}     # loop id -> region
        self.guards: List[N] = []                # rare-event conditions taken to be false (split_guards)
        self.rings: Dict[int, List[RingGroup]] = {}   # loop id -> its ring windows
        self.holdvars: List[str] = []            # written variables that may carry the HOLD marker at the end of a frame
        self.has_block = False                   # the kernel runs @block (and the pending-mask @slider) between the blocks
        self.has_pending = False                 # the script can raise slider masks: pending ones run @slider before a launch

    # (numpy restatement: section 4 below)


class TparAbort(Exception):
    """A chunk broke a run-time condition of the lowering at frame `f0`; the kernel stops there and the serial code finishes
    the launch."""

    def __init__(self, f0, why):
        super().__init__(f"frame {f0}: {why}")
        self.f0, self.why = f0, why


MAX_LIVE_NODES = 2400  # a frame larger than this is not lowered (kernel size: see _build_plan)
ULDS_THRESHOLD = 64   # block-constant values beyond which they live in LDS rather than in (spilled) scalar registers
SPEC_TOL = 1.0e-13    # relative change of a state between two iterations below which it counts as settled (ZT_SPEC_TOL)
SPEC_MAX = 8          # iterations of a switched recurrence before the chunk falls back to its serial loop (ZT_SPEC_MAX)


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


def _in_subtree(n: N, loop: Optional[LoopInfo]) -> bool:
    """n's value changes inside `loop` (or a loop nested in it); loop None: every node."""
    if loop is None:
        return True
    return n.loop is not None and n.loop.inside(loop)


class _BadCone(Unsupported):
    """An event's condition needs, directly or through the recurrences it reads, memory or a loop's result: it cannot be
    evaluated ahead of the chunk."""

    def __init__(self, event):
        super().__init__("event condition reads memory or a loop's result")
        self.event = event


class _Blame(Unsupported):
    """Something found after the walk (a recurrence through a loop, a conditional store into a delay line ...) that the statements
    around its source -- `ctx`, outermost first -- could take out of the frame by running as events."""

    def __init__(self, why, ctx):
        super().__init__(why)
        self.blame_ctx = tuple(ctx)


class _Replan(Exception):
    """The set of statements that run as events changed: build the plan again with it."""


def build_plan(prog: Program, nch: int) -> Plan:
    """Raises Unsupported when the leaf cannot take the time-parallel kernel."""
    if not prog.has("sample") or nch <= 0:
        raise Unsupported("no audio @sample")
    if os.environ.get("ZA_TPAR_NO_BLOCK") and prog.has("block"):
        raise Unsupported("@block present")
    event_ids: set = set()
    no_event: set = set()
    why: Dict[int, str] = {}
    for _round in range(24):
        try:
            plan = _build_plan(prog, nch, event_ids, no_event, why)
        except _Replan:
            continue
        except _Blame as bl:
            ast = next((x for x in reversed(bl.blame_ctx) if id(x) not in event_ids and id(x) not in no_event), None)
            if ast is None or os.environ.get("ZA_TPAR_NO_DYN_EVENTS"):
                raise Unsupported(str(bl))
            event_ids.add(id(ast))
            why[id(ast)] = f"line {getattr(ast, 'line', '?')}: {bl}"
            continue
        plan.dyn_events = sorted(why[i] for i in event_ids if i in why and isinstance(why[i], str))
        plan.stats["dyn_events"] = len(plan.dyn_events)
        return plan
    raise Unsupported("the set of event statements did not settle")


def _build_plan(prog: Program, nch: int, event_ids: set, no_event: set, why: Dict[int, str]) -> Plan:
    stmts, guard_asts = (list(prog.sections["sample"]), []) if os.environ.get("ZA_TPAR_NO_GUARDS") else split_guards(prog)
    ev_bodies = []
    origin: Dict[int, object] = {}
    if not os.environ.get("ZA_TPAR_NO_EVENTS"):
        stmts, ev_bodies = split_events(prog, stmts, keep=frozenset(no_event), origin=origin, cache=why.setdefault("#split", {}))
    g = FrameGraph(prog, nch, stmts, event_ids, no_event)
    g.event_origin = origin
    g.reasons = why
    guards = [g.ev(c) for c in guard_asts]
    if any(not x.su for x in guards) or g.env:
        raise Unsupported("guard condition is not an invariant")
    for st in stmts:
        g.ev(st)
    if g.scope or g.loop_stack:
        raise AssertionError("scope leak")
    if g.new_events:
        # what @sample assigns -- and with it which variables are invariants, states, HOLD carriers -- was judged with these
        # statements' bodies still counted in: once more, with the set known from the start
        why.update(g.event_why)
        raise _Replan()
    if g.rand_sites * WAVE > MT_N:
        raise Unsupported("more rand() calls per chunk than one generation of the generator holds")
    plan = Plan()
    plan.g, plan.nch = g, nch
    plan.has_block = prog.has("block")
    plan.has_pending = prog.uses("sliderchange", "slider_automate")
    plan.guards = guards
    plan.events = [e for e in g.events if not (e.kind == "const" and e.val == 0.0)]
    plan.full_exposed = exposed_vars(prog, list(prog.sections["sample"])) if plan.events else set()
    written = list(g.written)
    # variables @sample leaves as they were (x = x) are not state
    for name in list(written):
        vn = g.varnodes.get(name) or g.holds.get(name)
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

    hold_memo: Dict[int, bool] = {}

    def may_hold(n: N) -> bool:
        if n.i in hold_memo:
            return hold_memo[n.i]
        hold_memo[n.i] = False
        r = False
        if n.kind == "hold":
            r = True
        elif n.kind == "op" and n.op == "sel":
            r = may_hold(n.args[1]) or may_hold(n.args[2])
        elif n.kind in ("lout", "phi"):
            L = loop_by_id[n.val]
            r = may_hold(L.init[n.name]) or may_hold(L.next[n.name])
        hold_memo[n.i] = r
        return r

    loop_by_id = {L.id: L for L in g.loops}
    g.loop_of = loop_by_id
    plan.holdvars = [name for name in written if may_hold(plan.outs[name])]
    # variables an event's body reads from the frame before although the rest of the frame writes them first: they are no
    # states of the lowering, yet the section code that runs the event's frame wants them in memory
    # (the audio channels are set from the input before the section code runs)
    plan.event_exposed = sorted(nm for nm in plan.full_exposed if nm in plan.outs and nm not in plan.st and nm not in plan.holdvars
                                and not (is_spl_name(nm) is not None and is_spl_name(nm) < nch))
    for name in plan.holdvars:
        if name in plan.st or name.startswith("mem") or name == RNG_INDEX:
            raise AssertionError(f"{name}: HOLD marker on a state")

    # ---- delay-line writes: how each lands -------------------------------------------------------------------------------------
    def reaches_load(n: N, memo: Dict[int, bool]) -> bool:
        if n.i in memo:
            return memo[n.i]
        memo[n.i] = False
        r = n.kind in ("ld", "lout", "lcin") or any(reaches_load(a, memo) for a in n.args)
        memo[n.i] = r
        return r

    for ld in plan.loads:
        ld.name = ",".join(map(str, g._region(ld.args[0])))
    rl_memo: Dict[int, bool] = {}
    for st_ in plan.stores:
        reg = ",".join(map(str, st_.region))
        same = [ld for ld in plan.loads if ld.name == reg]
        # (several writes into one delay line per frame -- Alias's six lines at mem[0] -- stay "late": a read takes the last write
        #  in front of it in (frame, program) order; the chunk checks that such writes move in step, emit_site_pairs)
        shared = sum(1 for o in plan.stores if o.region == st_.region) > 1
        if st_.pred is not None and not st_.pred.su:
            if same or shared:
                raise _Blame("conditional store to a delay line that @sample reads", st_.ctx)
            st_.mode = "sparse"
        elif (not shared and any(ld.loop is not None for ld in same) and all(ld.val > st_.seq for ld in same)
              and not reaches_load(st_.addr, rl_memo) and not reaches_load(st_.value, rl_memo)
              and not os.environ.get("ZA_TPAR_NO_EARLY")):
            st_.mode = "early"
    for ld in plan.loads:
        # a load may have to take its value from a store of this chunk: it waits for every store of its own buffer (address
        # and value) and, for the aliasing check, for the addresses of all the others
        ex = []
        for st_ in plan.stores:
            ex.append(st_.addr)
            if st_.pred is not None:
                ex.append(st_.pred)
            if st_.mode == "early" or ",".join(map(str, st_.region)) == ld.name:
                ex.append(st_.value)
        ld.extra = tuple(ex)

    # ---- live nodes -------------------------------------------------------------------------------------------------------------
    live: Dict[int, N] = {}
    live_loops: Dict[int, LoopInfo] = {}
    todo = list(plan.outs.values()) + list(plan.spl_out) + list(guards) + list(plan.events)
    todo += [x for st_ in plan.stores for x in (st_.addr, st_.value) + ((st_.pred,) if st_.pred is not None else ())]
    todo += [a for a in plan.cells.values()]

    def loop_live(L: LoopInfo):
        while L is not None and L.id not in live_loops:
            live_loops[L.id] = L
            if L.count is not None:
                todo.append(L.count)
            if L.cond is not None:
                todo.append(L.cond)
            if L.entry_pred is not None and L.parent is None:
                todo.append(L.entry_pred)
            for key, o in L.cell_out.items():
                todo.extend((o, L.cells[key]))
                if key in L.cell_flag:
                    todo.append(L.cell_flag[key])
            L = L.parent

    def guard_cone(L: LoopInfo):
        """What a loop's guards need from OUTSIDE the loop is computed per block like any invariant; their nodes inside the loop
        are evaluated by the loop's address pass only (emit_address_pass), not in the trips themselves."""
        seen, work = set(), list(L.guards)
        while work:
            x = work.pop()
            if x.i in seen or x.kind == "const":
                continue
            seen.add(x.i)
            if not _in_subtree(x, L):
                todo.append(x)
                continue
            if x.kind == "phi":
                work.extend((L.init[x.name], L.next[x.name]))
            elif x.kind == "lcin":
                work.append(g.lcell_addr[x.name])
            work.extend(x.args)

    for L in g.loops:
        if L.cell_out or L.guards:
            loop_live(L)
        if L.guards:
            guard_cone(L)
    while todo:
        n = todo.pop()
        if n.i in live:
            continue
        live[n.i] = n
        todo.extend(n.args)
        todo.extend(n.extra)
        if n.loop is not None:
            loop_live(n.loop)
        if n.kind == "st":
            todo.append(plan.outs[n.name])
        elif n.kind in ("phi", "lout"):
            L = loop_by_id[n.val]
            loop_live(L)
            todo.extend((L.init[n.name], L.next[n.name]))
        elif n.kind == "lcin":
            L = loop_by_id[n.val]
            loop_live(L)
            todo.append(L.cells[n.name])
            if n.name in L.cell_out:
                todo.append(L.cell_out[n.name])
    if len(live) > MAX_LIVE_NODES and not os.environ.get("ZA_TPAR_ANY_SIZE"):
        # one kernel holds the whole frame: 3DPanner's 1 800 nodes are ~35 000 instructions (two minutes of device compiler,
        # branches past the 128 KB a short branch reaches), Sample's 8 750 would be several times that
        raise Unsupported(f"the frame is too large for one kernel ({len(live)} nodes)")
    loops = [L for L in g.loops if L.id in live_loops]
    loops.sort(key=lambda L: (L.depth, L.id))
    for L in loops:
        # (a per-trip cell takes its identity from its address EXPRESSION; that two expressions never name one cell is checked
        #  at run time, zt_sites_ok, for addresses that step evenly through the trips of ONE loop)
        if L.depth > 1 and L.cells:
            raise Unsupported("per-trip cells in a nested loop")
    plan.loops = loops
    plan.loop_by_id = loop_by_id

    # ---- regions ----------------------------------------------------------------------------------------------------------------
    top = Region(None)
    top.outs, top.st = plan.outs, plan.st
    regions: Dict[int, Region] = {}
    for L in loops:
        r = Region(L)
        r.outs = dict(L.cell_out)
        r.st = {key: L.cin[key] for key in L.cell_out if key in L.cin and L.cin[key].i in live}
        regions[L.id] = r
        (regions[L.parent.id] if L.parent is not None else top).subs.append(r)
    plan.top, plan.regions = top, regions

    def region_of(n: N) -> Region:
        return top if n.loop is None else regions[n.loop.id]

    for i in sorted(live):
        region_of(live[i]).nodes.append(live[i])

    def sched_deps(n: N) -> tuple:
        if n.kind == "lcin":
            return (loop_by_id[n.val].cells[n.name],)
        if n.kind in ("op", "ld"):
            return n.args + n.extra
        return ()

    def loop_ext(r: Region):
        L = r.loop
        ext: Dict[int, N] = {}

        def want(x: N):
            if not _in_subtree(x, L) and x.kind != "const":
                ext[x.i] = x

        def walk(rr: Region):
            LL = rr.loop
            for x in (LL.count, LL.cond, LL.entry_pred if LL.parent is None else None):
                if x is not None:
                    want(x)
            for v in LL.order:
                if LL.phis[v].i in live or (v in LL.louts and LL.louts[v].i in live):
                    want(LL.init[v])
                    want(LL.next[v])
            for key, o in LL.cell_out.items():
                want(o)
                want(LL.cells[key])
                if key in LL.cell_flag:
                    want(LL.cell_flag[key])
            for n in rr.nodes:
                for a in sched_deps(n):
                    want(a)
            for s in rr.subs:
                walk(s)

        walk(r)
        r.ext = [ext[i] for i in sorted(ext)]

    for r in regions.values():
        loop_ext(r)

    # ---- recurrences of every region ---------------------------------------------------------------------------------------------
    comp_of: Dict[int, Component] = {}
    all_regions = [top] + [regions[L.id] for L in loops]
    plan.fb_loads = []

    def region_comps(r: Region):
        nodes = list(r.nodes)
        pos = {n.i: k for k, n in enumerate(nodes)}
        pseudo = {s.loop.id: len(nodes) + k for k, s in enumerate(r.subs)}
        succ: List[List[int]] = [[] for _ in range(len(nodes) + len(r.subs))]
        for n in nodes:
            for a in sched_deps(n):
                if a.i in pos:
                    succ[pos[a.i]].append(pos[n.i])
            if n.kind == "lout":
                succ[pseudo[n.val]].append(pos[n.i])
            if n.kind in ("st", "lcin") and n.name in r.outs and r.outs[n.name].i in pos and n.name in r.st:
                succ[pos[r.outs[n.name].i]].append(pos[n.i])
        for s in r.subs:
            for x in s.ext:
                if x.i in pos:
                    succ[pos[x.i]].append(pseudo[s.loop.id])
        out = []
        for comp in _sccs(len(succ), succ):
            cyclic = len(comp) > 1 or comp[0] in succ[comp[0]]
            if not cyclic:
                continue
            if any(k >= len(nodes) for k in comp):
                inner = [r.subs[k - len(nodes)].loop for k in comp if k >= len(nodes)]
                raise _Blame("a recurrence over the frames runs through a loop", inner[0].ctx)
            out.append(sorted((nodes[k] for k in comp), key=lambda n: n.i))
        return out

    for r in all_regions:
        members_of = region_comps(r)
        for _again in range(4):
            fb = [m for members in members_of for m in members if m.kind == "ld"]
            if not fb:
                break
            # FEEDBACK THROUGH A DELAY LINE: a stored value depends on a read of the same buffer (a feedback echo, a reverb loop).
            # While the read lands behind the chunk -- the delay is at least the chunk's length -- nothing of the chunk reaches
            # it and the loop closes over memory only; so such a read is never forwarded to, and the chunk is CUT before the
            # first frame that would read what one of its own frames writes (the scheduler's "cut", as for events; the frames
            # behind the cut start the next segment). Its edges from the buffer's stored values go away, and with them the cycle.
            for ld in fb:
                if ld.loop is not None or r.loop is not None or os.environ.get("ZA_TPAR_NO_FEEDBACK"):
                    raise _Blame("feedback through a delay line (a stored value depends on a load of the same buffer)", ld.ctx)
                # (only the writes whose VALUE lies on the cycle are cut off from the read; another write into the same line --
                #  `ring[wp] = x; ring[wp] += y` -- is forwarded as ever)
                cyc = {m.i for members in members_of if ld in members for m in members}
                fbs = {st_.j for st_ in plan.stores if ",".join(map(str, st_.region)) == ld.name and st_.value.i in cyc}
                if not fbs:
                    raise _Blame("feedback through a delay line (a stored value depends on a load of the same buffer)", ld.ctx)
                ld.fb = frozenset(fbs | set(ld.fb or ()))
                keep = []
                for st_ in plan.stores:
                    keep.append(st_.addr)
                    if st_.pred is not None:
                        keep.append(st_.pred)
                    if st_.mode == "early" or (",".join(map(str, st_.region)) == ld.name and st_.j not in ld.fb):
                        keep.append(st_.value)
                ld.extra = tuple(keep)
                if ld not in plan.fb_loads:
                    plan.fb_loads.append(ld)
            members_of = region_comps(r)
        for members in members_of:
            if any(m.kind == "ld" for m in members):
                raise _Blame("feedback through a delay line (a stored value depends on a load of the same buffer)",
                             next(m for m in members if m.kind == "ld").ctx)
            names = [m.name for m in members if m.kind in ("st", "lcin")]
            order = written if r.loop is None else list(r.loop.cell_out)
            names.sort(key=lambda nm: order.index(nm))
            c = Component(names, members)
            c.reg = r
            for m in members:
                comp_of[m.i] = c
            r.comps.append(c)
    for ld in plan.fb_loads:
        sites = [s_ for s_ in plan.stores if s_.j in ld.fb]
        if any(s_.mode != "late" for s_ in sites):
            raise _Blame("feedback through a delay line whose write is not an ordinary one", ld.ctx)

    # uniform nodes: per block (the frame's) or per trip (a loop's)
    def set_uniform(n: N):
        if n.kind in ("const", "inv", "hold"):
            n.uniform = True
        elif n.kind in ("st", "in", "ld", "lout", "guess"):
            n.uniform = False
        elif n.kind == "phi":
            n.uniform = n.su
        elif n.kind == "lcin":
            n.uniform = n.name not in loop_by_id[n.val].cell_out
        else:
            n.uniform = all(a.uniform for a in n.args) and n.i not in comp_of

    for i in sorted(live):
        set_uniform(live[i])
    # affine forms
    for ci, c in enumerate(x for r in all_regions for x in r.comps):
        _classify(g, c.reg, c, ci, live)
        if c.kind == "spec" and os.environ.get("ZA_TPAR_NO_SPEC"):
            c.kind = "serial"
    # nodes created by the affine analysis: liveness / uniformity of the new coefficient nodes. Placeholder-dependent nodes
    # and the synthetic compares live inside their unit only.
    comps_all = [c for r in all_regions for c in r.comps]
    inside = {x.i for c in comps_all if c.kind == "spec" for x in c.gdep + c.gnodes + c.slice}
    extra: Dict[int, N] = {}
    todo = [x for c in comps_all if c.kind in ("scan", "spec") for row in c.A for x in row]
    todo += [x for c in comps_all if c.kind in ("scan", "spec") for x in c.b]
    todo += [x for c in comps_all if c.kind == "modc" for x in (c.modk, c.modn)]
    todo += [a for c in comps_all if c.kind == "spec" for x in c.gdep + c.slice for a in x.args]
    while todo:
        n = todo.pop()
        if n.i in live or n.i in extra or n.i in inside:
            continue
        extra[n.i] = n
        todo.extend(n.args)
    for i in sorted(extra):
        n = extra[i]
        live[i] = n
        set_uniform(n)
        region_of(n).nodes.append(n)
    for r in all_regions:
        r.nodes.sort(key=lambda n: n.i)
    for r in regions.values():
        loop_ext(r)                           # (coefficient nodes may read further outside values)

    # ---- gathers that read a ring relative to the frame's position: staged through LDS (RingGroup) -------------------------------
    plan.rings = {}
    if not os.environ.get("ZA_TPAR_NO_RING"):
        for L in loops:
            groups: List[RingGroup] = []
            for ld in plan.loads:
                if ld.loop is not L or ld.i not in live:
                    continue
                f = _ring_form(ld, L)
                sites = [s_ for s_ in plan.stores if ",".join(map(str, s_.region)) == ld.name]
                if f is None or any(s_.mode != "early" or s_.pred is not None for s_ in sites):
                    continue
                S, P, U, sign, mask = f
                key = (tuple(x.i for x in S), P.i, mask.i)
                grp = next((q for q in groups if q.key == key), None)
                if grp is None:
                    grp = RingGroup(len(groups), L, S, P, mask)
                    grp.key, grp.region, grp.site = key, ld.name, (sites[0] if sites else None)
                    groups.append(grp)
                grp.loads.append((ld, U, sign))
            if groups and len(groups) <= 4:
                plan.rings[L.id] = groups

    # ---- schedules -------------------------------------------------------------------------------------------------------------------
    plan.uniform = [n for n in top.nodes if n.uniform]
    plan.invariants = [n for n in plan.uniform if n.kind == "inv"]
    plan.inputs = [n for n in top.nodes if n.kind == "in"]
    for r in all_regions:
        try:
            _schedule(plan, r, comp_of)
        except _BadCone as bc:
            if bc.event.kind == "ld":
                raise _Blame("feedback through a delay line whose addresses depend on memory", bc.event.ctx)
            src = g.event_src.get(bc.event.i)
            if src is not None and src not in event_ids and any(src == id(o) for o in origin.values()):
                no_event.add(src)               # (one of split_events' picks: the statement itself is walked next time)
                raise _Replan()
            if src is None or src not in event_ids:
                raise
            # (a statement the walk made an event of: it is none after all; whatever holds it gets its chance)
            event_ids.discard(src)
            no_event.add(src)
            raise _Replan()
    plan.items = top.items
    plan.uses_rand = RNG_INDEX in plan.outs
    # (measured per leaf, 1024 x 48 000: TSEQ 156 -> 132 ms -- its bands' mode switches leave whole arms idle; BedRock, DPT, ATTACK,
    #  ERBTilt, PsychoConvolver: no change -- nearly all of their guarded nodes sit on the arm their default settings take -- at
    #  the price of registers: the branches pin values the straight-line form could sink to their uses, ERBTilt 176 -> 346)
    want = os.environ.get("ZA_TPAR_BRANCHES")
    on = prog.name in UNIFORM_BRANCH_LEAVES if want is None else want == "1"
    plan.node_guard = _uniform_guards(plan, live, all_regions) if on else {}

    def count_items(kind, pred=lambda it: True):
        return sum(1 for r in all_regions for it in r.items if it[0] == kind and pred(it))

    plan.stats = {
        "nodes": len(live), "uniform": len(plan.uniform), "events": len(plan.events), "par": count_items("par"), "shift": count_items("shift"),
        "scan1": count_items("scan", lambda it: len(it[1].names) == 1), "scan2": count_items("scan", lambda it: len(it[1].names) == 2),
        "spec_loops": count_items("spec"),
        "spec_chains": sum(len(it[1]) for r in all_regions for it in r.items if it[0] == "spec"),
        "spec_switches": sum(len(c.conds) for r in all_regions for it in r.items if it[0] == "spec" for c in it[1]),
        "serial_loops": count_items("serial"), "wrapped_counters": count_items("modc"),
        "serial_chains": sum(len(it[1]) for r in all_regions for it in r.items if it[0] == "serial"),
        "serial_ops": sum(len([m for m in c.members if m.kind not in ("st", "lcin")]) for r in all_regions for it in r.items
                          if it[0] == "serial" for c in it[1]),
        "states": len(plan.st), "written": len(plan.outs), "rand_sites": g.rand_sites,
        "mem_cells": len(plan.cells), "delay_writes": len(plan.stores), "delay_reads": len(plan.loads),
        "loops": len(loops), "trip_cells": sum(len(L.cells) for L in loops), "trip_cells_stored": sum(len(L.cell_out) for L in loops),
        "gathers": sum(1 for ld in plan.loads if ld.loop is not None and ld.i in live),
        "holds": len(plan.holdvars), "guards": len(guards),
        "early_writes": sum(1 for s in plan.stores if s.mode == "early"), "sparse_writes": sum(1 for s in plan.stores if s.mode == "sparse"),
        "block": int(plan.has_block), "pending": int(plan.has_pending),
        "loop_guards": sum(len(L.guards) for L in loops),
    }
    return plan


UNIFORM_BRANCH_LEAVES = {"TSEQ"}


def _uniform_guards(plan: "Plan", live: Dict[int, N], all_regions) -> Dict[int, tuple]:
    """If-conversion computes both arms of every conditional. Where the condition is constant over a block (a mode switch, an
    `enabled` flag, `ir_ready`), the arm not taken is dead weight for the whole block: nodes of the frame whose every use is the
    same arm of selects on ONE such condition (directly, or through nodes that are themselves only used there) are emitted under
    a wave-uniform branch on it. Returns node id -> (condition node, arm taken when it is true?). Only plain lane-parallel nodes
    of the frame take part; anything a recurrence, a loop, a store, an event or an output refers to is computed always."""
    top = plan.top
    par = {it[1].i: it[1] for it in top.items if it[0] == "par" and it[1].kind == "op"}
    always = set()
    for o in list(plan.outs.values()) + list(plan.spl_out) + list(plan.events) + list(plan.guards) + list(plan.cells.values()):
        always.add(o.i)
    for st_ in plan.stores:
        for x in (st_.addr, st_.value, st_.pred):
            if x is not None:
                always.add(x.i)
    for r in all_regions:
        for c in r.comps:
            for x in (list(c.members) + list(getattr(c, "inputs", [])) + list(getattr(c, "ext", []))
                      + [y for row in (c.A or []) for y in row] + list(c.b or [])
                      + list(getattr(c, "gdep", [])) + list(getattr(c, "gnodes", [])) + list(getattr(c, "slice", []))
                      + list(getattr(c, "conds", []))):
                if isinstance(x, N):
                    always.add(x.i)
        if r.loop is not None:
            for x in r.ext:
                always.add(x.i)
    uses: Dict[int, List[tuple]] = {}
    for n in live.values():
        for k, a in enumerate(n.args):
            uses.setdefault(a.i, []).append((n, k))
        for a in n.extra:
            uses.setdefault(a.i, []).append((n, -1))
    memo: Dict[int, Optional[tuple]] = {}

    def guard(n: N) -> Optional[tuple]:
        if n.i in memo:
            return memo[n.i]
        memo[n.i] = None
        if n.i in always or n.i not in par or n.i not in uses:
            return None
        gs = set()
        for u, k in uses[n.i]:
            if u.i in par and u.op == "sel" and k in (1, 2) and u.args[0].uniform and u.args[0].loop is None and u.args[0].kind != "const":
                gs.add((u.args[0].i, k == 1))
            elif u.i in par and k >= 0:
                gs.add(guard(u))
            else:
                gs.add(None)
            if len(gs) > 1:
                return None
        g_ = next(iter(gs))
        memo[n.i] = g_
        return g_

    import sys
    lim = sys.getrecursionlimit()
    sys.setrecursionlimit(max(lim, 20000))
    try:
        out = {}
        for i, n in par.items():
            g_ = guard(n)
            if g_ is not None:
                out[i] = (live[g_[0]], g_[1])
    finally:
        sys.setrecursionlimit(lim)
    return out


def _sum_terms(n: N) -> List[N]:
    out, todo = [], [n]
    while todo:
        x = todo.pop()
        if x.kind == "op" and x.op == "+":
            todo.extend(reversed(x.args))
        else:
            out.append(x)
    return out


def _ring_form(ld: N, L: LoopInfo):
    """(S terms, P, U, sign, mask) when the load's address is za_addr(S.., ((P +- U) & mask)) as RingGroup describes it."""
    a = ld.args[0]
    if a.kind != "op" or a.op != "addr":
        return None
    terms = _sum_terms(a.args[0]) + _sum_terms(a.args[1])
    S = [t for t in terms if t.uniform and t.loop is None]
    rest = [t for t in terms if not (t.uniform and t.loop is None)]
    if len(rest) != 1 or rest[0].kind != "op" or rest[0].op != "&":
        return None
    d, mask = rest[0].args
    if not (mask.uniform and mask.loop is None) or d.kind != "op" or d.op not in ("+", "-"):
        return None
    x, y = d.args
    if d.op == "+" and x.loop is L and y.loop is not L:
        x, y = y, x
    if x.loop is L or y.loop is not L or not y.uniform or x.kind == "const":
        return None
    if _in_subtree(x, L):
        return None
    return S, x, y, (1 if d.op == "+" else -1), mask


def _schedule(plan: Plan, r: Region, comp_of: Dict[int, Component]):
    """Order of one chunk's work in region r: nodes as soon as their operands exist, scans as soon as their coefficients do,
    switched / serial recurrences that are ready together in one shared loop, nested loops as single items."""
    L = r.loop
    comps = r.comps
    for c in comps:
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
        elif c.kind == "modc":
            c.inputs = [c.modk, c.modn]
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
    done = set()

    def is_done(x: N) -> bool:
        return x.kind == "const" or x.i in done or not _in_subtree(x, L) or (L is None and x.uniform)

    if L is None:
        done |= {n.i for n in plan.inputs}
        pending = [n for n in r.nodes if not n.uniform and n.kind != "in"]
    else:
        done |= {n.i for n in r.nodes if n.kind == "phi"}
        pending = [n for n in r.nodes if n.kind != "phi"]
    comp_done = {id(c): False for c in comps}
    sub_done = {s.loop.id: False for s in r.subs}
    items: List[tuple] = []
    site_done: set = set()
    loads_in = {s.loop.id: any(ld.loop is not None and ld.loop.inside(s.loop) for ld in plan.loads) for s in r.subs}

    def flush_sites():
        for st_ in plan.stores:              # every write's span is known before the first read is resolved
            if st_.j not in site_done:
                site_done.add(st_.j)
                items.append(("site", st_))

    remaining = list(pending)
    later: List[N] = []
    cone_comps = None
    if L is None and (plan.events or plan.fb_loads):
        # what the event conditions need comes first, then the cut (the chunk ends before the first frame whose condition holds);
        # likewise the addresses of feedback reads and of the writes into their buffers (the chunk ends before the first frame that
        # would read what an earlier frame of the chunk writes)
        cone: Dict[int, N] = {}
        cone_comps = set()
        todo = [(e, e) for e in plan.events]
        for ld in plan.fb_loads:
            todo.append((ld.args[0], ld))
            for st_ in plan.stores:
                if st_.j in ld.fb:
                    todo.append((st_.addr, ld))
                    if st_.pred is not None:
                        todo.append((st_.pred, ld))
        while todo:
            x, root = todo.pop()
            if x.i in cone or x.kind == "const" or (x.uniform and x.loop is None):
                continue
            if x.kind in ("ld", "lout", "lcin") or x.loop is not None:
                raise _BadCone(root)
            cone[x.i] = x
            todo.extend((y, root) for y in x.args + x.extra)
            if x.kind == "st" and x.name in r.st and r.st[x.name] is x:
                c = comp_of.get(x.i)
                if c is None:
                    todo.append((r.outs[x.name], root))
                elif id(c) not in cone_comps:
                    cone_comps.add(id(c))
                    todo.extend((y, root) for y in c.members)
                    todo.extend((y, root) for y in c.inputs)
        later = [n for n in remaining if n.i not in cone]
        remaining = [n for n in remaining if n.i in cone]
    guard = 0
    while True:
        if cone_comps is not None and not remaining:
            items.append(("cut",))
            cone_comps = None
            remaining = later
            continue
        if not (remaining or not all(sub_done.values())):
            break
        guard += 1
        if guard > 10 * len(r.nodes) + 100:
            raise AssertionError("scheduler made no progress")
        progressed = False
        nxt = []
        for n in remaining:
            if n.kind in ("st", "lcin") and n.name in r.st and r.st[n.name] is n:
                c = comp_of.get(n.i)
                if c is None:                              # delayed signal
                    if is_done(r.outs[n.name]):
                        items.append(("shift", n.name))
                        done.add(n.i)
                        progressed = True
                    else:
                        nxt.append(n)
                elif comp_done[id(c)]:
                    done.add(n.i)
                    progressed = True
                else:
                    nxt.append(n)
                continue
            if n.kind == "lout":
                if sub_done[n.val]:
                    done.add(n.i)
                    progressed = True
                else:
                    nxt.append(n)
                continue
            deps = (plan.loop_by_id[n.val].cells[n.name],) if n.kind == "lcin" else n.args + n.extra
            if all(is_done(a) for a in deps):
                if n.kind == "ld" and L is None:
                    flush_sites()
                items.append(("par", n))
                done.add(n.i)
                progressed = True
            else:
                nxt.append(n)
        remaining = nxt
        for s in r.subs:
            if cone_comps is not None:
                break
            if not sub_done[s.loop.id] and all(is_done(x) for x in s.ext):
                if loads_in[s.loop.id] and L is None:
                    flush_sites()
                items.append(("loop", s))
                sub_done[s.loop.id] = True
                progressed = True
        # scans as soon as their coefficients exist (they are lane-parallel work too)
        for c in comps:
            if cone_comps is not None and id(c) not in cone_comps:
                continue
            if not comp_done[id(c)] and c.kind in ("scan", "modc") and all(is_done(x) for x in c.inputs):
                items.append((c.kind, c))
                comp_done[id(c)] = True
                progressed = True
        if progressed:
            continue
        # only switched / serial recurrences can move now: every one of a kind that is ready shares one loop
        ready = []
        for kind in ("spec", "serial"):
            ready = [c for c in comps if not comp_done[id(c)] and c.kind == kind and all(is_done(x) for x in c.inputs)
                     and (cone_comps is None or id(c) in cone_comps)]
            if ready:
                break
        if not ready:
            raise AssertionError("dependency cycle outside the recurrences")
        items.append((kind, ready))
        for c in ready:
            comp_done[id(c)] = True
    if L is None:
        flush_sites()
    r.items = items


def _classify(g: FrameGraph, reg: Region, c: Component, ci: int = 0, live=None):
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
            gn.loop = reg.loop
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
            if n.kind in ("st", "lcin"):
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
            r = aff(reg.outs[nm])
            if r is None:
                return None
            rows.append(r)
        return rows, conds, gnodes

    if d == 1 and reg.loop is None and not os.environ.get("ZA_TPAR_NO_MODC"):
        # a wrapped counter, pos = (pos + K) % N with K and N constant over a block (ring positions): over non-negative integers
        # the t-th iterate is (pos + t K) % N -- exact, whatever the order; checked per chunk, the serial loop otherwise
        o0, st_ = reg.outs[names[0]], reg.st[names[0]]
        o, gate, extra_m = o0, None, set()
        if (o.kind == "op" and o.op == "sel" and o.args[0].uniform and o.args[0].loop is None and o.args[0].i not in mem
                and (o.args[1] is st_) != (o.args[2] is st_)):
            # the step under a block-constant condition (`enabled ? ( ...; pos = (pos + 1) & mask )`): a step of 0 where it is off
            gate, o, extra_m = (o.args[0], o.args[2] is st_), (o.args[1] if o.args[2] is st_ else o.args[2]), {o0.i}
        if o.kind == "op" and o.op in ("%", "&") and o.args[1].uniform and o.args[1].loop is None:
            a = o.args[0]
            if a.kind == "op" and a.op == "+" and st_ in a.args:
                k_ = a.args[1] if a.args[0] is st_ else a.args[0]
                if k_ is not st_ and k_.uniform and k_.loop is None and mem == {st_.i, a.i, o.i} | extra_m:
                    # (pos + K) & M with M = 2^k - 1 is (pos + K) % (M + 1) over non-negative integers (checked per chunk)
                    c.kind, c.modk = "modc", k_
                    if gate is not None:
                        c.modk = g.sel(gate[0], k_, g.ZERO) if gate[1] else g.sel(gate[0], g.ZERO, k_)
                    c.modn = o.args[1] if o.op == "%" else g.op("+", o.args[1], g.ONE)
                    c.modmask = o.op == "&" or gate is not None       # (the start must lie inside [0, N) then)
                    c.modpow2 = o.op == "&"
                    return
    if d == 1 and _persistent_rounding(g, reg, c, mem):
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
        c.slice = [sl[i] for i in sorted(sl) if sl[i].kind not in ("st", "lcin")]


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


def _fold(n: N, memo: Optional[Dict[int, Optional[float]]] = None) -> Optional[float]:
    """Value of a node built from constants only, whatever the operators (None: not a constant)."""
    if n.kind == "const":
        return float(n.val)
    if n.kind != "op" or not n.args or n.op in ("mtout", "addr"):
        return None
    memo = {} if memo is None else memo
    if n.i in memo:
        return memo[n.i]
    memo[n.i] = None
    v = []
    for a in n.args:
        x = _fold(a, memo)
        if x is None:
            return None
        v.append(x)
    with np.errstate(all="ignore"):
        r = float(_np_op(n.op, [np.float64(x) for x in v]))
    memo[n.i] = r
    return r


def _persistent_rounding(g: FrameGraph, reg: Region, c: Component, mem) -> bool:
    """A recurrence y = y + b with a fractional step keeps every rounding error it ever made (coefficient exactly 1: nothing
    decays), and scripts put thresholds exactly where such sums are meant to land -- `pos += 1 / N; pos < 1 ? ...` reaches
    1 after N steps only up to rounding, so the frame at which the test flips depends on the ORDER of the additions. A scan
    re-associates them. Such components therefore keep their serial loop (exact order); integer-valued steps (counters,
    hold timers) are exact in any order and stay scans, |a| < 1 forgets its rounding, and a step computed from this frame's
    input (`energy += x * x`) has no value it is meant to land on: thresholds on those are generic.
    Decided on the branch-wise affine forms of the new state: (coefficient on itself, constant term) per path through ?: /
    min / max; any path with coefficient 1 and a term that is not an integer literal and is built from invariants and states
    only (a rate that changes now and then is still a rate) marks the component."""
    nm = c.names[0]
    limit = 256

    def forms(n: N):
        if n.i not in mem:
            return [(g.ZERO, n)]
        if n.kind in ("st", "lcin"):
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

    sig_memo: Dict[int, bool] = {}

    def from_input(n: N) -> bool:
        """Built from this frame's audio (an input sample or a delay-line read), not only from invariants and states."""
        if n.i in sig_memo:
            return sig_memo[n.i]
        sig_memo[n.i] = False
        if n.kind in ("in", "ld"):
            r = True
        elif n.kind in ("phi", "lout"):
            L = g.loop_of[n.val]
            r = from_input(L.init[n.name]) or from_input(L.next[n.name])
        elif n.kind == "op" and n.op == "sel":       # (which value is taken may follow the input; the values are what is summed)
            r = from_input(n.args[1]) or from_input(n.args[2])
        else:
            r = any(from_input(a) for a in n.args)
        sig_memo[n.i] = r
        return r

    fs = forms(reg.outs[nm])
    if fs is None:
        return False                              # not affine even branch-wise: the classification below decides
    for k, v in fs:
        if _const_value(k) == 1.0:
            cv = _const_value(v)
            if cv is not None and cv == math.floor(cv):
                continue
            if not from_input(v) or os.environ.get("ZA_TPAR_STRICT_SUMS"):
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


class _Emit:
    """Kernel text of one plan."""

    def __init__(self, plan: Plan, prog: Program, kernel_macro: str):
        self.plan, self.prog, self.km = plan, prog, kernel_macro
        self.g = plan.g
        self.L: List[str] = []
        p = plan
        # Block-constant values: a few dozen fit the scalar registers (ZT_UNI); past that the compiler spills them into lanes of
        # vector registers and every use costs two v_readlane. Large scripts keep them in LDS instead: one broadcast ds_read_b64
        # per use, the `zo` offset (an opaque 0 set per chunk) keeping the reads inside the iteration.
        self.n_uni = sum(1 for n in p.uniform if n.kind not in ("const", "hold"))
        mode = os.environ.get("ZA_TPAR_ULDS", "auto")
        self.ulds = mode == "1" or (mode == "auto" and self.n_uni > ULDS_THRESHOLD)
        self.uslot = {n.i: k for k, n in enumerate(x for x in p.uniform if x.kind not in ("const", "hold"))}
        self.in_loop = False
        self.bctx = None
        self.sctx = None
        self.cname = {name: f"c{k}" for k, name in enumerate(p.st)}
        self.hname = {name: f"h{k}" for k, name in enumerate(p.holdvars)}
        self.cell_addrs: List[N] = []
        for a in p.cells.values():
            if a not in self.cell_addrs:
                self.cell_addrs.append(a)
        self.lcell_loops = [L for L in p.loops if L.cells or L.id in p.rings or L.guards]
        self.ring_lds: Dict[int, str] = {}        # ld node id -> its LDS address, while a loop's staged form is being emitted
        self.ring_u: Dict[int, N] = {}
        self.has_mem = bool(p.cells or p.stores or p.loads or self.lcell_loops)
        self.has_streams = bool(p.stores)
        self.has_serial = p.has_block or p.has_pending
        self.has_events = bool(p.events)
        self.loop_guards = any(L.guards for L in p.loops)
        self.has_fb = bool(getattr(p, "fb_loads", []))
        self.has_cut = bool(p.events) or self.has_fb       # a chunk may end early (tn shrinks at the scheduler's "cut")
        self.segmented = bool(p.events or p.guards or self.loop_guards or self.has_fb)    # blocks are walked in segments, single frames in between run serially
        self.has_abort = bool(p.cells or p.stores or p.loads or p.guards or self.lcell_loops or p.events or self.loop_guards)
        self.early = [s for s in p.stores if s.mode == "early"]
        self.phi_name: Dict[int, str] = {}        # phi / lout node id -> C++ variable
        for L in p.loops:
            for k, v in enumerate(L.order):
                nm = f"p{L.id}_{k}"
                self.phi_name[L.phis[v].i] = nm
                if v in L.louts:
                    self.phi_name[L.louts[v].i] = nm

    # -- names ---------------------------------------------------------------------------------------------------------------
    def ref(self, n: N) -> str:
        if n.kind == "const":
            return c_double(n.val)
        if n.kind == "hold":
            return "ZT_HOLD"
        if self.sctx is not None and n.loop is self.sctx[0] and n.kind != "lout":      # a strip of 64 trips (emit_strip)
            Lp, mode = self.sctx
            if isinstance(mode, tuple):               # sub-trip u of a group of trips whose fetches go out together
                u = mode[1]
                if n.uniform:
                    return f"e{n.i}_{u}"
                if n.kind == "phi":
                    if u == 0:
                        return self.phi_name[n.i]
                    self.sctx = (Lp, ("g", u - 1))
                    try:
                        return self.ref(Lp.next[n.name])
                    finally:
                        self.sctx = (Lp, mode)
                return f"n{n.i}_{u}"
            if n.uniform:
                if mode == "vec":                     # lane j = trip k0 + j
                    return f"t{n.i}"
                return f"e{n.i}"                      # per trip: the strip's value for this trip (v_readlane)
            if n.kind == "phi":
                return self.phi_name[n.i]
            return f"k{n.i}" if mode == "cold" else f"n{n.i}"
        if self.bctx is not None and n.loop is self.bctx[0] and n.kind != "lout":      # a batch of trips (emit_batched): sub-trip names
            Lp, u = self.bctx
            if n.kind == "phi":
                return self.phi_name[n.i] if u == 0 else self.bref(Lp.next[n.name], u - 1)
            return f"n{n.i}_{u}"
        if n.kind in ("phi", "lout"):
            return self.phi_name[n.i]
        if n.uniform and n.loop is None:
            if self.ulds and self.in_loop:
                return f"zt_u[{self.uslot[n.i]} + zo]"
            return f"u{n.i}"
        return f"n{n.i}"

    def bref(self, n: N, u: int) -> str:
        """Name of node n in sub-trip u of the batch being emitted."""
        save, self.bctx = self.bctx, (self.bctx[0], u)
        try:
            return self.ref(n)
        finally:
            self.bctx = save

    def inv_src(self, name: str) -> str:
        p, prog = self.plan, self.prog
        k = is_slider_name(name)
        if k is not None:
            return f"b.sliders[{k - 1} * b.sl_se + inst * b.sl_si]"
        if name == "srate":
            return "b.srate"
        if name == "samplesblock":
            return "(double)bn"
        if name in ("midi_bus", "ext_midi_bus", RNG_INDEX) or name.startswith("memw@"):
            return "0.0"
        if name in p.cells:
            return f"(ca{p.cells[name].i} < mcap ? memp[ca{p.cells[name].i} * mse] : 0.0)"
        k = is_spl_name(name)
        if k is not None:
            return f"b.spl[{k} * b.sl_se + inst * b.sl_si]"
        return f"b.vars[{prog.vars[name]} * b.var_se + inst * b.var_si]"

    def dst(self, name: str) -> str:
        p, prog = self.plan, self.prog
        if name in p.cells:
            return f"memp[ca{p.cells[name].i} * mse]"
        k = is_spl_name(name)
        if k is not None:
            return f"b.spl[{k} * b.sl_se + inst * b.sl_si]"
        return f"b.vars[{prog.vars[name]} * b.var_se + inst * b.var_si]"

    def carry(self, reg: Region, nm: str) -> str:
        """C++ name of the wave-uniform value a recurrence state carries into the chunk."""
        if reg.loop is None:
            return self.cname[nm]
        return f"lc{reg.loop.cin[nm].i}"

    # -- the kernel ----------------------------------------------------------------------------------------------------------
    def emit(self) -> str:
        p, L, ref = self.plan, self.L, self.ref
        km = self.km
        L.append("// ---- time-parallel kernel: one wavefront per instance, lane = frame (generated by zajit/tpar.py) ----")
        L.append(f"// schedule: {p.stats}")
        L.append("#ifndef ZT_SPEC_MAX")
        L.append(f"#define ZT_SPEC_MAX {SPEC_MAX}")
        L.append("#endif")
        L.append(f"#define ZT_SPEC_TOL {SPEC_TOL!r}")
        L.append("#ifndef ZT_SECTION_FN")
        L.append("#ifdef ZA_INLINE_ALL")
        L.append("#define ZT_SECTION_FN __device__ inline __attribute__((always_inline))")
        L.append("#else")
        L.append("#define ZT_SECTION_FN __device__ __attribute__((noinline))")
        L.append("#endif")
        L.append("#endif")
        L.append("#ifndef ZT_UNI")
        L.append("#define ZT_UNI(x) zt_uniform(x)")
        L.append("#endif")
        self.stamps = bool(os.environ.get("ZA_TPAR_STAMPS"))
        if self.stamps:      # in-kernel phase clock (tools/tpar_stamps.py): cycles per phase, summed over the waves of a launch
            L.append("__device__ unsigned long long zt_stamps[64];")
            L.append('extern "C" void zab_tpar_stamps(unsigned long long* out, int reset) {')
            L.append("  if (out) (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(zt_stamps), sizeof(zt_stamps));")
            L.append("  if (reset) { unsigned long long z[64] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(zt_stamps), z, sizeof(z)); }")
            L.append("}")
            L.append("#define ZT_STAMP(k) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); if (lane == 0) atomicAdd(&zt_stamps[k], t_ - zt_t0); zt_t0 = __builtin_amdgcn_s_memtime(); }")
        else:
            L.append("#define ZT_STAMP(k)")
        if self.has_serial:
            self.emit_serial_fn()
        if self.segmented:
            self.emit_frame_fn()
        L.append(f'extern "C" __global__ void __launch_bounds__(64) ZA_OCC {km}(ZabBatch b, ZabAudio a) {{')
        if self.has_serial or self.segmented:
            L.append("  ZA_KERNEL_ENTRY();")
        L.append("  const int lane = threadIdx.x;")
        L.append("  const int64_t inst = blockIdx.x;")
        L.append("  const int64_t frames = a.frames;")
        L.append("  if (frames <= 0 || inst >= b.n_inst) return;")
        if self.stamps:
            L.append("  unsigned long long zt_t0 = __builtin_amdgcn_s_memtime();")
        if p.uses_rand:
            L.append("  __shared__ uint32_t zt_mt[2 * ZT_MT_N];      // rand(): current and next generation of the instance's MT19937")
            L.append("  uint32_t* const zt_gmt = b.mt + inst * b.mt_si;")
            L.append("  int zt_pos0 = 0;")
        if self.ulds:
            L.append(f"  __shared__ double zt_u[{max(1, self.n_uni)}];")
        # recurrences whose coefficient is constant over a block: one LDS row of per-lane weights per distinct coefficient
        self.inv_coefs: List[N] = []
        for it in p.top.items:
            if it[0] == "scan" and len(it[1].names) == 1:
                a = it[1].A[0][0]
                if a.uniform and a.kind != "const" and a not in self.inv_coefs and not os.environ.get("ZA_TPAR_NO_INVSCAN"):
                    self.inv_coefs.append(a)
        # coupled pairs with a block-constant matrix (biquads): one table per distinct matrix, within an LDS budget that still
        # lets four wavefronts share a CU (one per SIMD, the 1024-instance case)
        self.inv_mats: List[tuple] = []
        budget = 36 * 1024 - len(self.inv_coefs) * (64 + 4) * 8 - (2 * 624 * 4 if p.uses_rand else 0)
        for it in p.top.items:
            if it[0] == "scan" and len(it[1].names) == 2 and not os.environ.get("ZA_TPAR_NO_INVSCAN"):
                key = tuple(x for row in it[1].A for x in row)
                if (all(x.uniform or x.kind == "const" for x in key) and key not in self.inv_mats
                        and (len(self.inv_mats) + 1) * (12 + 8 * 64) * 8 <= budget):
                    self.inv_mats.append(key)
        if self.inv_mats:
            L.append(f"  __shared__ double zt_m[{len(self.inv_mats)} * ZT_MAT_TABLE_DOUBLES];      // per block-constant 2 x 2 matrix: powers and per-lane weights")
        if self.inv_coefs:
            L.append(f"  __shared__ double zt_w[{len(self.inv_coefs)} * 64];      // a^((lane & 15) + 1) per block-constant coefficient")
            L.append(f"  __shared__ double zt_q[{len(self.inv_coefs)} * 4];       // a^2, a^4, a^8, a^16")
        if p.rings:
            L.append("  __shared__ double zt_ring[ZT_RING_DOUBLES];     // a chunk's window of the ring a loop gathers from (RingGroup)")
        self.cell_loops = [L_ for L_ in p.loops if L_.cell_out and not os.environ.get("ZA_TPAR_NO_LDS_CELLS")]
        # a chunk that breaks a run-time condition is handed back as it began: where that can happen in a chunk whose loops have
        # already moved their cells on (reads / writes at moving addresses are checked as they are met), the cells are copied aside
        # at the start of every chunk (the second half of zt_cells)
        self.cell_undo = bool(self.cell_loops) and bool(p.stores or p.loads)
        if self.cell_loops:
            L.append(f"  __shared__ double zt_cells[{'2 * ' if self.cell_undo else ''}ZT_CELL_DOUBLES];    // the per-trip cells of a block's loops: staged per block, kept here from chunk to chunk")
            if self.cell_undo:
                L.append("  int zt_cn = 0;     // cells staged (the copy of a chunk's start sits ZT_CELL_DOUBLES further on)")
        if self.has_abort:
            L.append(f"  __shared__ double zt_snap[{max(1, len(self.cname))}];")
        if self.has_mem:
            L.append("  // mem[]: block-constant addresses are cells (named state kept in registers), addresses that follow a uniform loop's")
            L.append("  // counters are per-trip cells, moving ones are delay lines")
            L.append("  double* const memp = b.mem + inst * b.mem_si;")
            L.append("  const int64_t mse = b.mem_se, mcap = b.mem_cap;")
        L.append(f"  const float* const in_ = a.in + inst * {p.nch} * a.frame_stride;")
        L.append(f"  float* const out_ = a.out + inst * {p.nch} * a.frame_stride;")
        if self.has_serial:
            L.append("  uint64_t zt_pend_seen = 0;     // slider masks the script raised in any block of this launch (host: consumeDspSliderChanges)")
        L.append("  // the audio of a chunk is read one iteration ahead, so that its HBM latency is hidden behind the previous chunk's work")
        for n in p.inputs:
            L.append(f"  float x{n.i} = lane < frames ? in_[{int(n.val)} * a.frame_stride + lane] : 0.0f;")
        # ZT_PIN: an empty asm that takes the prefetched registers, i.e. the point where the compiler waits for their loads. It
        # sits before the loop and, in the loop, before the chunk's stores: the loads have had the whole chunk to land, and no
        # path reaches the top of the loop with them pending -- there the wait would be a full vmcnt(0), taken right after the
        # NEXT chunk's loads were issued (every chunk would pay an HBM round trip).
        self.pin = ", ".join(f'"+v"(x{n.i})' for n in p.inputs)
        if self.pin:
            L.append(f"  asm volatile(\"\" : {self.pin});")
        self.cell_slot: Dict[tuple, int] = {}
        for lid, groups in p.rings.items():
            L.append(f"  bool zrok{lid} = true;     // ring reads of loop {lid}: offsets of every trip (integers), per read")
            for grp in groups:
                for ld, _, _ in grp.loads:
                    L.append(f"  int zro_lo{ld.i} = 2147483647, zro_hi{ld.i} = -2147483647;")
        for Lp in self.cell_loops:
            keys = self.pass_keys(Lp)
            for j, k in enumerate(keys):
                self.cell_slot[(Lp.id, k)] = j
            L.append(f"  int zln{Lp.id} = 0, zlo{Lp.id} = 0; bool zlds{Lp.id} = false;     // loop {Lp.id}: trips, its place in zt_cells, staged or not")
        L.append(f"  const int64_t blk = {'a.block > 0 ? (int64_t)a.block : frames' if p.has_block else 'frames'};   // a script without @block sees one block per launch")
        self.pass_memo: Dict[int, tuple] = {}
        self.memo_off: Dict[int, int] = {}
        self.site_off: Dict[int, int] = {}
        tot_m = tot_s = 0
        for Lp in self.lcell_loops:                  # (sizes are needed before the passes are written: count their inputs first)
            self.memo_off[Lp.id] = tot_m
            tot_m += self.pass_inputs(Lp)
            self.site_off[Lp.id] = tot_s
            tot_s += 5 * len(self.pass_keys(Lp))
        if tot_m:
            L.append(f"  __shared__ unsigned long long zt_memo[{tot_m}];     // what the address passes read last time (bit patterns)")
        if tot_s:
            L.append(f"  __shared__ long long zt_site[{tot_s}];     // per address expression of a loop with per-trip cells: a0, stride, previous, lo, hi")
        memo_at = len(L)
        if self.segmented:
            # a block is walked in segments: each ends at the block's end or right before a frame an event falls on (or starts
            # with a frame a guard holds for); that frame runs serially (zt_frame) and the next segment starts behind it, with
            # the per-block values computed afresh
            L.append("  int64_t zt_bend = 0, zt_bn = 0, zt_nev = 0;")
            L.append("  for (int64_t pos = 0; pos < frames; ) {")
            L.append("    const bool zt_new = pos >= zt_bend;")
            L.append("    if (zt_new) { zt_bn = frames - pos < blk ? frames - pos : blk; zt_bend = pos + zt_bn; }")
            L.append("    const int64_t bn = zt_bn;")
            L.append("    int64_t bend = zt_bend, zt_evf = -1;")
            L.append("    bool zt_fbc = false;     // the segment ended at a feedback read's cut: the next one starts right behind it, no serial frame")
            L.append("    ZT_STAMP(7)")
            if self.has_serial:
                L.append("    if (zt_new)")
                self.emit_serial_phase()
        else:
            L.append("  for (int64_t pos = 0; pos < frames; pos += blk) {")
            L.append("    const int64_t bn = frames - pos < blk ? frames - pos : blk;")
            L.append("    const int64_t bend = pos + bn;")
            L.append("    ZT_STAMP(7)")
            if self.has_serial:
                self.emit_serial_phase()
        L.append("    ZT_STAMP(0)")
        self.emit_block_prologue()
        L.append("    ZT_STAMP(1)")
        self.emit_chunk_loop()
        if self.segmented:
            L.append("    if (zt_evf < 0) {")
            L.append("      pos = bend;")
            if self.has_fb:
                L.append("      if (zt_fbc) {   // (the audio read ahead was that of frame f0 + 64)")
                for n in p.inputs:
                    L.append(f"        x{n.i} = pos + lane < frames ? in_[{int(n.val)} * a.frame_stride + pos + lane] : 0.0f;")
                L.append("      }")
            L.append("      continue;")
            L.append("    }")
            self.emit_serial_frame("    ")
        L.append("  }")
        decl = []
        for lid, (memo, xs) in self.pass_memo.items():
            decl.append(f"  bool zpv{lid} = false, zpg{lid} = false; int64_t zph{lid} = 0;     // address pass of loop {lid}: done for these inputs, high-water mark it found, an event of its body is due")
        L[memo_at:memo_at] = decl
        if self.has_serial:
            L.append("  if (lane == 0 && zt_pend_seen) b.pend[3 * (int64_t)b.n_pad + inst] |= zt_pend_seen;")
        if self.has_abort:
            L.append("  if (lane == 0) b.resume[inst] = frames;")
        L.append("}")
        if self.has_abort:
            self.emit_tail()
        L.append("static int32_t za_fast_applies(const ZabBatch* b, const ZabAudio* a) { (void)b; return a->frames > 0 ? 1 : 0; }")
        L.append("static hipError_t za_launch_fast(const ZabBatch* b, const ZabAudio* a, hipStream_t st) {")
        L.append(f"  hipLaunchKernelGGL({km}, dim3(b->n_inst), dim3(64), 0, st, *b, *a);")
        if self.has_abort:
            L.append(f"  hipLaunchKernelGGL({km[:-1]}_tail), dim3((b->n_inst + 63) / 64), dim3(64), 0, st, *b, *a);")
        L.append("  return hipGetLastError();")
        L.append("}")
        return "\n".join(L) + "\n"

    # -- between the blocks: @block and the pending-mask @slider, run by the wavefront with the leaf's section code -----------------
    def emit_serial_fn(self):
        """`zt_serial`: what jsfx_process_block does before a block's frames (dsp_jsfx_aot.py:5766-5804) -- samplesblock, @block,
        @slider if a mask is pending -- as a function of its own, called by the kernel between the blocks. The state is in vars[] /
        mem[] there (every block ends with its values stored), so the section code runs on it as in the generic kernel: on lane 0,
        or -- leaves with cooperative builtins -- on all 64 lanes as replicas of the instance. Only what @block (and @slider, where
        the script can raise a mask) names is loaded, only what they assign is stored: a script's few hundred variables need not
        all be live across its @block. Not inlined: the section code keeps its own registers and stack frame instead of sharing
        the kernel's allocation (CMD's @block inside the kernel body put it at 512 registers plus scratch, where the device compiler
        has produced wrong code before: DESIGN.md "compiler hazard")."""
        p, L = self.plan, self.L
        prog = self.prog
        secs = list(prog.sections.get("block", [])) if p.has_block else []
        if p.has_pending:
            secs += list(prog.sections.get("slider", []))
        body = ["    s.samplesblock = (double)bn;", "    s.block_size = (int)bn;"]
        if p.has_block:
            body += ["#if ZA_USES_MSG", "    za_msg_begin_block(s);", "#endif", "    za_section_block(s);"]
        if p.has_pending:
            body.append("    if (s.pend_change | s.pend_automate | s.pend_automate_end) za_section_slider(s);")
            body.append("    seen = s.pend_change | s.pend_automate | s.pend_automate_end;")
            body.append("    s.pend_change = s.pend_automate = s.pend_automate_end = 0;")
        L.append("// between the blocks: @block and the pending-mask @slider, run by the wavefront with the leaf's section code")
        self.emit_section_fn("zt_serial", "const int64_t bn", secs, body, [])

    def emit_frame_fn(self):
        """`zt_frame`: one whole frame of @sample with the leaf's section code -- the frame an event falls on (split_events). The
        frames before it have left every variable they write in vars[] / mem[] (a leaf with events stores them at the end of every
        chunk), so the frame runs on that state exactly as the generic kernel's would."""
        p, L = self.plan, self.L
        secs = list(self.prog.sections.get("sample", []))
        body = ["    s.samplesblock = (double)bn;", "    s.block_size = (int)bn;"]
        body += [f"    s.spl[{ch}] = (double)in_[{ch} * fs + t];" for ch in range(p.nch)]
        body.append("    za_section_sample(s);")
        tail = [f"      out_[{ch} * fs + t] = (float)s.spl[{ch}];" for ch in range(p.nch)]
        L.append("// the frame an event falls on, run by the wavefront with the leaf's section code")
        self.emit_section_fn("zt_frame", "const int64_t bn, const float* __restrict__ in_, float* __restrict__ out_, const int64_t fs, const int64_t t",
                             secs, body, tail)

    def emit_serial_frame(self, ind: str):
        """Frame zt_evf with the section code, then on to the next segment (inside the kernel's loop over pos)."""
        p, L = self.plan, self.L
        L.append(f"{ind}// when such frames come thicker than one in 16 this kernel is the wrong tool: the serial code takes the rest")
        L.append(f"{ind}if (++zt_nev * 16 > zt_evf + 256) {{")
        self.emit_leave(ind + "  ", "zt_evf")
        L.append(f"{ind}}}")
        L.append(f"{ind}__builtin_amdgcn_fence(__ATOMIC_RELEASE, \"workgroup\");")
        L.append(f"{ind}__builtin_amdgcn_wave_barrier();")
        L.append(f"{ind}__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, \"workgroup\");")
        L.append(f"{ind}(void)zt_frame((const ZabBatch*)__builtin_amdgcn_kernarg_segment_ptr(), inst, lane, bn, in_, out_, a.frame_stride, zt_evf);")
        L.append(f"{ind}__builtin_amdgcn_fence(__ATOMIC_RELEASE, \"workgroup\");")
        L.append(f"{ind}__builtin_amdgcn_wave_barrier();")
        L.append(f"{ind}__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, \"workgroup\");")
        L.append(f"{ind}pos = zt_evf + 1;")
        for n in p.inputs:
            L.append(f"{ind}x{n.i} = pos + lane < frames ? in_[{int(n.val)} * a.frame_stride + pos + lane] : 0.0f;")
        L.append(f"{ind}ZT_STAMP(6)")
        L.append(f"{ind}continue;")

    def emit_section_fn(self, fname: str, params: str, secs, body: List[str], tail: List[str]):
        p, L = self.plan, self.L
        prog = self.prog
        rd = sorted(prog.vars[nm] for nm in _read_names(prog, secs) if nm in prog.vars)
        wr = sorted(prog.vars[nm] for nm in (_assigned_names(prog, secs) | _outarg_names(prog, secs)) if nm in prog.vars)
        if os.environ.get("ZA_TPAR_FULL_STATE"):
            rd = wr = list(range(prog.nvars))
        L.append(f"static ZT_SECTION_FN unsigned long long {fname}(const ZabBatch* __restrict__ zt_pb, const int64_t inst, const int lane, {params}) {{")
        L.append("  const ZabBatch& b = *zt_pb;")
        L.append("  unsigned long long seen = 0;")
        L.append("#ifdef ZA_REPLICAS")
        L.append("  const bool zt_run = true;")
        L.append("#else")
        L.append("  const bool zt_run = lane == 0;")
        L.append("#endif")
        L.append("  if (zt_run) {")
        L.append("    ZaS s;")
        L.append("    za_state_bind(s, b, (int)inst);")
        for k0 in range(0, len(rd), 8):
            L.append("    " + " ".join(f"s.v[{k}] = b.vars[{k} * b.var_se + inst * b.var_si];" for k in rd[k0:k0 + 8]))
        L.append("#define ZA_X(k) s.sl[k] = b.sliders[(k) * b.sl_se + inst * b.sl_si];")
        L.append("    ZA_FOR_USED_SL(ZA_X)")
        L.append("#undef ZA_X")
        L.append("#define ZA_X(k) s.spl[k] = b.spl[(k) * b.sl_se + inst * b.sl_si];")
        L.append("    ZA_FOR_USED_SPL(ZA_X)")
        L.append("#undef ZA_X")
        L.append("#ifdef ZA_REPLICAS")
        L.append("    s.replica = lane != 0 ? 1u : 0u; s.rep_i = (uint32_t)lane; s.rep_n = 64u; s.rep_stride = 1u;")
        L.append("#endif")
        L.extend(body)
        L.append("    if (lane == 0) {")
        L.extend(tail)
        for k0 in range(0, len(wr), 8):
            L.append("      " + " ".join(f"b.vars[{k} * b.var_se + inst * b.var_si] = s.v[{k}];" for k in wr[k0:k0 + 8]))
        L.append("#define ZA_X(k) b.sliders[(k) * b.sl_se + inst * b.sl_si] = s.sl[k];")
        L.append("      ZA_FOR_USED_SL(ZA_X)")
        L.append("#undef ZA_X")
        L.append("#define ZA_X(k) b.spl[(k) * b.sl_se + inst * b.sl_si] = s.spl[k];")
        L.append("      ZA_FOR_USED_SPL(ZA_X)")
        L.append("#undef ZA_X")
        L.append("      b.mem_high[inst] = s.mem_high; b.mem_need[inst] = s.mem_need; b.mti[inst] = s.mti; b.err[inst] = s.err;")
        L.append("      b.pend[inst] = s.pend_change; b.pend[b.n_pad + inst] = s.pend_automate; b.pend[2 * (int64_t)b.n_pad + inst] = s.pend_automate_end;")
        L.append("      b.vis_mask[inst] = s.vis_mask; b.vis_init[inst] = s.vis_init;")
        L.append("      if (b.gmem_att) b.gmem_att[inst] = s.gmem_attached;")
        L.append("    }")
        L.append("  }")
        L.append("  return seen;")
        L.append("}")

    def emit_serial_phase(self):
        p, L = self.plan, self.L
        if p.has_block:
            L.append("    {")
        else:
            L.append("    if (pos == 0 && (b.pend[inst] | b.pend[b.n_pad + inst] | b.pend[2 * (int64_t)b.n_pad + inst]) != 0ull) {")
        L.append("      // (the ZabBatch the function reads is this kernel's own first argument, where it lies in the kernarg segment)")
        L.append("      zt_pend_seen |= zt_serial((const ZabBatch*)__builtin_amdgcn_kernarg_segment_ptr(), inst, lane, bn);")
        L.append("      __builtin_amdgcn_fence(__ATOMIC_RELEASE, \"workgroup\");")
        L.append("      __builtin_amdgcn_wave_barrier();")
        L.append("      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, \"workgroup\");")
        L.append("    }")

    def emit_leave(self, ind: str, frame: str):
        """Hand the rest of the launch, from `frame` on, to the serial code (state already in vars[] / mem[])."""
        L = self.L
        L.append(f"{ind}if (lane == 0) b.resume[inst] = {frame};")
        if self.has_serial:
            L.append(f"{ind}if (lane == 0 && zt_pend_seen) b.pend[3 * (int64_t)b.n_pad + inst] |= zt_pend_seen;")
        L.append(f"{ind}return;")

    def emit_block_prologue(self):
        p, L, ref = self.plan, self.L, self.ref
        if p.uses_rand:
            L.append("    __syncthreads();")
            L.append("    zt_pos0 = zt_mt_begin(zt_mt, zt_gmt, b.mt_se, b.mti[inst], lane);     // (a block ends with the generator's state stored)")
        L.append("    // per block: invariants and everything that depends on them only")
        ca_done = set()
        for n in p.uniform:
            if n.kind in ("const", "hold"):
                continue
            if n.kind == "inv":
                if n.name in p.cells and p.cells[n.name].i not in ca_done:      # a cell @sample only reads
                    ca_done.add(p.cells[n.name].i)
                    L.append(f"    const int64_t ca{p.cells[n.name].i} = (int64_t){ref(p.cells[n.name])};")
                L.append(f"    const double u{n.i} = {self.inv_src(n.name)};   // {n.name}")
            else:
                L.append(f"    const double u{n.i} = ZT_UNI({_expr(n.op, [ref(x) for x in n.args])});")
        if self.ulds:
            L.append("    __syncthreads();")
            L.append("    if (lane == 0) {")
            for n in p.uniform:
                if n.kind not in ("const", "hold"):
                    L.append(f"      zt_u[{self.uslot[n.i]}] = u{n.i};")
            L.append("    }")
            L.append("    __syncthreads();")
        for cn in sorted({g_[0].i: g_[0] for g_ in p.node_guard.values()}.values(), key=lambda n: n.i):
            L.append(f"    const bool zg{cn.i} = za_truthy({ref(cn)});")
        for gn in p.guards:
            L.append(f"    if (za_truthy({ref(gn)})) {{   // a rare-event branch the lowering left out (tpar.split_guards) is due: this frame runs with the section code")
            L.append("      zt_evf = pos;")
            self.emit_serial_frame("      ")
            L.append("    }")
        if self.has_mem:
            L.append("    int64_t zt_high = b.mem_high[inst], zt_hc = 0;")
            for a in self.cell_addrs:
                if a.i not in ca_done:
                    L.append(f"    const int64_t ca{a.i} = (int64_t){ref(a)};")
            if self.cell_addrs:
                ca = self.cell_addrs
                clash = " || ".join([f"ca{a.i} >= mcap" for a in ca] + [f"ca{a.i} == ca{b_.i}" for i_, a in enumerate(ca) for b_ in ca[i_ + 1:]])
                L.append(f"    int64_t cmin = ca{ca[0].i}, cmax = ca{ca[0].i};")
                for a in ca[1:]:
                    L.append(f"    cmin = ca{a.i} < cmin ? ca{a.i} : cmin; cmax = ca{a.i} > cmax ? ca{a.i} : cmax;")
                L.append(f"    if ({clash}) {{   // cells that alias each other (or lie past the arena): not a case for this kernel")
                self.emit_leave("      ", "pos")
                L.append("    }")
        for Lp in self.lcell_loops:
            self.emit_address_pass(Lp)
        self.has_lbox = bool(p.stores) and any(self.pass_keys(Lp) for Lp in self.lcell_loops)
        self.has_wbox = bool(p.loads) and any(k in Lp.cell_out for Lp in self.lcell_loops for k in self.pass_keys(Lp))
        if self.has_wbox:
            # ... and a read at a moving address (a delay line's, a gather's) must stay clear of the cells some loop STORES to: their
            # values live in registers / LDS while a block runs, memory has what they were when it began
            L.append("    int64_t wmin = INT64_MAX, wmax = -1;     // bounding box of the per-trip cells that loops store to")
            for Lp in self.lcell_loops:
                keys = self.pass_keys(Lp)
                nk, so = len(keys), self.site_off[Lp.id]
                for j, k in enumerate(keys):
                    if k in Lp.cell_out:
                        L.append(f"    {{ const int64_t lo = zt_site[{so + 3 * nk + j}], hi = zt_site[{so + 4 * nk + j}]; if (hi >= lo) {{ wmin = lo < wmin ? lo : wmin; wmax = hi > wmax ? hi : wmax; }} }}")
        if self.has_lbox:
            # per-trip cells are read as they stand when the block begins (or live in LDS for its length): a delay-line write
            # that lands among them would have to be seen by the loop of the very next frame -- not a case for this kernel
            L.append("    int64_t lmin = INT64_MAX, lmax = -1;     // bounding box of the loops' per-trip cells")
            for Lp in self.lcell_loops:
                nk, so = len(self.pass_keys(Lp)), self.site_off[Lp.id]
                for j in range(nk):
                    L.append(f"    {{ const int64_t lo = zt_site[{so + 3 * nk + j}], hi = zt_site[{so + 4 * nk + j}]; if (hi >= lo) {{ lmin = lo < lmin ? lo : lmin; lmax = hi > lmax ? hi : lmax; }} }}")
        if self.cell_loops:
            L.append("    {   // per-trip cells into LDS for the length of the block (a loop whose cells do not fit keeps them in memory)")
            L.append("      int zoff = 0;")
            for Lp in self.cell_loops:
                nk = len(self.pass_keys(Lp))
                L.append(f"      zlo{Lp.id} = zoff; zlds{Lp.id} = zln{Lp.id} > 0 && zoff + {nk} * zln{Lp.id} <= ZT_CELL_DOUBLES; if (zlds{Lp.id}) zoff += {nk} * zln{Lp.id};")
                L.append(f"      if (zlds{Lp.id}) {{")
                so = self.site_off[Lp.id]
                L.append(f"        for (int q = lane; q < {nk} * zln{Lp.id}; q += 64) {{ const int j = q / zln{Lp.id}, k = q - j * zln{Lp.id}; const int64_t A = zt_site[{so} + j] + (int64_t)k * zt_site[{so + nk} + j];")
                L.append(f"          zt_cells[zlo{Lp.id} + q] = A < mcap ? memp[A * mse] : 0.0; }}")
                L.append("      }")
                if self.cell_undo:
                    L.append(f"      else if (zln{Lp.id} > 0) {{   // (cells that do not fit would be stored to memory trip by trip: no way back from that)")
                    self.emit_leave("        ", "pos")
                    L.append("      }")
            if self.cell_undo:
                L.append("      zt_cn = zoff;")
            L.append("      __syncthreads();")
            L.append("    }")
        if self.inv_mats:
            L.append("    __syncthreads();")
            for k, key in enumerate(self.inv_mats):
                L.append(f"    {{ const ZtMat2 am = {{{ref(key[0])}, {ref(key[1])}, {ref(key[2])}, {ref(key[3])}}}; zt_mat_table(zt_m + {k} * ZT_MAT_TABLE_DOUBLES, am, lane); }}")
            if not self.inv_coefs:
                L.append("    __syncthreads();")
        if self.inv_coefs:
            if not self.inv_mats:
                L.append("    __syncthreads();")
            for k, a in enumerate(self.inv_coefs):
                L.append(f"    zt_w[{k} * 64 + lane] = zt_pow_row({ref(a)}, lane);")
                L.append(f"    if (lane == 0) {{ const double p2 = {ref(a)} * {ref(a)}, p4 = p2 * p2, p8 = p4 * p4; zt_q[{k} * 4] = p2; zt_q[{k} * 4 + 1] = p4; zt_q[{k} * 4 + 2] = p8; zt_q[{k} * 4 + 3] = p8 * p8; }}")
            L.append("    __syncthreads();")
        L.append("    // state carried from frame to frame (wave-uniform)")
        for name, c in self.cname.items():
            L.append(f"    double {c} = {self.inv_src(name)};   // {name}")
        for name, h in self.hname.items():
            L.append(f"    double {h} = {self.inv_src(name)};   // {name}: its last written value (frames may leave it alone)")

    def pass_inputs(self, Lp: LoopInfo) -> int:
        return len(self.pass_analysis(Lp)[3])

    def pass_analysis(self, Lp: LoopInfo):
        """(need, uphis, rloads, outside inputs) of a loop's address pass."""
        p = self.plan
        keys = self.pass_keys(Lp)
        rloads = [x for grp in p.rings.get(Lp.id, []) for x in grp.loads]
        need: Dict[int, N] = {}
        todo = [Lp.cells[k] for k in keys] + ([Lp.cond] if Lp.cond is not None else []) + [u for _, u, _ in rloads] + list(Lp.guards)
        uphis = []
        while todo:
            n = todo.pop()
            if n.i in need or not _in_subtree(n, Lp):
                continue
            need[n.i] = n
            if n.kind == "phi":
                if n.name not in uphis:
                    uphis.append(n.name)
                todo.append(Lp.next[n.name])
            if n.kind == "lcin":
                todo.append(Lp.cells[n.name])
            todo.extend(n.args)
        ext: Dict[int, N] = {}
        for n in list(need.values()) + [Lp.init[v] for v in uphis] + ([Lp.count] if Lp.count is not None else []) + list(self.cell_addrs):
            for x in ((n,) if not _in_subtree(n, Lp) else n.args):
                if not _in_subtree(x, Lp) and x.kind not in ("const", "hold"):
                    ext[x.i] = x
        return need, uphis, rloads, [ext[i] for i in sorted(ext)]

    def emit_address_pass(self, Lp: LoopInfo):
        """Before a block's first chunk: walk the trips of a loop with per-trip cells once, addresses only. Every address expression
        must step evenly through the trips (a[k] = a[0] + k * stride) inside the arena, and two expressions may never name one cell
        (zt_sites_ok: disjoint ranges, or interleaved records -- same stride, offsets that differ by less than a multiple of it)."""
        p, L, ref = self.plan, self.L, self.ref
        reg = p.regions[Lp.id]
        keys = self.pass_keys(Lp)
        need, uphis, rloads, exts = self.pass_analysis(Lp)
        memo = (not any(n.kind == "lcin" for n in need.values()) and all(x.uniform for x in exts)
                and not os.environ.get("ZA_TPAR_NO_MEMO"))
        self.pass_memo[Lp.id] = (memo, exts)
        L.append(f"    {{   // per-trip cells of loop {Lp.id}: addresses step evenly through the trips and never meet; offsets of its ring reads")
        xs = self.pass_memo[Lp.id][1]
        mo = self.memo_off[Lp.id]
        so = self.site_off.get(Lp.id, 0)
        nk = len(keys)
        if memo:
            L.append(f"      bool zsame = zpv{Lp.id};")
            for k, x in enumerate(xs):
                L.append(f"      zsame &= __builtin_bit_cast(unsigned long long, {ref(x)}) == zt_memo[{mo + k}];")
            L.append("      if (!zsame) {")
            L.append("      __syncthreads();")
            for k, x in enumerate(xs):
                L.append(f"      zt_memo[{mo + k}] = __builtin_bit_cast(unsigned long long, {ref(x)});")
            L.append(f"      zpv{Lp.id} = true; zph{Lp.id} = 0;")
        else:
            L.append(f"      zph{Lp.id} = 0;")
            L.append("      {")
        if Lp.id in p.rings:
            L.append(f"      zrok{Lp.id} = true;")
            for ld, _, _ in rloads:
                L.append(f"      zro_lo{ld.i} = 2147483647; zro_hi{ld.i} = -2147483647;")
        L.append("      bool zt_abad = false;")
        if Lp.guards:
            L.append(f"      zpg{Lp.id} = false;")
        for v in uphis:
            L.append(f"      double {self.phi_name[Lp.phis[v].i]} = {ref(Lp.init[v])};")
        # per address expression j: first address, stride, previous, lowest, highest -- in LDS (zt_site), every lane the same values
        if nk:
            L.append(f"      long long* const zs = zt_site + {so};      // [5][{nk}]: a0, stride, previous, lo, hi")
            L.append(f"      for (int j = lane; j < {nk}; j += 64) {{ zs[j] = 0; zs[{nk} + j] = 1; zs[{2 * nk} + j] = 0; zs[{3 * nk} + j] = 0; zs[{4 * nk} + j] = -1; }}")
            L.append("      __syncthreads();")
        L.append("      int64_t zkn = 0;")
        if Lp.count is not None:
            L.append(f"      const int64_t zt_cnt = za_loopcount(ZT_UNI({ref(Lp.count)}));")
            L.append("      for (int64_t zk = 0; zk < zt_cnt; ++zk, ++zkn) {")
        else:
            L.append("      for (int64_t zk = 0; zk < ZA_LOOP_CAP; ++zk, ++zkn) {")
        for i in sorted(need):
            n = need[i]
            if n.kind == "phi":
                continue
            if n.kind == "lcin":
                an = ref(Lp.cells[n.name])
                L.append(f"        const double n{n.i} = ZT_UNI((int64_t){an} < mcap ? memp[(int64_t){an} * mse] : 0.0);")
                continue
            L.append(f"        const double n{n.i} = ZT_UNI({_expr(n.op, [ref(x) for x in n.args])});")
            if n is Lp.cond:
                L.append(f"        if (!za_truthy(n{n.i})) break;")
        if Lp.cond is not None and Lp.cond.i not in need:
            L.append(f"        if (!za_truthy({ref(Lp.cond)})) break;")
        for gc in Lp.guards:
            L.append(f"        zpg{Lp.id} |= za_truthy({ref(gc)});     // a statement of this trip that runs as an event is due")
        for ld, u, sign in rloads:
            L.append(f"        {{ const double o = {'' if sign > 0 else '-'}{ref(u)}; const int oi = (int)o; zrok{Lp.id} &= (double)oi == o && fabs(o) < 1.0e9;")
            L.append(f"          zro_lo{ld.i} = oi < zro_lo{ld.i} ? oi : zro_lo{ld.i}; zro_hi{ld.i} = oi > zro_hi{ld.i} ? oi : zro_hi{ld.i}; }}")
        if nk:
            # lane j follows address expression j (the addresses of a trip are wave-uniform values: every lane has them all)
            L.append("        {")
            L.append("          long long A = 0;")
            for j, k in enumerate(keys):
                L.append(f"          A = lane == {j} ? (long long)(int){ref(Lp.cells[k])} : A;")
            stored = sum(1 << j for j, k in enumerate(keys) if k in Lp.cell_out)
            L.append(f"          if (lane < {nk}) {{")
            L.append(f"            zt_abad |= ((0x{stored:x}ull >> lane) & 1ull) && A >= mcap;")
            L.append(f"            if (zk == 0) {{ zs[lane] = A; zs[{3 * nk} + lane] = A; zs[{4 * nk} + lane] = A; }}")
            L.append(f"            else {{ if (zk == 1) zs[{nk} + lane] = A - zs[{2 * nk} + lane]; else zt_abad |= (A - zs[{2 * nk} + lane]) != zs[{nk} + lane];")
            L.append(f"              zs[{3 * nk} + lane] = A < zs[{3 * nk} + lane] ? A : zs[{3 * nk} + lane]; zs[{4 * nk} + lane] = A > zs[{4 * nk} + lane] ? A : zs[{4 * nk} + lane]; }}")
            L.append(f"            zs[{2 * nk} + lane] = A;")
            L.append("          }")
            L.append("        }")
        for v in uphis:
            L.append(f"        const double q{self.phi_name[Lp.phis[v].i]} = {ref(Lp.next[v])};")
        for v in uphis:
            L.append(f"        {self.phi_name[Lp.phis[v].i]} = q{self.phi_name[Lp.phis[v].i]};")
        L.append("      }")
        if nk:
            L.append("      __syncthreads();")
            stored = sum(1 << j for j, k in enumerate(keys) if k in Lp.cell_out)
            always = sum(1 << j for j, k in enumerate(keys) if k in Lp.cell_out and not (k in Lp.cell_flag and Lp.cell_flag[k].kind != "const"))
            L.append(f"      zt_abad = __ballot(zt_abad) != 0ull;")
            L.append(f"      zt_abad |= !zt_sites_all_ok(zs, {nk}, 0x{stored:x}ull, lane);")
            for a in self.cell_addrs:
                L.append(f"      zt_abad |= __ballot(lane < {nk} && ca{a.i} >= zs[{3 * nk} + lane] && ca{a.i} <= zs[{4 * nk} + lane]) != 0ull;")
            L.append(f"      {{ const long long h = (lane < {nk} && ((0x{always:x}ull >> lane) & 1ull) && zs[{4 * nk} + lane] >= 0) ? zs[{4 * nk} + lane] + 1 : 0;")
            L.append(f"        zph{Lp.id} = zt_wave_max_i64(h); }}     // (cells stored to in every frame)")
        if Lp in getattr(self, "cell_loops", []):
            L.append(f"      zln{Lp.id} = (int)zkn;")
        L.append("      if (zt_abad) {")
        self.emit_leave("        ", "pos")
        L.append("      }")
        L.append("      }")
        L.append(f"      zt_high = zph{Lp.id} > zt_high ? zph{Lp.id} : zt_high;")
        L.append("    }")
        if Lp.guards:
            L.append(f"    if (zpg{Lp.id}) {{   // ... in the segment's first frame: that frame runs with the section code")
            L.append("      zt_evf = pos;")
            self.emit_serial_frame("      ")
            L.append("    }")

    def pass_keys(self, Lp: LoopInfo) -> List[str]:
        return [k for k in Lp.cells if (k in Lp.cin and Lp.cin[k].i in self.live_ids()) or k in Lp.cell_out]

    def cell_ld(self, Lp: LoopInfo, key: str, A: str) -> str:
        """A per-trip cell's value before the chunk: from LDS when the block staged the loop's cells, else from the arena."""
        mem = f"({A} < mcap ? memp[{A} * mse] : 0.0)"
        if (Lp.id, key) not in self.cell_slot:
            return mem
        j = self.cell_slot[(Lp.id, key)]
        return f"(zlds{Lp.id} ? zt_cells[zlo{Lp.id} + {j} * zln{Lp.id} + (int)zk{Lp.id}] : {mem})"

    def live_ids(self):
        if not hasattr(self, "_live"):
            self._live = {n.i for r in [self.plan.top] + list(self.plan.regions.values()) for n in r.nodes}
        return self._live

    # -- one block's chunks ----------------------------------------------------------------------------------------------------
    def emit_chunk_loop(self):
        p, L, ref = self.plan, self.L, self.ref
        cname = self.cname
        L.append("    for (int64_t f0 = pos; f0 < bend; f0 += 64) {")
        q = "" if self.has_cut else "const "
        L.append(f"    {q}int tn = (int)(bend - f0 < 64 ? bend - f0 : 64);")
        L.append(f"    {q}int last = tn - 1;")
        L.append(f"    {q}bool valid = lane < tn;")
        for n in p.inputs:
            L.append(f"    const double n{n.i} = (double)x{n.i};")
        L.append("    {   // the next chunk's audio (of the next block, at a block's end)")
        L.append("      const int64_t nf = f0 + 64 < bend ? f0 + 64 : bend;")
        L.append("      if (nf + lane < frames) {")
        for n in p.inputs:
            L.append(f"        x{n.i} = in_[{int(n.val)} * a.frame_stride + nf + lane];")
        L.append("      } else {")
        for n in p.inputs:
            L.append(f"        x{n.i} = 0.0f;")
        L.append("      }")
        L.append("    }")
        # Values leave the registers as early as possible: a state's carry is taken (v_readlane at the chunk's last frame) as soon
        # as both its recurrence and its new value exist, and the values a block must leave in vars[] -- needed in the block's
        # last chunk only -- are stored in small conditional batches right after they are computed, instead of all living to the
        # end of the chunk body (144 written variables would be 288 registers per lane there).
        self.in_loop = True
        if self.inv_coefs or self.inv_mats or self.ulds:
            L.append("    int zo; asm volatile(\"s_mov_b32 %0, 0\" : \"=s\"(zo));   // opaque 0: keeps the table reads inside the iteration")
        if self.has_abort:
            L.append("    if (lane == 0) {   // the states as they stand before this chunk, in case it has to be handed to the serial code")
            for k, (name, c) in enumerate(cname.items()):
                L.append(f"      zt_snap[{k}] = {c};")
            L.append("    }")
            L.append("    bool zt_bad = false, zt_badl = false;")
            if self.cell_undo:
                L.append("    for (int q = lane; q < zt_cn; q += 64) zt_cells[ZT_CELL_DOUBLES + q] = zt_cells[q];     // the cells as this chunk finds them")
                L.append("    __syncthreads();")
        if self.has_events and p.event_exposed:
            L.append(f"    bool fin = true;   // every chunk leaves what it wrote in memory: an event's body reads {', '.join(p.event_exposed[:4])} from the frame before")
        elif self.has_cut:
            L.append("    bool fin = f0 + 64 >= bend;   // the block's last chunk -- or the one an event cuts short (set at the cut)")
        else:
            L.append("    const bool fin = f0 + 64 >= bend;   // the block's last chunk: its last frame leaves every written variable as the script would")
        self.avail = {n.i for n in p.inputs}
        self.raw_issued: set = set()
        self.unit_done: set = set()
        self.carried: set = set()
        self.stored: set = set()
        self.finals = [(name, o) for name, o in p.outs.items() if name != RNG_INDEX and name not in self.hname]
        self.finals += [(f"spl{ch}", p.spl_out[ch]) for ch in range(p.nch) if f"spl{ch}" not in p.outs]
        self.pending: List[tuple] = []
        self.before_cut = self.has_cut
        self.emit_region(p.top, "    ")
        L.append("    ZT_STAMP(2)")
        if self.has_abort:
            self.emit_abort_block()
        if self.has_streams:
            L.append("    // the chunk's writes land after all of its reads are resolved")
            for st_ in p.stores:
                j = st_.j
                if st_.mode == "late":
                    gate = f"valid && zsu{j}" if st_.pred is not None else "valid"
                    L.append(f"    if ({gate}) memp[(int64_t){ref(st_.addr)} * mse] = {ref(st_.value)};")
                if st_.mode in ("late", "early"):
                    upd = (f"{{ const int64_t h0 = zq0_{j} + zqk_{j}, h1 = zqk_{j} < tn ? zq1_{j} + (tn - zqk_{j}) : 0; zt_high = h0 > zt_high ? h0 : zt_high; "
                           f"zt_high = h1 > zt_high ? h1 : zt_high; }}")
                    L.append(f"    if (zsu{j}) {upd}" if st_.pred is not None else f"    {upd}")
                else:
                    L.append(f"    for (uint64_t m = zsm{j}; m; m &= m - 1) {{   // in frame order: a later frame's store to the same cell wins")
                    L.append(f"      const int l = (int)__ffsll((long long)m) - 1;")
                    L.append(f"      if (lane == l) memp[(int64_t){ref(st_.addr)} * mse] = {ref(st_.value)};")
                    L.append(f"      const int64_t h = (int64_t)zt_readlane({ref(st_.addr)}, l) + 1; zt_high = h > zt_high ? h : zt_high;")
                    L.append("    }")
        # variables that frames may leave alone: the last value written in this chunk, if any
        for name, h in self.hname.items():
            o = p.outs[name]
            L.append(f"    {{ const double v = {ref(o)}; const uint64_t m = __ballot(valid && !zt_is_hold(v)); if (m) {h} = zt_readlane(v, 63 - __clzll((long long)m)); }}")
        self.retire(final=True)
        if self.hname:
            L.append("    if (fin && lane == last) {")
            for name, h in self.hname.items():
                L.append(f"      {self.dst(name)} = {h};")
            L.append("    }")
        if self.has_mem:
            L.append("    if (fin && lane == last) b.mem_high[inst] = zt_high > zt_hc ? zt_high : zt_hc;")
        if self.pin:
            L.append(f"    asm volatile(\"\" : {self.pin});   // the next chunk's audio has landed; its wait comes before this chunk's stores")
        L.append("    if (valid) {")
        for ch in range(p.nch):
            L.append(f"      out_[{ch} * a.frame_stride + f0 + lane] = (float){ref(p.spl_out[ch])};")
        L.append("    }")
        if p.uses_rand:
            L.append(f"    zt_mt_retire(zt_mt, zt_pos0, (int){cname[RNG_INDEX]}, lane);")
        L.append("    ZT_STAMP(5)")
        L.append("    }")
        self.in_loop = False
        if p.uses_rand:
            L.append(f"    zt_mt_end(zt_mt, zt_pos0, (int){cname[RNG_INDEX]}, zt_gmt, b.mt_se, b.mti + inst, lane);")
        if self.cell_loops:
            L.append("    __syncthreads();")
            self.emit_cells_writeback("    ")
        if p.has_block and (self.has_mem or True):
            L.append("    __builtin_amdgcn_fence(__ATOMIC_RELEASE, \"workgroup\");     // the block's values are in vars[] / mem[] before @block reads them")
            L.append("    __builtin_amdgcn_wave_barrier();")
            L.append("    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, \"workgroup\");")

    def emit_cells_writeback(self, ind: str):
        L = self.L
        for Lp in self.cell_loops:
            stored = [(self.cell_slot[(Lp.id, k)], k) for k in self.pass_keys(Lp) if k in Lp.cell_out]
            L.append(f"{ind}if (zlds{Lp.id}) {{   // the block's cells go back to the arena")
            for j, k in stored:
                so, nk = self.site_off[Lp.id], len(self.pass_keys(Lp))
                L.append(f"{ind}  for (int k = lane; k < zln{Lp.id}; k += 64) memp[(zt_site[{so + j}] + (int64_t)k * zt_site[{so + nk + j}]) * mse] = zt_cells[zlo{Lp.id} + {j} * zln{Lp.id} + k];")
            L.append(f"{ind}}}")

    def ready(self, o: N) -> bool:
        return (o.uniform and o.loop is None) or o.kind in ("const", "hold") or o.i in self.avail

    def retire(self, final: bool = False):
        p, L, ref = self.plan, self.L, self.ref
        for name, c in self.cname.items():
            o = p.outs[name]
            if name not in self.carried and name in self.unit_done and self.ready(o):
                self.carried.add(name)
                L.append(f"    {c} = {ref(o) if (o.uniform or o.kind == 'const') else f'zt_readlane({ref(o)}, last)'};")
        for name, o in self.finals:
            if (name in p.cells or name.startswith("memw@")) and not final:
                continue                                   # (a cell needs its "stored to" flag beside it: both go out at the end)
            if name not in self.stored and self.ready(o) and (final or not (o.uniform or o.kind == "const")):
                self.stored.add(name)
                self.pending.append((name, o))
        if self.pending and (final or len(self.pending) >= 12):
            L.append("    if (fin && lane == last) {")
            for name, o in self.pending:
                if name.startswith("memw@"):
                    continue
                if name in p.cells:                     # a cell is written back only if the launch stored to it at all
                    flag = p.outs.get("memw@" + name[4:])
                    if flag is not None:
                        L.append(f"      if ({ref(flag)} != 0.0) {{ {self.dst(name)} = {ref(o)}; zt_hc = zt_hc > ca{p.cells[name].i} + 1 ? zt_hc : ca{p.cells[name].i} + 1; }}")
                    continue
                L.append(f"      {self.dst(name)} = {ref(o)};")
            L.append("    }")
            self.pending.clear()

    def emit_flush_states(self, ind: str):
        """The carried states and last-written values as they stand in registers, into vars[] / mem[] (the section code is about
        to run on them). Variables a frame writes before it reads them are not among them: the section code writes them itself."""
        p, L = self.plan, self.L
        cname = self.cname
        L.append(f"{ind}if (lane == 0) {{")
        for name, c in cname.items():
            if name.startswith("memw@") or name == RNG_INDEX:
                continue
            if name in p.cells:
                flag = "memw@" + name[4:]
                if flag in cname:
                    L.append(f"{ind}  if ({cname[flag]} != 0.0) {{ {self.dst(name)} = {c}; zt_hc = zt_hc > ca{p.cells[name].i} + 1 ? zt_hc : ca{p.cells[name].i} + 1; }}")
                continue
            L.append(f"{ind}  {self.dst(name)} = {c};")
        for name, h in self.hname.items():
            L.append(f"{ind}  {self.dst(name)} = {h};")
        if self.has_mem:
            L.append(f"{ind}  b.mem_high[inst] = zt_high > zt_hc ? zt_high : zt_hc;")
        L.append(f"{ind}}}")

    def emit_abort_block(self):
        p, L = self.plan, self.L
        cname = self.cname
        L.append("    if (zt_bad || __ballot(valid && zt_badl)) {")
        L.append("      // a condition of the lowering does not hold in this chunk: put the states back as they were before it and leave")
        L.append("      // the rest of the launch to the serial code (za_launch_fast runs it right behind this kernel)")
        for st_ in reversed(self.early):
            L.append(f"      if (valid && zse{st_.j}) memp[(int64_t){self.ref(st_.addr)} * mse] = zso{st_.j};      // (what this chunk's early stores replaced)")
        if self.cell_undo:
            L.append("      __syncthreads();")
            L.append("      for (int q = lane; q < zt_cn; q += 64) zt_cells[q] = zt_cells[ZT_CELL_DOUBLES + q];")
            L.append("      __syncthreads();")
            self.emit_cells_writeback("      ")
        L.append("      if (lane == 0) {")
        for k, name in enumerate(cname):
            if name.startswith("memw@") or name == RNG_INDEX:
                continue
            if name in p.cells:
                flag = "memw@" + name[4:]
                if flag in cname:
                    fk = list(cname).index(flag)
                    L.append(f"        if (zt_snap[{fk}] != 0.0) {{ {self.dst(name)} = zt_snap[{k}]; zt_high = zt_high > ca{p.cells[name].i} + 1 ? zt_high : ca{p.cells[name].i} + 1; }}")
                continue
            L.append(f"        {self.dst(name)} = zt_snap[{k}];")
        for name, h in self.hname.items():
            L.append(f"        {self.dst(name)} = {h};")
        if self.has_mem:
            L.append("        b.mem_high[inst] = zt_high;")
        L.append("      }")
        if p.uses_rand:
            k = list(cname).index(RNG_INDEX)
            L.append(f"      zt_mt_end(zt_mt, zt_pos0, (int)zt_snap[{k}], zt_gmt, b.mt_se, b.mti + inst, lane);")
        self.emit_leave("      ", "f0")
        L.append("    }")

    # -- one region's schedule ----------------------------------------------------------------------------------------------------
    def serial_loop(self, reg: Region, comps: List[Component], ind: str):
        """64 uniform steps; leaves the state before each frame in k<st> of that frame's lane."""
        L, ref = self.L, self.ref
        for c in comps:
            for nm in c.names:
                s = reg.st[nm].i
                L.append(f"{ind}double y{s} = {self.carry(reg, nm)}, k{s} = {self.carry(reg, nm)};")
        L.append(f"{ind}for (int t = 0; t < tn; ++t) {{")
        L.append(f"{ind}  const bool me = lane == t;")
        seen_ext = set()
        for c in comps:
            mem = {m.i for m in c.members}
            for nm in c.names:
                s = reg.st[nm].i
                L.append(f"{ind}  k{s} = me ? y{s} : k{s};")
            for x in c.ext:
                if not x.uniform and x.kind not in ("const", "hold") and x.i not in seen_ext:
                    seen_ext.add(x.i)
                    L.append(f"{ind}  const double e{x.i} = zt_readlane({ref(x)}, t);")

            def sref(x: N, mem=mem) -> str:
                if x.kind in ("st", "lcin") and x.i in mem:
                    return f"y{x.i}"
                if x.i in mem:
                    return f"m{x.i}"
                if x.kind in ("const", "hold") or x.uniform:
                    return ref(x)
                return f"e{x.i}"

            for m in c.members:
                if m.kind in ("st", "lcin"):
                    continue
                L.append(f"{ind}  const double m{m.i} = {_expr(m.op, [sref(x) for x in m.args])};")
            for nm in c.names:            # all new states are computed from the old ones before any is replaced
                L.append(f"{ind}  const double q{reg.st[nm].i} = {sref(reg.outs[nm])};")
            for nm in c.names:
                L.append(f"{ind}  y{reg.st[nm].i} = q{reg.st[nm].i};")
        L.append(f"{ind}}}")

    def emit_site(self, st_: StoreSite, ind: str):
        p, L, ref = self.plan, self.L, self.ref
        j, an = st_.j, ref(st_.addr)
        if st_.mode == "sparse":
            L.append(f"{ind}// conditional write {j} into a buffer @sample never reads: the frames whose condition holds, inside the arena, away")
            L.append(f"{ind}// from every cell and from this chunk's other writes")
            L.append(f"{ind}const uint64_t zsm{j} = __ballot(valid && za_truthy({ref(st_.pred)}));")
            L.append(f"{ind}int64_t zlo{j} = 0, zhi{j} = -1;")
            L.append(f"{ind}if (zsm{j}) {{")
            L.append(f"{ind}  const bool on = (zsm{j} >> lane) & 1ull;")
            L.append(f"{ind}  const int64_t A = (int64_t){an};")
            L.append(f"{ind}  zlo{j} = zt_wave_min_i64(on ? A : INT64_MAX); zhi{j} = zt_wave_max_i64(on ? A : INT64_MIN);")
            L.append(f"{ind}  zt_bad |= zhi{j} >= mcap;")
            if self.cell_addrs:
                L.append(f"{ind}  zt_bad |= zlo{j} <= cmax && zhi{j} >= cmin;")
            if self.has_lbox:
                L.append(f"{ind}  zt_bad |= zlo{j} <= lmax && zhi{j} >= lmin;")
            L.append(f"{ind}}}")
            return
        self.emit_site_span(st_, ind, "")
        j = st_.j
        cond = f"__popcll(zqm_{j}) > 1 || zq0_{j} + zqk_{j} > mcap || (zqk_{j} < tn && zq1_{j} + (tn - zqk_{j}) > mcap)"
        if self.cell_addrs:
            cond += f" || (zq0_{j} <= cmax && zq0_{j} + zqk_{j} > cmin) || (zqk_{j} < tn && zq1_{j} <= cmax && zq1_{j} + (tn - zqk_{j}) > cmin)"
        if self.has_lbox:
            cond += f" || (zq0_{j} <= lmax && zq0_{j} + zqk_{j} > lmin) || (zqk_{j} < tn && zq1_{j} <= lmax && zq1_{j} + (tn - zqk_{j}) > lmin)"
        L.append(f"{ind}const bool zsb{j} = {cond};")
        L.append(f"{ind}zt_bad |= {'zsu%d && ' % j if st_.pred is not None else ''}zsb{j};")

    def emit_site_span(self, st_: StoreSite, ind: str, sfx: str):
        """Where a delay-line write goes in this chunk: s0 + [0, sk) and, behind at most one jump (a ring's wrap), s1 + [0, tn - sk)."""
        L, ref = self.L, self.ref
        j, an = f"{st_.j}{sfx}", ref(st_.addr)
        L.append(f"{ind}// delay-line write {st_.j}: must advance by one cell per frame (at most one wrap inside the chunk)")
        if st_.pred is not None:
            L.append(f"{ind}const bool zsu{j} = za_truthy({ref(st_.pred)});      // (block-constant condition)")
        L.append(f"{ind}const double zqp_{j} = zt_shift1({an}, {an} - 1.0);")
        L.append(f"{ind}const uint64_t zqm_{j} = __ballot(valid && lane > 0 && ({an} - zqp_{j} != 1.0));")
        L.append(f"{ind}const int zqk_{j} = zqm_{j} ? (int)__ffsll((long long)zqm_{j}) - 1 : tn;")
        L.append(f"{ind}const int64_t zq0_{j} = (int64_t)zt_readlane({an}, 0), zq1_{j} = zqk_{j} < tn ? (int64_t)zt_readlane({an}, zqk_{j}) : 0;")

    def emit_site_pairs(self, ind: str):
        """No two writes of a chunk may touch one cell (different buffers are an assumption: checked here)."""
        p, L = self.plan, self.L
        dense = [s for s in p.stores if s.mode != "sparse"]
        for x, sa in enumerate(dense):
            for sb in dense[x + 1:]:
                a, b = sa.j, sb.j
                both = " && ".join(f"zsu{x_.j}" for x_ in (sa, sb) if x_.pred is not None)
                meet = f"zt_spans_meet(zq0_{a}, zqk_{a}, zq1_{a}, tn - zqk_{a}, zq0_{b}, zqk_{b}, zq1_{b}, tn - zqk_{b})"
                if sa.region == sb.region and sa.mode == "late" and sb.mode == "late":
                    # writes into ONE delay line: fine while they move in step -- the same cell in the same frame, where program
                    # order decides (the late stores go out in program order; a read takes the last write in front of it)
                    meet += f" && !(zq0_{a} == zq0_{b} && zqk_{a} == zqk_{b} && zq1_{a} == zq1_{b})"
                L.append(f"{ind}zt_bad |= {both + ' && ' if both else ''}{meet};")
            for sp_ in (s for s in p.stores if s.mode == "sparse"):
                a, b = sa.j, sp_.j
                L.append(f"{ind}zt_bad |= zhi{b} >= zlo{b} && (zt_span_hits(zq0_{a}, zqk_{a}, zlo{b}, zhi{b}) || zt_span_hits(zq1_{a}, tn - zqk_{a}, zlo{b}, zhi{b}));")
        sparse = [s for s in p.stores if s.mode == "sparse"]
        for x, sa in enumerate(sparse):
            for sb in sparse[x + 1:]:
                L.append(f"{ind}zt_bad |= zhi{sa.j} >= zlo{sa.j} && zhi{sb.j} >= zlo{sb.j} && zlo{sa.j} <= zhi{sb.j} && zlo{sb.j} <= zhi{sa.j};")
        for st_ in self.early:
            j = st_.j
            gate = f"zsu{j} && !zsb{j}" if st_.pred is not None else f"!zsb{j}"
            L.append(f"{ind}// write {j} goes out now: the loops that gather from its buffer read memory (what it replaces is kept for a hand-back)")
            L.append(f"{ind}const bool zse{j} = {gate} && !zt_bad;")
            L.append(f"{ind}double zso{j} = 0.0;")
            L.append(f"{ind}if (valid && zse{j}) {{ zso{j} = memp[(int64_t){self.ref(st_.addr)} * mse]; memp[(int64_t){self.ref(st_.addr)} * mse] = {self.ref(st_.value)}; }}")

    def emit_load(self, n: N, ind: str):
        p, L, ref = self.plan, self.L, self.ref
        if n.i in self.ring_lds:
            L.append(f"{ind}const double n{n.i} = zt_ring[{self.ring_lds[n.i].replace('{U%d}' % n.i, ref(self.ring_u[n.i]))}];     // (staged: emit_ring_stage)")
            return
        L.append(f"{ind}double n{n.i};   // delay-line read: memory as it was before this chunk, or the value an earlier frame of the chunk writes")
        L.append(f"{ind}{{")
        if n.i in self.raw_issued:
            L.append(f"{ind}  const int64_t B = B{n.i};")
            L.append(f"{ind}  double v = raw{n.i};")
        else:
            L.append(f"{ind}  const int64_t B = (int64_t){ref(n.args[0])};")
            L.append(f"{ind}  double v = B < mcap ? memp[B * mse] : 0.0;")
        self.load_checks(n, "B", ind + "  ", forward=True)
        L.append(f"{ind}  n{n.i} = v;")
        L.append(f"{ind}}}")

    def has_late_site(self, n: N) -> bool:
        return any(s.mode == "late" and ",".join(map(str, s.region)) == n.name for s in self.plan.stores)

    def load_checks(self, n: N, B: str, ind: str, forward: bool):
        """A read at address B against this chunk's writes: other buffers' spans and cells must not be hit; its own buffer's
        late write is forwarded from the writing frame's lane (`v`, only with forward), an early one is already in memory."""
        p, L, ref = self.plan, self.L, self.ref
        if forward and any(s.mode == "late" and ",".join(map(str, s.region)) == n.name and s.j not in n.fb for s in p.stores):
            L.append(f"{ind}int best = -1;")
        for st_ in p.stores:
            j = st_.j
            if st_.mode == "sparse":
                L.append(f"{ind}zt_badl |= {B} >= zlo{j} && {B} <= zhi{j};")
                continue
            on = f"zsu{j} && " if st_.pred is not None else ""
            same = ",".join(map(str, st_.region)) == n.name
            if not same:
                L.append(f"{ind}zt_badl |= {on}((uint64_t)({B} - zq0_{j}) < (uint64_t)zqk_{j} || (uint64_t)({B} - zq1_{j}) < (uint64_t)(tn - zqk_{j}));")
                continue
            L.append(f"{ind}{{ int tw = -1; const int64_t d0 = {B} - zq0_{j}, d1 = {B} - zq1_{j};")
            L.append(f"{ind}  if ((uint64_t)d0 < (uint64_t)zqk_{j}) tw = (int)d0;")
            L.append(f"{ind}  if ((uint64_t)d1 < (uint64_t)(tn - zqk_{j})) tw = zqk_{j} + (int)d1;")
            if st_.mode == "early":
                # memory already holds this chunk's values: right for frames at or before this one, wrong for later ones
                before = "false" if st_.seq < n.val else "true"
                L.append(f"{ind}  zt_badl |= {on}(tw > lane || (tw == lane && {before})); }}")
            elif st_.j in n.fb:
                # (the chunk was cut before the first frame that reads what an earlier one of its frames writes: nothing to see)
                before = "true" if st_.seq < n.val else "false"
                L.append(f"{ind}  zt_badl |= {on}(tw >= 0 && (tw < lane || (tw == lane && {before}))); }}")
            else:
                assert forward
                before = "true" if st_.seq < n.val else "false"
                L.append(f"{ind}  const bool vis = {on}valid && tw >= 0 && (tw < lane || (tw == lane && {before})) && tw >= best;")
                L.append(f"{ind}  if (__ballot(vis)) {{ const double fw = zt_bperm({ref(st_.value)}, tw); v = vis ? fw : v; best = vis ? tw : best; }} }}")
        if self.cell_addrs:
            L.append(f"{ind}zt_badl |= {B} >= cmin && {B} <= cmax;")
        if self.has_wbox:
            L.append(f"{ind}zt_badl |= {B} >= wmin && {B} <= wmax;     // (a per-trip cell some loop stores to: its value lives in LDS)")

    def emit_region(self, reg: Region, ind: str):
        p, L, ref = self.plan, self.L, self.ref
        top = reg.loop is None
        sites_open = False
        run: List[N] = []           # consecutive nodes of the frame that hang on one arm of a block-constant condition

        def flush_run():
            if not run:
                return
            cn, arm = p.node_guard[run[0].i]
            L.append(f"{ind}double " + ", ".join(f"n{n.i} = 0.0" for n in run) + ";")
            L.append(f"{ind}if ({'' if arm else '!'}zg{cn.i}) {{   // only this arm of the block-constant condition needs them")
            for n in run:
                L.append(f"{ind}  n{n.i} = {_expr(n.op, [ref(x) for x in n.args])};")
            L.append(f"{ind}}}")
            run.clear()
            if not self.before_cut:
                self.retire()

        for gid, it in enumerate(reg.items):
            kind = it[0]
            if top and kind == "par" and it[1].i in p.node_guard:
                if run and p.node_guard[run[0].i] != p.node_guard[it[1].i]:
                    flush_run()
                run.append(it[1])
                self.avail.add(it[1].i)
                continue
            flush_run()
            if kind != "site" and sites_open:
                self.emit_site_pairs(ind)
                sites_open = False
            if top:
                if kind == "par":
                    self.avail.add(it[1].i)
                elif kind == "shift":
                    self.avail.add(reg.st[it[1]].i)
                    self.unit_done.add(it[1])
                elif kind in ("scan", "modc"):
                    for nm in it[1].names:
                        self.avail.add(reg.st[nm].i)
                        self.unit_done.add(nm)
                elif kind in ("spec", "serial"):
                    for c in it[1]:
                        for nm in c.names:
                            self.avail.add(reg.st[nm].i)
                            self.unit_done.add(nm)
                elif kind == "loop":
                    for lo in it[1].loop.louts.values():
                        self.avail.add(lo.i)
            if kind == "site":
                self.emit_site(it[1], ind)
                sites_open = True
                continue
            if kind == "cut":
                L.append(f"{ind}{{   // the first frame an event falls on ends the segment: the frames before it are this chunk")
                if p.events:
                    cond = " || ".join(f"za_truthy({ref(e)})" for e in p.events)
                    L.append(f"{ind}  const uint64_t m = __ballot(valid && ({cond}));")
                else:
                    L.append(f"{ind}  const uint64_t m = 0;")
                if self.has_fb:
                    L.append(f"{ind}  // ... and so does the first frame that would read, from a delay line in a feedback loop, what an earlier frame")
                    L.append(f"{ind}  // of this chunk writes: the next segment starts AT that frame")
                    L.append(f"{ind}  bool fbh = false;")
                    done_sites = set()
                    for ld in p.fb_loads:
                        for st_ in p.stores:
                            if st_.j in ld.fb and st_.j not in done_sites:
                                done_sites.add(st_.j)
                                self.emit_site_span(st_, ind + "  ", "c")
                    for ld in p.fb_loads:
                        L.append(f"{ind}  {{ const int64_t B = (int64_t){ref(ld.args[0])};")
                        for st_ in p.stores:
                            if st_.j not in ld.fb:
                                continue
                            j = f"{st_.j}c"
                            on = f"zsu{j} && " if st_.pred is not None else ""
                            before = "true" if st_.seq < ld.val else "false"
                            L.append(f"{ind}    {{ int tw = -1; const int64_t d0 = B - zq0_{j}, d1 = B - zq1_{j};")
                            L.append(f"{ind}      if ((uint64_t)d0 < (uint64_t)zqk_{j}) tw = (int)d0;")
                            L.append(f"{ind}      if ((uint64_t)d1 < (uint64_t)(tn - zqk_{j})) tw = zqk_{j} + (int)d1;")
                            L.append(f"{ind}      fbh |= {on}(tw >= 0 && (tw < lane || (tw == lane && {before}))); }}")
                        L.append(f"{ind}  }}")
                    L.append(f"{ind}  const uint64_t mf = __ballot(valid && fbh);")
                else:
                    L.append(f"{ind}  const uint64_t mf = 0;")
                L.append(f"{ind}  if (m | mf) {{")
                L.append(f"{ind}    const int ee = m ? (int)__ffsll((long long)m) - 1 : 64, ef = mf ? (int)__ffsll((long long)mf) - 1 : 64;")
                L.append(f"{ind}    if (ee <= ef) {{ zt_evf = f0 + ee; bend = zt_evf; tn = ee; }} else {{ zt_fbc = true; bend = f0 + ef; tn = ef; }}")
                L.append(f"{ind}    last = tn - 1; valid = lane < tn; fin = true;")
                L.append(f"{ind}  }}")
                L.append(f"{ind}}}")
                if self.has_fb:
                    L.append(f"{ind}if (zt_fbc && tn < 16) {{   // a feedback delay this short is serial work: the section code takes the rest of the launch")
                    self.emit_flush_states(ind + "  ")
                    self.emit_leave(ind + "  ", "f0")
                    L.append(f"{ind}}}")
                L.append(f"{ind}if (tn == 0) {{   // the event falls on this chunk's first frame: the states as the chunk before left them go to memory")
                self.emit_flush_states(ind + "  ")
                L.append(f"{ind}  break;")
                L.append(f"{ind}}}")
                self.before_cut = False
                self.retire()
                continue
            if kind == "loop":
                if top:
                    L.append(f"{ind}ZT_STAMP(2)")
                self.emit_loop(it[1], ind)
                if top:
                    L.append(f"{ind}ZT_STAMP({8 + (it[1].loop.id // 2) % 48})")
            elif kind == "par" and it[1].kind == "ld":
                self.emit_load(it[1], ind)
            elif kind == "par" and it[1].kind == "lcin":
                n = it[1]
                A = f"la{reg.loop.cells[n.name].i}"
                L.append(f"{ind}const double n{n.i} = ZT_UNI({self.cell_ld(reg.loop, n.name, A)});     // a cell this loop only reads")
            elif kind == "par":
                n = it[1]
                e = _expr(n.op, [ref(x) for x in n.args])
                L.append(f"{ind}const double n{n.i} = {('ZT_UNI(' + e + ')') if (n.uniform and not top) else e};")
                if not top and n is reg.loop.cond:
                    L.append(f"{ind}if (!za_truthy(n{n.i})) break;")
                for ld in p.loads:            # the reads of this address go out now: their latency overlaps everything up to their use
                    if ld.args[0] is n and ld.i not in self.raw_issued and not any(s.mode == "early" for s in p.stores):
                        self.raw_issued.add(ld.i)
                        L.append(f"{ind}const int64_t B{ld.i} = (int64_t)n{n.i};")
                        L.append(f"{ind}const double raw{ld.i} = B{ld.i} < mcap ? memp[B{ld.i} * mse] : 0.0;")
                if not top:                   # a per-trip cell's address: its value before the chunk
                    for key, a in reg.loop.cells.items():
                        if a is n:
                            L.append(f"{ind}const int64_t la{n.i} = (int64_t)n{n.i};")
                            if key in reg.st:
                                L.append(f"{ind}const double lc{reg.loop.cin[key].i} = ZT_UNI({self.cell_ld(reg.loop, key, 'la%d' % n.i)});")
                            break
            elif kind == "shift":
                name = it[1]
                L.append(f"{ind}const double n{reg.st[name].i} = zt_shift1({ref(reg.outs[name])}, {self.carry(reg, name)});   // {name}[t-1]")
            elif kind == "scan":
                self.emit_scan(reg, it[1], ind)
            elif kind == "modc":
                c = it[1]
                nm, sid = c.names[0], reg.st[c.names[0]].i
                cv, kk, nn = self.carry(reg, nm), ref(c.modk), ref(c.modn)
                L.append(f"{ind}// {nm}: a wrapped counter, (y + K) % N over non-negative integers: the state before frame t is (y + t K) % N")
                L.append(f"{ind}double k{sid};")
                pw2 = (f" && {cv} < {nn}" + (f" && zt_pow2({nn})" if c.modpow2 else "")) if c.modmask else ""
                L.append(f"{ind}if (zt_small_int({cv}) && {cv} >= 0.0 && zt_small_int({kk}) && {kk} >= 0.0 && zt_small_int({nn}) && {nn} >= 1.0 && {nn} < 2147483647.0 && {cv} + 64.0 * {kk} < 2147483647.0{pw2}) {{")
                L.append(f"{ind}  k{sid} = lane == 0 ? {cv} : za_mod({cv} + (double)lane * {kk}, {nn});")
                L.append(f"{ind}}} else {{")
                mark = len(L)
                self.serial_loop(reg, [c], ind + "  ")
                # (the loop declares y / k itself: keep its k as the block-local it is and copy it out)
                L[mark] = L[mark].replace(f"k{sid} = {cv};", f"zk{sid} = {cv};")
                for q in range(mark + 1, len(L)):
                    L[q] = L[q].replace(f"k{sid} = me ? y{sid} : k{sid};", f"zk{sid} = me ? y{sid} : zk{sid};")
                L.append(f"{ind}  k{sid} = zk{sid};")
                L.append(f"{ind}}}")
                L.append(f"{ind}const double n{sid} = k{sid};")
            elif kind == "serial":
                names = [nm for c in it[1] for nm in c.names]
                L.append(f"{ind}// serial recurrences sharing one loop: {', '.join(names)}")
                if top:
                    L.append(f"{ind}ZT_STAMP(2)")
                self.serial_loop(reg, it[1], ind)
                if top:
                    L.append(f"{ind}ZT_STAMP(3)")
                for nm in names:
                    s = reg.st[nm].i
                    L.append(f"{ind}const double n{s} = k{s};")
            elif kind == "spec":
                if top:
                    L.append(f"{ind}ZT_STAMP(2)")
                self.emit_spec(reg, it[1], gid, ind)
                if top:
                    L.append(f"{ind}ZT_STAMP(4)")
            else:
                raise AssertionError(kind)
            if top and not self.before_cut:
                self.retire()
        flush_run()
        if sites_open:
            self.emit_site_pairs(ind)

    def emit_scan(self, reg: Region, c: Component, ind: str):
        L, ref = self.L, self.ref
        cn = lambda nm: self.carry(reg, nm)
        top = reg.loop is None
        if len(c.names) == 1 and c.A[0][0].kind == "const" and c.A[0][0].val == 1.0:
            nm = c.names[0]
            s = reg.st[nm].i
            L.append(f"{ind}const double n{s} = zt_shift1(zt_scan1_sum({ref(c.b[0])}, {cn(nm)}, lane), {cn(nm)});   // {nm}: running sum")
        elif len(c.names) == 1 and top and c.A[0][0] in self.inv_coefs:
            nm = c.names[0]
            s = reg.st[nm].i
            k = self.inv_coefs.index(c.A[0][0])
            L.append(f"{ind}const ZtPow sq{s} = {{zt_q[{k} * 4 + zo], zt_q[{k} * 4 + 1 + zo], zt_q[{k} * 4 + 2 + zo], zt_q[{k} * 4 + 3 + zo]}};   // {nm}: constant-coefficient recurrence")
            L.append(f"{ind}const double n{s} = zt_shift1(zt_scan1_inv({ref(c.b[0])}, {ref(c.A[0][0])}, sq{s}, zt_w[{k} * 64 + lane + zo], {cn(nm)}, lane), {cn(nm)});")
        elif len(c.names) == 1:
            nm = c.names[0]
            s = reg.st[nm].i
            L.append(f"{ind}double sa{s} = {ref(c.A[0][0])}, sb{s} = {ref(c.b[0])};   // {nm}: affine recurrence")
            L.append(f"{ind}zt_scan1(sa{s}, sb{s});")
            L.append(f"{ind}const double n{s} = zt_shift1(__builtin_fma(sa{s}, {cn(nm)}, sb{s}), {cn(nm)});")
        elif len(c.names) == 2 and top and tuple(x for row in c.A for x in row) in self.inv_mats:
            n0, n1 = c.names
            s0, s1 = reg.st[n0].i, reg.st[n1].i
            k = self.inv_mats.index(tuple(x for row in c.A for x in row))
            L.append(f"{ind}double sb{s0} = {ref(c.b[0])}, sb{s1} = {ref(c.b[1])};   // {n0}, {n1}: coupled pair, block-constant matrix")
            L.append(f"{ind}{{ const ZtMat2 am = {{{ref(c.A[0][0])}, {ref(c.A[0][1])}, {ref(c.A[1][0])}, {ref(c.A[1][1])}}};")
            L.append(f"{ind}  zt_scan2_inv(sb{s0}, sb{s1}, am, zt_m + {k} * ZT_MAT_TABLE_DOUBLES, zo, {cn(n0)}, {cn(n1)}, lane); }}")
            L.append(f"{ind}const double n{s0} = zt_shift1(sb{s0}, {cn(n0)});")
            L.append(f"{ind}const double n{s1} = zt_shift1(sb{s1}, {cn(n1)});")
        else:
            n0, n1 = c.names
            s0, s1 = reg.st[n0].i, reg.st[n1].i
            L.append(f"{ind}ZtMap2 sm{s0} = {{{ref(c.A[0][0])}, {ref(c.A[0][1])}, {ref(c.A[1][0])}, {ref(c.A[1][1])}, {ref(c.b[0])}, {ref(c.b[1])}}};   // {n0}, {n1}: coupled affine pair")
            L.append(f"{ind}zt_scan2(sm{s0});")
            L.append(f"{ind}const double n{s0} = zt_shift1(__builtin_fma(sm{s0}.a00, {cn(n0)}, __builtin_fma(sm{s0}.a01, {cn(n1)}, sm{s0}.b0)), {cn(n0)});")
            L.append(f"{ind}const double n{s1} = zt_shift1(__builtin_fma(sm{s0}.a10, {cn(n0)}, __builtin_fma(sm{s0}.a11, {cn(n1)}, sm{s0}.b1)), {cn(n1)});")

    def emit_spec(self, reg: Region, comps: List[Component], gid: int, ind: str):
        L, ref = self.L, self.ref
        cn = lambda nm: self.carry(reg, nm)
        tag = f"{reg.loop.id if reg.loop is not None else 0}_{gid}"
        names = [nm for c in comps for nm in c.names]
        L.append(f"{ind}// switched recurrences (affine once their state-dependent conditions are fixed), solved by iterating the")
        L.append(f"{ind}// condition pattern to its fixed point: {', '.join(names)}")
        for nm in names:
            L.append(f"{ind}double s{reg.st[nm].i} = {cn(nm)}, p{reg.st[nm].i} = {cn(nm)};")
        gname = {}
        for c in comps:
            for k, gn in enumerate(c.gnodes):
                gname[gn.i] = f"g{gn.name}_{k}"
                L.append(f"{ind}bool {gname[gn.i]};")

        def xref(x: N, loc: Dict[int, str]) -> str:
            if x.i in loc:
                return loc[x.i]
            if x.kind == "guess":
                return f"({gname[x.i]} ? 1.0 : 0.0)"
            return ref(x)

        def slice_eval(c: Component, ind2: str, out_prefix: str):
            loc = {reg.st[nm].i: f"s{reg.st[nm].i}" for nm in c.names}
            for m in c.slice:
                loc[m.i] = f"v{m.i}"
                L.append(f"{ind2}const double v{m.i} = {_expr(m.op, [xref(x, loc) for x in m.args])};")
            for k, (cnd, gn) in enumerate(zip(c.conds, c.gnodes)):
                L.append(f"{ind2}{out_prefix}{gname[gn.i]} = za_truthy({xref(cnd, loc)});")

        L.append(f"{ind}{{   // first pattern: the states taken to stay at their carried values")
        for c in comps:
            slice_eval(c, ind + "  ", "")
        L.append(f"{ind}}}")
        L.append(f"{ind}bool sch{tag}; int sit{tag} = 0, sst{tag} = 0;")
        L.append(f"{ind}do {{")
        for c in comps:
            loc: Dict[int, str] = {}
            for n in c.gdep:
                loc[n.i] = f"d{n.i}"
                L.append(f"{ind}  const double d{n.i} = {_expr(n.op, [xref(x, loc) for x in n.args])};")
            if len(c.names) == 1:
                nm = c.names[0]
                s = reg.st[nm].i
                L.append(f"{ind}  double sa{s} = {xref(c.A[0][0], loc)}, sb{s} = {xref(c.b[0], loc)};")
                L.append(f"{ind}  zt_scan1(sa{s}, sb{s});")
                L.append(f"{ind}  s{s} = zt_shift1(__builtin_fma(sa{s}, {cn(nm)}, sb{s}), {cn(nm)});")
            else:
                n0, n1 = c.names
                s0, s1 = reg.st[n0].i, reg.st[n1].i
                L.append(f"{ind}  ZtMap2 sm{s0} = {{{xref(c.A[0][0], loc)}, {xref(c.A[0][1], loc)}, {xref(c.A[1][0], loc)}, {xref(c.A[1][1], loc)}, {xref(c.b[0], loc)}, {xref(c.b[1], loc)}}};")
                L.append(f"{ind}  zt_scan2(sm{s0});")
                L.append(f"{ind}  s{s0} = zt_shift1(__builtin_fma(sm{s0}.a00, {cn(n0)}, __builtin_fma(sm{s0}.a01, {cn(n1)}, sm{s0}.b0)), {cn(n0)});")
                L.append(f"{ind}  s{s1} = zt_shift1(__builtin_fma(sm{s0}.a10, {cn(n0)}, __builtin_fma(sm{s0}.a11, {cn(n1)}, sm{s0}.b1)), {cn(n1)});")
        L.append(f"{ind}  // the pattern these states imply")
        for c in comps:
            slice_eval(c, ind + "  ", "const bool h")
        diffs = " || ".join(f"(h{gname[gn.i]} != {gname[gn.i]})" for c in comps for gn in c.gnodes)
        moved = " || ".join(f"(fabs(s{reg.st[nm].i} - p{reg.st[nm].i}) > ZT_SPEC_TOL * fmax(fabs(s{reg.st[nm].i}), fabs(p{reg.st[nm].i})))" for nm in names)
        L.append(f"{ind}  // settled = the pattern reproduced itself -- or it still flips, but only where its branches agree: the states have not")
        L.append(f"{ind}  // moved over two passes in a row, i.e. also under the pattern they themselves imply")
        L.append(f"{ind}  sst{tag} = (__ballot(valid && ({moved})) != 0ull) ? 0 : sst{tag} + 1;")
        L.append(f"{ind}  sch{tag} = (__ballot(valid && ({diffs})) != 0ull) && sst{tag} < 2;")
        for c in comps:
            for gn in c.gnodes:
                L.append(f"{ind}  {gname[gn.i]} = h{gname[gn.i]};")
        for nm in names:
            L.append(f"{ind}  p{reg.st[nm].i} = s{reg.st[nm].i};")
        L.append(f"{ind}}} while (sch{tag} && ++sit{tag} < ZT_SPEC_MAX);")
        L.append(f"{ind}if (sch{tag}) {{   // no fixed point within the budget (a pattern that keeps moving along the chunk): the serial loop")
        self.serial_loop(reg, comps, ind + "  ")
        for nm in names:
            s = reg.st[nm].i
            L.append(f"{ind}  s{s} = k{s};")
        L.append(f"{ind}}}")
        for nm in names:
            s = reg.st[nm].i
            L.append(f"{ind}const double n{s} = s{s};")

    def emit_loop(self, reg: Region, ind: str):
        """A uniform loop: trip k of all the chunk's frames, then trip k + 1."""
        p, L, ref = self.plan, self.L, self.ref
        Lp = reg.loop
        live = self.live_ids()
        carried = [v for v in Lp.order if Lp.phis[v].i in live or (v in Lp.louts and Lp.louts[v].i in live)]
        L.append(f"{ind}// uniform loop {Lp.id}: every frame runs the same trips; {len(carried)} values handed from trip to trip, {len(Lp.cell_out)} per-trip cells")
        for v in carried:
            L.append(f"{ind}double {self.phi_name[Lp.phis[v].i]} = {ref(Lp.init[v])};   // {v}")
        skip = Lp.entry_pred is not None and Lp.parent is None and not os.environ.get("ZA_TPAR_NO_LOOP_SKIP")
        if skip:
            # the loop stands under a condition: every value it hands on is merged with what was there before (if-conversion) and
            # its cells stay as they were where the condition is false, so a chunk none of whose frames takes the branch skips it
            ep = Lp.entry_pred
            test = f"za_truthy({ref(ep)})" if (ep.uniform or ep.kind == "const") else f"__ballot(valid && za_truthy({ref(ep)})) != 0ull"
            L.append(f"{ind}if ({test}) {{")
            ind0, ind = ind, ind + "  "
        self.emit_loop_forms(reg, ind, carried)
        if skip:
            L.append(f"{ind0}}}")

    def emit_loop_forms(self, reg: Region, ind: str, carried: List[str]):
        p, L, ref = self.plan, self.L, self.ref
        Lp = reg.loop
        groups = p.rings.get(Lp.id)
        steps = self.strip_steps(reg)
        if steps is not None:
            # counters that step by integers: 64 trips' worth of the loop's wave-uniform work can be done at once, one trip per
            # lane (emit_strip); checked here, per chunk
            conds = []
            for v, (c, sg) in steps.items():
                conds.append(f"zt_small_int({ref(Lp.init[v])}) && zt_small_int({ref(c)})")
            L.append(f"{ind}const bool zs{Lp.id} = {' && '.join(conds) if conds else 'true'};")
        if groups:
            self.emit_ring_stage(reg, groups, ind)
            L.append(f"{ind}if (zw{Lp.id}{' && zs%d' % Lp.id if steps is not None else ''}) {{")
            if steps is not None:
                self.emit_strip(reg, ind + "  ", carried, steps)
            else:
                self.emit_loop_body(reg, ind + "  ", carried)
            self.ring_lds = {}
            L.append(f"{ind}}} else {{     // (a window that does not fit, or a ring this is not: gathers from memory, trip by trip)")
            self.emit_loop_body(reg, ind + "  ", carried, batch=steps is None)
            L.append(f"{ind}}}")
        elif steps is not None and any(it[1].kind in ("ld", "lcin") for it in reg.items):
            L.append(f"{ind}if (zs{Lp.id}) {{")
            self.emit_strip(reg, ind + "  ", carried, steps)
            L.append(f"{ind}}} else {{     // (counters that are not small integers: trip by trip)")
            self.emit_loop_body(reg, ind + "  ", carried, batch=False)
            L.append(f"{ind}}}")
        else:
            self.emit_loop_body(reg, ind, carried)

    def strip_steps(self, reg: Region):
        """{counter: (step node, sign)} when the loop can run in strips of 64 trips: a counted loop of plain nodes whose wave-uniform
        values handed from trip to trip are all counters, next = this + / - a loop-invariant step. None otherwise."""
        Lp = reg.loop
        if os.environ.get("ZA_TPAR_NO_STRIP"):
            return None
        if Lp.count is None or reg.comps or reg.subs or Lp.cell_out or any(it[0] != "par" for it in reg.items):
            return None
        if any(it[1].kind == "ld" and self.has_late_site(it[1]) for it in reg.items):
            return None
        live = self.live_ids()
        out = {}
        for v in Lp.order:
            ph = Lp.phis[v]
            if not ph.uniform:
                continue
            if ph.i not in live and not (v in Lp.louts and Lp.louts[v].i in live):
                continue
            nx = Lp.next[v]
            if nx is ph:
                continue
            if nx.kind != "op" or nx.op not in ("+", "-") or len(nx.args) != 2:
                return None
            a, b = nx.args
            if a is ph and not _in_subtree(b, Lp) and (b.uniform or b.kind == "const"):
                out[v] = (b, 1 if nx.op == "+" else -1)
            elif nx.op == "+" and b is ph and not _in_subtree(a, Lp) and (a.uniform or a.kind == "const"):
                out[v] = (a, 1)
            else:
                return None
        return out

    def emit_strip(self, reg: Region, ind: str, carried: List[str], steps):
        """The loop in strips of 64 trips. Everything wave-uniform in a trip (counters, tap offsets, table reads) depends on the trip
        number only, so a strip computes it for 64 trips at once, one trip per lane; a trip then fetches its values from its lane
        (v_readlane) and does the per-frame work: for a FIR tap that is one LDS read and one multiply-add. Same operations on the
        same values as the trip-by-trip form (integer counters are exact either way), in the same order per frame."""
        p, L = self.plan, self.L
        Lp = reg.loop
        live = self.live_ids()
        nodes = [it[1] for it in reg.items]
        uni = [n for n in nodes if n.uniform]
        per = [n for n in nodes if not n.uniform]
        # per-frame nodes that only feed values nobody reads inside the loop (locals of a called function: `idx`, `lag`): needed
        # after the LAST trip only
        by_id = {n.i: n for n in nodes}

        def cone(roots) -> set:
            seen: set = set()
            todo = list(roots)
            while todo:
                n = todo.pop()
                if n.i in seen or n.loop is not Lp or n.i not in by_id:
                    continue
                seen.add(n.i)
                if not (n.kind == "ld" and n.i in self.ring_lds):     # (a staged read does not need its address)
                    todo.extend(n.args)
                if n.kind == "lcin":
                    todo.append(Lp.cells[n.name])
            return seen

        dead = [v for v in carried if Lp.phis[v].i not in live and not Lp.phis[v].uniform]
        while True:
            ccone = cone([Lp.next[v] for v in dead])
            clash = [v for v in dead if any(a.kind == "phi" and a.loop is Lp and not a.uniform
                                            for i in cone([Lp.next[v]]) for a in by_id[i].args)]
            if not clash:
                break
            dead = [v for v in dead if v not in clash]
        hot = cone([Lp.next[v] for v in carried if v not in dead and not Lp.phis[v].uniform])
        per_hot = [n for n in per if n.i in hot]
        cold = [n for n in per if n.i in ccone]
        ring_u_hot = [self.ring_u[m.i] for m in per_hot if m.i in self.ring_lds]
        uni_x = [Lp.phis[v] for v in steps] + uni         # (the counters are wave-uniform values of a trip too)
        exports_hot = [n for n in uni_x if any(n in m.args for m in per_hot if not (m.kind == "ld" and m.i in self.ring_lds))]
        exports_cold = [n for n in uni_x if any(n in m.args for m in cold if not (m.kind == "ld" and m.i in self.ring_lds))
                        or any(Lp.next[v] is n for v in dead)]
        uph = [v for v in steps]
        ind2, ind3 = ind + "  ", ind + "    "
        L.append(f"{ind}const int64_t zc{Lp.id} = za_loopcount(ZT_UNI({self.ref(Lp.count)}));")
        L.append(f"{ind}for (int64_t zs0 = 0; zs0 < zc{Lp.id}; zs0 += 64) {{     // a strip: lane j holds what trip zs0 + j needs")
        self.sctx = (Lp, "vec")
        for v in uph:
            c, sg = steps[v]
            nm = f"t{Lp.phis[v].i}"
            L.append(f"{ind2}const double {nm} = {self.ref_out(Lp.init[v])} {'+' if sg > 0 else '-'} (double)(zs0 + lane) * {self.ref_out(c)};   // {v}")
        for n in uni:
            if n.kind == "lcin":
                a = self.ref(Lp.cells[n.name])
                L.append(f"{ind2}const int64_t ta{n.i} = (int64_t)(int){a};")
                L.append(f"{ind2}const double t{n.i} = (zs0 + lane < zc{Lp.id} && ta{n.i} < mcap) ? memp[ta{n.i} * mse] : 0.0;     // (a cell this loop only reads)")
            else:
                L.append(f"{ind2}const double t{n.i} = {_expr(n.op, [self.ref(x) for x in n.args])};")
        # ring offsets as integers (one v_readlane per trip instead of two and a conversion)
        ring_int = {}
        self.sctx = (Lp, "vec")
        for m in per_hot + [c_ for c_ in cold if c_ not in per_hot]:
            if m.i in self.ring_lds:
                u = self.ring_u[m.i]
                ring_int[m.i] = f"to{m.i}"
                expr = self.ring_lds[m.i].replace('{U%d}' % m.i, self.ref(u))
                expr = expr.replace("lane + ", "")       # (the frame's lane is added per trip)
                L.append(f"{ind2}const int to{m.i} = {expr};")
        L.append(f"{ind2}const int zm = (int)(zc{Lp.id} - zs0 < 64 ? zc{Lp.id} - zs0 : 64);")
        L.append(f"{ind2}int zj = 0;")
        lv = [v for v in carried if v not in dead and not Lp.phis[v].uniform and Lp.next[v] is not Lp.phis[v]]
        G = int(os.environ.get("ZA_TPAR_STRIP_GROUP", "8"))
        if G > 1 and not any(n.kind == "ld" and n.i not in self.ring_lds for n in per_hot):
            # groups of G trips: every fetch of the group (lane reads, LDS reads) before its arithmetic, so that their latencies
            # overlap instead of adding up trip by trip
            L.append(f"{ind2}for (; zj + {G} <= zm; zj += {G}) {{")
            for u in range(G):
                self.sctx = (Lp, ("g", u))
                for n in exports_hot:
                    L.append(f"{ind3}const double e{n.i}_{u} = zt_readlane(t{n.i}, zj + {u});")
                for n in per_hot:
                    if n.kind == "ld":
                        lane_term = "lane + " if "lane + " in self.ring_lds[n.i] else ""
                        L.append(f"{ind3}const double n{n.i}_{u} = zt_ring[{lane_term}__builtin_amdgcn_readlane({ring_int[n.i]}, zj + {u})];")
            for u in range(G):
                self.sctx = (Lp, ("g", u))
                for n in per_hot:
                    if n.kind != "ld":
                        L.append(f"{ind3}const double n{n.i}_{u} = {_expr(n.op, [self.ref(x) for x in n.args])};")
            self.sctx = (Lp, ("g", G - 1))
            for v in lv:
                L.append(f"{ind3}const double q{self.phi_name[Lp.phis[v].i]} = {self.ref(Lp.next[v])};")
            for v in lv:
                L.append(f"{ind3}{self.phi_name[Lp.phis[v].i]} = q{self.phi_name[Lp.phis[v].i]};")
            L.append(f"{ind2}}}")
        self.sctx = (Lp, "trip")
        L.append(f"{ind2}for (; zj < zm; ++zj) {{")
        for n in exports_hot:
            L.append(f"{ind3}const double e{n.i} = zt_readlane(t{n.i}, zj);")

        def ring_read(n: N) -> str:
            lane_term = "lane + " if "lane + " in self.ring_lds[n.i] else ""
            return f"zt_ring[{lane_term}__builtin_amdgcn_readlane({ring_int[n.i]}, zj)]"

        for n in per_hot:
            if n.kind == "ld" and n.i in self.ring_lds:
                L.append(f"{ind3}const double n{n.i} = {ring_read(n)};")
            elif n.kind == "ld":
                L.append(f"{ind3}const int64_t B{n.i} = (int64_t)(int){self.ref(n.args[0])};")
                L.append(f"{ind3}const double n{n.i} = B{n.i} < mcap ? memp[B{n.i} * mse] : 0.0;")
                self.load_checks(n, f"B{n.i}", ind3, forward=False)
            else:
                L.append(f"{ind3}const double n{n.i} = {_expr(n.op, [self.ref(x) for x in n.args])};")
        tmp = [v for v in lv if Lp.next[v].kind == "phi"]
        for v in tmp:
            L.append(f"{ind3}const double q{self.phi_name[Lp.phis[v].i]} = {self.ref(Lp.next[v])};")
        for v in lv:
            src = f"q{self.phi_name[Lp.phis[v].i]}" if v in tmp else self.ref(Lp.next[v])
            L.append(f"{ind3}{self.phi_name[Lp.phis[v].i]} = {src};")
        L.append(f"{ind2}}}")
        if dead:
            self.sctx = (Lp, "cold")
            L.append(f"{ind2}if (zs0 + 64 >= zc{Lp.id}) {{   // after the last trip: values the loop hands on without reading them itself")
            L.append(f"{ind3}const int zj = zm - 1;")
            for n in exports_cold:
                L.append(f"{ind3}const double e{n.i} = zt_readlane(t{n.i}, zj);")
            for n in cold:
                if n.kind == "ld" and n.i in self.ring_lds:
                    L.append(f"{ind3}const double k{n.i} = {ring_read(n)};")
                elif n.kind == "ld":
                    L.append(f"{ind3}const int64_t Bk{n.i} = (int64_t)(int){self.ref(n.args[0])};")
                    L.append(f"{ind3}const double k{n.i} = Bk{n.i} < mcap ? memp[Bk{n.i} * mse] : 0.0;")
                else:
                    L.append(f"{ind3}const double k{n.i} = {_expr(n.op, [self.ref(x) for x in n.args])};")
            for v in dead:
                if Lp.next[v] is not Lp.phis[v]:
                    L.append(f"{ind3}{self.phi_name[Lp.phis[v].i]} = {self.ref(Lp.next[v])};")
            L.append(f"{ind2}}}")
        L.append(f"{ind}}}")
        self.sctx = None
        for v in uph:                                     # the counters after the loop
            c, sg = steps[v]
            if Lp.phis[v].i in live or (v in Lp.louts and Lp.louts[v].i in live):
                L.append(f"{ind}{self.phi_name[Lp.phis[v].i]} = {self.ref(Lp.init[v])} {'+' if sg > 0 else '-'} (double)zc{Lp.id} * {self.ref(c)};")

    def ref_out(self, n: N) -> str:
        """A node outside the loop, named from inside a strip."""
        save, self.sctx = self.sctx, None
        try:
            return self.ref(n)
        finally:
            self.sctx = save

    def emit_ring_stage(self, reg: Region, groups: List[RingGroup], ind: str):
        """Stage, per RingGroup of the loop, the chunk's window of the ring in LDS -- after checking everything the LDS form of
        the loop takes for granted: the mask is 2^k - 1, base and offsets are integers, consecutive frames sit one ring cell
        apart, the window fits, no staged cell belongs to another buffer's freshly written span or to a mem[] cell, and no read
        reaches a cell that a LATER frame of this chunk has already overwritten (the ring's own early write)."""
        p, L, ref = self.plan, self.L, self.ref
        Lp = reg.loop
        cap = f"(ZT_RING_DOUBLES / {len(groups)})"
        L.append(f"{ind}bool zw{Lp.id} = zrok{Lp.id};")
        for grp in groups:
            g = f"{Lp.id}_{grp.idx}"
            ssum = " + ".join(ref(x) for x in grp.S) if grp.S else "0.0"
            lo = " , ".join(f"zro_lo{ld.i}" for ld, _, _ in grp.loads)
            L.append(f"{ind}const double zwSd{g} = {ssum}, zwMd{g} = {ref(grp.mask)};")
            L.append(f"{ind}const int zwS{g} = (int)zwSd{g}, zwM{g} = (int)zwMd{g};")
            L.append(f"{ind}zw{Lp.id} &= (double)zwS{g} == zwSd{g} && zwS{g} >= 0 && (double)zwM{g} == zwMd{g} && zwM{g} >= 63 && (zwM{g} & (zwM{g} + 1)) == 0;")
            L.append(f"{ind}int zwo{g} = 2147483647, zwh{g} = -2147483647;")
            for ld, _, _ in grp.loads:
                L.append(f"{ind}zwo{g} = zro_lo{ld.i} < zwo{g} ? zro_lo{ld.i} : zwo{g}; zwh{g} = zro_hi{ld.i} > zwh{g} ? zro_hi{ld.i} : zwh{g};")
            step = 0 if grp.P.uniform else 1
            L.append(f"{ind}const int zwn{g} = zwh{g} >= zwo{g} ? zwh{g} - zwo{g} + {64 if step else 1} : 0;")
            L.append(f"{ind}zw{Lp.id} &= zwn{g} > 0 && zwn{g} <= {cap} && zwn{g} <= zwM{g} - 63;")
            if step:
                L.append(f"{ind}const double zwPd{g} = zt_readlane({ref(grp.P)}, 0);")
                L.append(f"{ind}const int zwP{g} = (int)zwPd{g};")
                L.append(f"{ind}zw{Lp.id} &= (double)zwP{g} == zwPd{g} && fabs(zwPd{g}) < 1.0e9 && __ballot(valid && ({ref(grp.P)} != zwPd{g} + (double)lane) && "
                         f"({ref(grp.P)} != zwPd{g} + (double)lane - (double)(zwM{g} + 1))) == 0ull;")
            else:
                L.append(f"{ind}const double zwPd{g} = {ref(grp.P)};")
                L.append(f"{ind}const int zwP{g} = (int)zwPd{g};")
                L.append(f"{ind}zw{Lp.id} &= (double)zwP{g} == zwPd{g} && fabs(zwPd{g}) < 1.0e9;")
            if grp.site is not None:
                j = grp.site.j
                L.append(f"{ind}if (zw{Lp.id}) {{   // the ring's own write of this chunk is in memory already: no read may reach a cell a later frame wrote")
                L.append(f"{ind}  const int M1 = zwM{g} + 1, w0 = (int)(zq0_{j} - zwS{g});")
                L.append(f"{ind}  const bool ring = w0 >= 0 && w0 <= zwM{g} && (zqk_{j} >= tn || (zq1_{j} == zwS{g} && ((w0 + zqk_{j}) & zwM{g}) == 0));")
                L.append(f"{ind}  const int a = (zwP{g} - w0 + zwo{g}) & zwM{g}, len = zwh{g} - zwo{g} + 1;")
                L.append(f"{ind}  zw{Lp.id} &= ring && !(len >= M1 - 64 || (a <= 63 && a + len - 1 >= 1) || a + len - 1 >= M1 + 1);")
                L.append(f"{ind}}}")
        L.append(f"{ind}if (zw{Lp.id}) {{")
        L.append(f"{ind}  bool zwb = false;")
        for grp in groups:
            g = f"{Lp.id}_{grp.idx}"
            off = f"{grp.idx} * {cap}"
            L.append(f"{ind}  for (int j = lane; j < zwn{g}; j += 64) {{")
            L.append(f"{ind}    const int64_t Be = (int64_t)zwS{g} + ((zwP{g} + zwo{g} + j) & zwM{g});")
            L.append(f"{ind}    zt_ring[{off} + j] = Be < mcap ? memp[Be * mse] : 0.0;")
            for st_ in p.stores:
                if st_ is grp.site:
                    continue
                j2 = st_.j
                if st_.mode == "sparse":
                    L.append(f"{ind}    zwb |= Be >= zlo{j2} && Be <= zhi{j2};")
                else:
                    on = f"zsu{j2} && " if st_.pred is not None else ""
                    L.append(f"{ind}    zwb |= {on}((uint64_t)(Be - zq0_{j2}) < (uint64_t)zqk_{j2} || (uint64_t)(Be - zq1_{j2}) < (uint64_t)(tn - zqk_{j2}));")
            if self.cell_addrs:
                L.append(f"{ind}    zwb |= Be >= cmin && Be <= cmax;")
            L.append(f"{ind}  }}")
            for ld, u, sign in grp.loads:
                lane_term = "lane + " if not grp.P.uniform else ""
                self.ring_lds[ld.i] = f"{off} + {lane_term}((int){'' if sign > 0 else '-'}{{U{ld.i}}} - zwo{g})"
        L.append(f"{ind}  zt_bad |= __ballot(zwb) != 0ull;")
        L.append(f"{ind}  __syncthreads();")
        L.append(f"{ind}}}")
        # (the offsets name each load's own U node: resolved where the load is emitted, plain or per sub-trip of a batch)
        self.ring_u = {ld.i: u for grp in groups for ld, u, _ in grp.loads}

    def emit_loop_body(self, reg: Region, ind: str, carried: List[str], batch: bool = True):
        p, L, ref = self.plan, self.L, self.ref
        Lp = reg.loop
        if Lp.count is not None:
            L.append(f"{ind}const int64_t zc{Lp.id} = za_loopcount(ZT_UNI({ref(Lp.count)}));")
            L.append(f"{ind}int64_t zk{Lp.id} = 0;")
            G = self.batch_width(reg) if batch else 1
            if G > 1:
                self.emit_batched(reg, ind, G, carried)
            L.append(f"{ind}for (; zk{Lp.id} < zc{Lp.id}; ++zk{Lp.id}) {{")
        else:
            L.append(f"{ind}for (int64_t zk{Lp.id} = 0; zk{Lp.id} < ZA_LOOP_CAP; ++zk{Lp.id}) {{")
            if Lp.cond is not None and not _in_subtree(Lp.cond, Lp):
                L.append(f"{ind}  if (!za_truthy({ref(Lp.cond)})) break;")
        self.emit_region(reg, ind + "  ")
        for key, o in Lp.cell_out.items():
            A = f"la{Lp.cells[key].i}"
            if (Lp.id, key) in self.cell_slot:
                j = self.cell_slot[(Lp.id, key)]
                L.append(f"{ind}  if (lane == last) {{ if (zlds{Lp.id}) zt_cells[zlo{Lp.id} + {j} * zln{Lp.id} + (int)zk{Lp.id}] = {ref(o)}; else memp[{A} * mse] = {ref(o)}; }}     // {key}: the cell after the chunk's last frame")
            else:
                L.append(f"{ind}  if (lane == last) memp[{A} * mse] = {ref(o)};     // {key}: the cell after the chunk's last frame")
            fl = Lp.cell_flag.get(key)
            if fl is not None and fl.kind != "const":
                test = f"za_truthy({ref(fl)})" if fl.uniform else f"__ballot(valid && za_truthy({ref(fl)}))"
                L.append(f"{ind}  if ({test}) zt_high = {A} + 1 > zt_high ? {A} + 1 : zt_high;")
        tmp = [v for v in carried if Lp.next[v].kind == "phi" and Lp.next[v].val == Lp.id and Lp.next[v] is not Lp.phis[v]]
        for v in tmp:
            L.append(f"{ind}  const double q{self.phi_name[Lp.phis[v].i]} = {ref(Lp.next[v])};")
        for v in carried:
            if Lp.next[v] is Lp.phis[v]:
                continue
            src = f"q{self.phi_name[Lp.phis[v].i]}" if v in tmp else ref(Lp.next[v])
            L.append(f"{ind}  {self.phi_name[Lp.phis[v].i]} = {src};")
        L.append(f"{ind}}}")

    def batch_levels(self, reg: Region):
        """Per node of a gather loop's trip: how many loads lie in front of it (None: it follows a value handed from trip to trip,
        i.e. it belongs to the accumulation phase). None for the whole loop when its trips cannot be batched."""
        Lp = reg.loop
        if Lp.count is None or reg.comps or reg.subs or Lp.cell_out or any(it[0] != "par" for it in reg.items):
            return None
        nodes = [it[1] for it in reg.items]
        if not any(n.kind in ("ld", "lcin") for n in nodes) or any(n.kind == "ld" and self.has_late_site(n) for n in nodes):
            return None
        lev: Dict[int, Optional[int]] = {}
        for n in nodes:
            deps = (Lp.cells[n.name],) if n.kind == "lcin" else n.args
            v: Optional[int] = 0
            for d in deps:
                if d.loop is not Lp or d.kind == "const":
                    continue
                if d.kind == "phi":
                    dl = 0 if d.uniform else None
                else:
                    dl = lev.get(d.i, 0)
                    if dl is not None and d.kind in ("ld", "lcin"):
                        dl += 1
                if dl is None:
                    v = None
                    break
                v = max(v, dl)
            lev[n.i] = v
        for v_ in Lp.order:                       # a counter's next value must not wait for a load
            ph = Lp.phis[v_]
            if ph.uniform and ph.i in self.live_ids():
                nx = Lp.next[v_]
                if nx.loop is Lp and nx.kind != "phi" and lev.get(nx.i, 0) != 0:
                    return None
        return lev

    def batch_width(self, reg: Region) -> int:
        env = os.environ.get("ZA_TPAR_GATHER_BATCH")
        if env is not None and int(env) <= 1:
            return 1
        lev = self.batch_levels(reg)
        if lev is None:
            return 1
        loads = sum(1 for it in reg.items if it[1].kind in ("ld", "lcin"))
        return int(env) if env is not None else (8 if loads <= 3 else 4)

    def emit_batched(self, reg: Region, ind: str, G: int, carried: List[str]):
        """G trips of a gather loop at a time: the trips' addresses first, then all of their loads (G memory latencies overlap
        instead of adding up), then the accumulation in trip order -- the same operations per trip as the plain loop behind it,
        which takes the remaining trips."""
        p, L = self.plan, self.L
        Lp = reg.loop
        lev = self.batch_levels(reg)
        nodes = [it[1] for it in reg.items]
        top = max((v for v in lev.values() if v is not None), default=0)
        ind2 = ind + "  "
        L.append(f"{ind}for (; zk{Lp.id} + {G} <= zc{Lp.id}; zk{Lp.id} += {G}) {{     // {G} trips per pass: their loads are in flight together")

        def finish(level: int):
            for u in range(G):
                self.bctx = (Lp, u)
                for n in nodes:
                    if lev[n.i] != level:
                        continue
                    if n.kind == "lcin":
                        L.append(f"{ind2}const double n{n.i}_{u} = ZT_UNI(r{n.i}_{u});")
                    elif n.kind == "ld":
                        if n.i not in self.ring_lds:
                            self.load_checks(n, f"B{n.i}_{u}", ind2, forward=False)
                        L.append(f"{ind2}const double n{n.i}_{u} = r{n.i}_{u};")

        for level in range(top + 1):
            if level:
                finish(level - 1)
            for u in range(G):
                self.bctx = (Lp, u)
                for n in nodes:
                    if lev[n.i] != level:
                        continue
                    if n.kind == "lcin":
                        a = self.ref(Lp.cells[n.name])
                        L.append(f"{ind2}const int64_t la{n.i}_{u} = (int64_t){a};")
                        L.append(f"{ind2}const double r{n.i}_{u} = la{n.i}_{u} < mcap ? memp[la{n.i}_{u} * mse] : 0.0;     // (a cell this loop only reads)")
                    elif n.kind == "ld" and n.i in self.ring_lds:
                        L.append(f"{ind2}const double r{n.i}_{u} = zt_ring[{self.ring_lds[n.i].replace('{U%d}' % n.i, self.ref(self.ring_u[n.i]))}];")
                    elif n.kind == "ld":
                        L.append(f"{ind2}const int64_t B{n.i}_{u} = (int64_t){self.ref(n.args[0])};")
                        L.append(f"{ind2}const double r{n.i}_{u} = B{n.i}_{u} < mcap ? memp[B{n.i}_{u} * mse] : 0.0;")
                    else:
                        e = _expr(n.op, [self.ref(x) for x in n.args])
                        L.append(f"{ind2}const double n{n.i}_{u} = {('ZT_UNI(' + e + ')') if n.uniform else e};")
        finish(top)
        for u in range(G):                            # the accumulation, trip by trip
            self.bctx = (Lp, u)
            for n in nodes:
                if lev[n.i] is None:
                    e = _expr(n.op, [self.ref(x) for x in n.args])
                    L.append(f"{ind2}const double n{n.i}_{u} = {e};")
        self.bctx = (Lp, G - 1)
        for v in carried:
            if Lp.next[v] is Lp.phis[v]:
                continue
            L.append(f"{ind2}const double q{self.phi_name[Lp.phis[v].i]} = {self.ref(Lp.next[v])};")
        self.bctx = None
        for v in carried:
            if Lp.next[v] is not Lp.phis[v]:
                L.append(f"{ind2}{self.phi_name[Lp.phis[v].i]} = q{self.phi_name[Lp.phis[v].i]};")
        L.append(f"{ind}}}")

    # -- the serial finish ---------------------------------------------------------------------------------------------------------
    def emit_tail(self):
        p, L = self.plan, self.L
        km = self.km
        # the generic code of the leaf, one lane per instance, from wherever the kernel above stopped (normally: nowhere)
        L.append("// instances the time-parallel kernel handed back (b.resume[i] < frames) finish the launch here, frame by frame, with the")
        L.append("// generic section code -- the exact serial semantics; every other lane leaves at once. The block the hand-back happened in")
        L.append("// has had its @block already.")
        L.append(f'extern "C" __global__ void __launch_bounds__(64) {km[:-1]}_tail)(ZabBatch b, ZabAudio a) {{')
        L.append("  ZA_KERNEL_ENTRY();")
        L.append("  const int inst = blockIdx.x * 64 + threadIdx.x;")
        L.append("  if (inst >= b.n_inst) return;")
        L.append("  const int64_t from = b.resume[inst];")
        L.append("  if (from >= a.frames) return;")
        L.append("  // (what the host can ask for afterwards, zab_handback_stats: instances handed back in this zab_process call, frames run here)")
        L.append("  atomicAdd((unsigned long long*)&b.resume[b.n_pad], 1ull);")
        L.append("  atomicAdd((unsigned long long*)&b.resume[b.n_pad + 1], (unsigned long long)(a.frames - from));")
        L.append("  ZaS s;")
        L.append("  za_state_load(s, b, inst);")
        L.append("  uint64_t pend_seen = 0;")
        L.append(f"  const float* in = a.in + (int64_t)inst * {p.nch} * a.frame_stride;")
        L.append(f"  float* out = a.out + (int64_t)inst * {p.nch} * a.frame_stride;")
        L.append(f"  const int64_t blk = {'a.block > 0 ? (int64_t)a.block : a.frames' if p.has_block else 'a.frames'};")
        L.append("  { const int64_t b0 = (from / blk) * blk, n0 = a.frames - b0 < blk ? a.frames - b0 : blk; s.samplesblock = (double)n0; s.block_size = (int)n0; }")
        L.append("  for (int64_t t = from; t < a.frames; ++t) {")
        if p.has_block:
            L.append("    if (t != from && t % blk == 0) {")
            L.append("      const int64_t n = a.frames - t < blk ? a.frames - t : blk;")
            L.append("      s.samplesblock = (double)n;")
            L.append("      s.block_size = (int)n;")
            L.append("#if ZA_USES_MSG")
            L.append("      za_msg_begin_block(s);")
            L.append("#endif")
            L.append("      za_section_block(s);")
            L.append("      if (s.pend_change | s.pend_automate | s.pend_automate_end) za_section_slider(s);")
            L.append("      pend_seen |= s.pend_change | s.pend_automate | s.pend_automate_end;")
            L.append("      s.pend_change = s.pend_automate = s.pend_automate_end = 0;")
            L.append("    }")
        for ch in range(p.nch):
            L.append(f"    s.spl[{ch}] = (double)in[{ch} * a.frame_stride + t];")
        L.append("    za_section_sample(s);")
        for ch in range(p.nch):
            L.append(f"    out[{ch} * a.frame_stride + t] = (float)s.spl[{ch}];")
        L.append("  }")
        L.append("  za_state_store(s, b, inst);")
        L.append("  if (pend_seen) b.pend[3 * (int64_t)b.n_pad + inst] |= pend_seen;")
        L.append("  b.resume[inst] = a.frames;")
        L.append("}")


def emit_hip(plan: Plan, prog: Program, kernel_macro: str = "ZA_KERNEL(tpar)") -> str:
    """Kernel + launcher text, appended to a leaf module after zab_generic.hip.h (which defines ZabBatch / ZabAudio)."""
    return _Emit(plan, prog, kernel_macro).emit()


# ----------------------------------------------------------------------------------------------------------------------
# 4. numpy restatement of the staged algorithm (tests)
# ----------------------------------------------------------------------------------------------------------------------
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


def _is_hold(a):
    return np.asarray(a, dtype=np.float64).view(np.uint64) == np.uint64(HOLD_BITS)


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


def _sites_ok(a0, s0, lo0, hi0, a1, s1, lo1, hi1) -> bool:
    """zt_sites_ok of csrc/zart_tpar.h: two address sequences a0 + k * s0 and a1 + k * s1 never name one cell."""
    if hi0 < lo0 or hi1 < lo1:
        return True                         # (no trips)
    if hi0 < lo1 or hi1 < lo0:
        return True
    if s0 != s1 or s0 == 0:
        return False
    return (a1 - a0) % s0 != 0


class _Recut(Exception):
    """Plan.simulate: the chunk ends before frame e (a feedback read would need a value of its own chunk)."""

    def __init__(self, e):
        super().__init__(e)
        self.e = e


class _Sim:
    """State of one Plan.simulate call."""

    def __init__(self, plan: Plan, memv, tn):
        self.plan, self.memv, self.tn = plan, memv, tn


def _simulate(self: Plan, vars0: Dict[str, float], x: np.ndarray, sliders=None, srate=48000.0, spl0=None, mt=None, mem=None):
    """x: [nch, frames] float32. vars0: name -> value before the launch (missing names are 0). mt: (randMT[624], randIndex)
    before the launch for scripts that call rand(); self.mt_after holds the pair after it. mem: the arena before the launch
    (numpy doubles) for scripts that touch mem[]; self.mem_after / self.mem_high_after hold it after. The launch is taken as
    one block (@block, if the script has one, is not run here).
    Returns (y float32 [nch, frames], vars after {name: value}, spl after {k: value}). Raises TparAbort when a chunk breaks
    one of the run-time conditions of the lowering (the kernel hands such a launch to the serial code)."""
    if self.events or any(L.guards for L in self.loops):
        raise NotImplementedError("plans with events run a frame of the script's own section code: device only")
    memv = np.zeros(1 << 16) if mem is None else np.array(mem, dtype=np.float64)
    mcap = len(memv)
    mem_high = [0]
    stream = MtStream(*(mt if mt is not None else (None, 0))) if self.uses_rand else None
    _MT_CTX[0] = stream
    x = np.asarray(x, dtype=np.float32)
    frames = x.shape[1]
    sliders = np.zeros(64) if sliders is None else np.asarray(sliders, dtype=np.float64)
    spl_state = dict(spl0 or {})
    top = self.top
    lane = np.arange(WAVE)

    def inv_value(name):
        k = is_slider_name(name)
        if k is not None:
            return float(sliders[k - 1])
        if name == "srate":
            return float(srate)
        if name == "samplesblock":
            return float(frames)
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

    def V(n: N):
        if n.kind == "const":
            return np.float64(n.val)
        if n.kind == "hold":
            return HOLD
        return val[n.i]

    def vec(n: N):
        return np.broadcast_to(V(n), (WAVE,)).astype(np.float64)

    def uni(n: N, what):
        v = V(n)
        if np.ndim(v):
            if not np.all(v.view(np.uint64) == v.view(np.uint64)[0]):
                raise AssertionError(f"{what}: not wave-uniform")
            v = v[0]
        return float(v)

    def sim_serial(reg: Region, comp: Component, carry, tn):
        cur = {nm: carry[nm] for nm in comp.names}
        caps = {nm: np.zeros(WAVE) for nm in comp.names}
        for t in range(tn):
            loc: Dict[int, np.float64] = {}
            for nm in comp.names:
                caps[nm][t] = cur[nm]
                loc[reg.st[nm].i] = cur[nm]
            for m in comp.members:
                if m.kind in ("st", "lcin"):
                    continue
                ops = []
                for a in m.args:
                    if a.i in loc:
                        ops.append(loc[a.i])
                    else:
                        v = V(a)
                        ops.append(v if np.ndim(v) == 0 else v[t])
                loc[m.i] = np.float64(_np_op(m.op, ops))
            for nm in comp.names:
                cur[nm] = loc[reg.outs[nm].i] if reg.outs[nm].i in loc else np.float64(vec(reg.outs[nm])[t])
        for nm in comp.names:
            caps[nm][tn:] = cur[nm]
            val[reg.st[nm].i] = caps[nm]

    live_ids = {n.i for r in [self.top] + list(self.regions.values()) for n in r.nodes}
    lbox = [1 << 62, -1]          # bounding box of the loops' per-trip cells (address_pass)
    wbox = [1 << 62, -1]          # ... of those that loops store to

    def run_items(reg: Region, carry, f0, tn, sites):
        """One chunk's (or one trip's) schedule. carry: state name -> value before the chunk."""
        for it in reg.items:
            kind = it[0]
            if kind == "site":
                st_: StoreSite = it[1]
                A = vec(st_.addr).astype(np.int64)
                if st_.mode == "sparse":
                    on = _truthy(vec(st_.pred)) & (lane < tn)
                    si = {"A": A, "on": on, "lo": int(A[on].min()) if on.any() else 0, "hi": int(A[on].max()) if on.any() else -1}
                    if si["hi"] >= mcap or any(si["lo"] <= a <= si["hi"] for a in cell_addr.values()):
                        raise TparAbort(f0, "a conditional write leaves the arena or runs over a mem[] cell")
                    if si["lo"] <= lbox[1] and si["hi"] >= lbox[0]:
                        raise TparAbort(f0, "a conditional write lands among a loop's per-trip cells")
                    sites[st_.j] = si
                    continue
                live_site = st_.pred is None or bool(_truthy(np.float64(uni(st_.pred, "store condition"))))
                d = np.diff(A[:tn])
                brk = np.flatnonzero(d != 1)
                k = int(brk[0]) + 1 if len(brk) else tn
                si = {"A": A, "a0": int(A[0]), "k": k, "ak": int(A[k]) if k < tn else 0, "live": live_site}
                sites[st_.j] = si
                if live_site:
                    if len(brk) > 1 or A[:tn].min() < 0 or A[:tn].max() >= len(memv):
                        raise TparAbort(f0, "a delay-line write does not advance by one cell per frame (or leaves the arena)")
                    if any(lo <= a <= hi for a in cell_addr.values() for lo, hi in ((A[:tn].min(), A[:tn].max()),)):
                        raise TparAbort(f0, "a delay line runs over a mem[] cell")
                    if A[:tn].min() <= lbox[1] and A[:tn].max() >= lbox[0]:
                        raise TparAbort(f0, "a delay line runs over a loop's per-trip cells")
                if len(sites) == len(self.stores):         # every span known: no two writes may touch one cell
                    spans = {}
                    for s2 in self.stores:
                        q = sites[s2.j]
                        if s2.mode == "sparse":
                            spans[s2.j] = set(int(a) for a in q["A"][q["on"]])
                        else:
                            spans[s2.j] = set(int(a) for a in q["A"][:tn]) if q["live"] else set()
                    js = list(spans)
                    for i1 in range(len(js)):
                        for i2 in range(i1 + 1, len(js)):
                            lo1, hi1 = (min(spans[js[i1]]), max(spans[js[i1]])) if spans[js[i1]] else (0, -1)
                            lo2, hi2 = (min(spans[js[i2]]), max(spans[js[i2]])) if spans[js[i2]] else (0, -1)
                            sa_, sb_ = self.stores[js[i1]], self.stores[js[i2]]
                            both_sparse = sa_.mode == "sparse" and sb_.mode == "sparse"
                            if (sa_.region == sb_.region and sa_.mode == "late" and sb_.mode == "late"
                                    and np.array_equal(sites[sa_.j]["A"][:tn], sites[sb_.j]["A"][:tn])):
                                continue                   # writes into one delay line that move in step: program order decides
                            if (hi1 >= lo1 and hi2 >= lo2 and lo1 <= hi2 and lo2 <= hi1) if both_sparse else (spans[js[i1]] & spans[js[i2]]):
                                raise TparAbort(f0, "two writes of a chunk touch one cell")
                    for s2 in self.stores:                  # early writes go out now
                        if s2.mode == "early" and sites[s2.j]["live"]:
                            q = sites[s2.j]
                            q["old"] = memv[q["A"][:tn]].copy()
                            memv[q["A"][:tn]] = vec(s2.value)[:tn]
            elif kind == "par" and it[1].kind == "ld":
                n = it[1]
                B = vec(n.args[0]).astype(np.int64)
                out = np.where(B < len(memv), memv[np.minimum(B, len(memv) - 1)], 0.0)
                best = np.full(WAVE, -1)
                for st_ in self.stores:
                    si = sites[st_.j]
                    if st_.mode == "sparse":
                        if np.any((B[:tn] >= si["lo"]) & (B[:tn] <= si["hi"])):
                            raise TparAbort(f0, "a delay-line read falls into a conditional write's span")
                        continue
                    if not si["live"]:
                        continue
                    tw = np.full(WAVE, -1)
                    d0 = B - si["a0"]
                    tw = np.where((d0 >= 0) & (d0 < si["k"]), d0, tw)
                    d1 = B - si["ak"]
                    tw = np.where((d1 >= 0) & (d1 < tn - si["k"]), si["k"] + d1, tw)
                    if ",".join(map(str, st_.region)) != n.name:
                        if np.any(tw[:tn] >= 0):
                            raise TparAbort(f0, "a delay-line read falls into another buffer's freshly written span")
                        continue
                    if st_.mode == "early":
                        late = (tw > lane) | ((tw == lane) & (not st_.seq < n.val))
                        if np.any(late[:tn]):
                            raise TparAbort(f0, "a gather reads a cell that a later frame of the chunk has already overwritten")
                        continue
                    if st_.j in n.fb:
                        if np.any(((tw >= 0) & ((tw < lane) | ((tw == lane) & (st_.seq < n.val))))[:tn]):
                            raise TparAbort(f0, "a feedback read would need a value of its own chunk (the cut should have ended it)")
                        continue
                    vis = (tw >= 0) & ((tw < lane) | ((tw == lane) & (st_.seq < n.val))) & (tw >= best)
                    Vv = vec(st_.value)
                    out = np.where(vis, Vv[np.clip(tw, 0, WAVE - 1)], out)
                    best = np.where(vis, tw, best)
                if any(np.any(B[:tn] == a) for a in cell_addr.values()):
                    raise TparAbort(f0, "a delay-line read hits a mem[] cell")
                if np.any((B[:tn] >= wbox[0]) & (B[:tn] <= wbox[1])):
                    raise TparAbort(f0, "a read at a moving address hits a per-trip cell that a loop stores to")
                val[n.i] = out
            elif kind == "par" and it[1].kind == "lcin":
                n = it[1]
                a = int(uni(reg.loop.cells[n.name], "cell address"))
                val[n.i] = np.float64(memv[a] if a < mcap else 0.0)
            elif kind == "par":
                n = it[1]
                val[n.i] = _np_op(n.op, [V(a) for a in n.args])
                if reg.loop is None or not n.uniform:
                    val[n.i] = np.broadcast_to(val[n.i], (WAVE,)).astype(np.float64)
                if reg.loop is not None and n is reg.loop.cond and not _truthy(np.float64(uni(n, "while condition"))):
                    return False
            elif kind == "loop":
                run_loop(it[1], f0, tn, sites)
            elif kind == "cut":
                hit = np.zeros(WAVE, dtype=bool)
                for ld in self.fb_loads:
                    B = vec(ld.args[0]).astype(np.int64)
                    for st_ in self.stores:
                        if st_.j not in ld.fb:
                            continue
                        if st_.pred is not None and not _truthy(np.float64(uni(st_.pred, "store condition"))):
                            continue
                        A = vec(st_.addr).astype(np.int64)
                        brk = np.flatnonzero(np.diff(A[:tn]) != 1)
                        k = int(brk[0]) + 1 if len(brk) else tn
                        tw = np.full(WAVE, -1)
                        d0 = B - int(A[0])
                        tw = np.where((d0 >= 0) & (d0 < k), d0, tw)
                        if k < tn:
                            d1 = B - int(A[k])
                            tw = np.where((d1 >= 0) & (d1 < tn - k), k + d1, tw)
                        hit |= (tw >= 0) & ((tw < lane) | ((tw == lane) & (st_.seq < ld.val)))
                hit[tn:] = False
                if hit.any():
                    e = int(np.flatnonzero(hit)[0])
                    if e < 16:
                        raise TparAbort(f0, "a feedback delay shorter than 16 frames")
                    raise _Recut(e)
            elif kind == "shift":
                name = it[1]
                src = vec(reg.outs[name])
                sh = np.empty(WAVE)
                sh[0] = carry[name]
                sh[1:] = src[:-1]
                val[reg.st[name].i] = sh
            elif kind == "scan":
                comp: Component = it[1]
                d = len(comp.names)
                A = np.stack([np.stack([vec(comp.A[r][c]) for c in range(d)]) for r in range(d)]).copy()
                b = np.stack([vec(comp.b[r]) for r in range(d)]).copy()          # [d,d,64], [d,64]
                states = _scan_exclusive(A, b, np.array([carry[nm] for nm in comp.names]))
                for r, nm in enumerate(comp.names):
                    val[reg.st[nm].i] = states[r]
            elif kind == "modc":
                comp = it[1]
                nm = comp.names[0]
                c0, kk, nn = float(carry[nm]), float(uni(comp.modk, "counter step")), float(uni(comp.modn, "counter length"))
                small = lambda x_: x_ == math.floor(x_) and abs(x_) < 1.0e12
                pw2 = (not comp.modmask) or (c0 < nn and (not comp.modpow2 or (nn >= 1 and (int(nn) & (int(nn) - 1)) == 0)))
                if small(c0) and c0 >= 0 and small(kk) and kk >= 0 and small(nn) and 1 <= nn < 2147483647.0 and c0 + 64 * kk < 2147483647.0 and pw2:
                    t = np.arange(WAVE, dtype=np.float64)
                    st_v = np.mod(c0 + t * kk, nn)
                    st_v[0] = c0
                    val[reg.st[nm].i] = st_v
                else:
                    sim_serial(reg, comp, carry, tn)
            elif kind == "serial":
                for comp in it[1]:
                    sim_serial(reg, comp, carry, tn)
            elif kind == "spec":
                for comp in it[1]:
                    d = len(comp.names)

                    def conds_from(states):
                        loc = {reg.st[nm].i: states[r] for r, nm in enumerate(comp.names)}
                        for m in comp.slice:
                            loc[m.i] = np.broadcast_to(_np_op(m.op, [loc[a.i] if a.i in loc else V(a) for a in m.args]), (WAVE,))
                        return [_truthy(np.broadcast_to(loc[c.i] if c.i in loc else V(c), (WAVE,))) for c in comp.conds]

                    prev = [np.full(WAVE, carry[nm]) for nm in comp.names]
                    gs = conds_from(prev)
                    converged, iters, still = False, 0, 0
                    while iters < SPEC_MAX:
                        iters += 1
                        loc = {gn.i: np.where(gs[k], 1.0, 0.0) for k, gn in enumerate(comp.gnodes)}
                        for n in comp.gdep:
                            loc[n.i] = _np_op(n.op, [loc[a.i] if a.i in loc else V(a) for a in n.args])
                        gv = lambda n: np.broadcast_to(loc[n.i] if n.i in loc else V(n), (WAVE,)).astype(np.float64)
                        A = np.stack([np.stack([gv(comp.A[r][c]) for c in range(d)]) for r in range(d)]).copy()
                        b = np.stack([gv(comp.b[r]) for r in range(d)]).copy()
                        states = _scan_exclusive(A, b, np.array([carry[nm] for nm in comp.names]))
                        ng = conds_from(states)
                        changed = any(bool(np.any(x_[:tn] != y_[:tn])) for x_, y_ in zip(ng, gs))
                        # a pattern that only still flips where both of its branches agree (a smoother sitting on its target,
                        # a value on its clamp) leaves the states where they were, also under the pattern they themselves imply
                        moved = any(bool(np.any(np.abs(a_[:tn] - b_[:tn]) > SPEC_TOL * np.maximum(np.abs(a_[:tn]), np.abs(b_[:tn]))))
                                    for a_, b_ in zip(states, prev))
                        still = 0 if moved else still + 1
                        gs, prev = ng, states
                        if not changed or still >= 2:
                            converged = True
                            break
                    self.spec_log.append((tuple(comp.names), iters, converged))
                    if converged:
                        for r, nm in enumerate(comp.names):
                            val[reg.st[nm].i] = states[r]
                    else:
                        sim_serial(reg, comp, carry, tn)
            else:
                raise AssertionError(kind)
        return True

    class _CellCarry:
        """A per-trip cell's value before the chunk: read when its recurrence runs (its address is a node of the trip)."""

        def __init__(self, Lp):
            self.Lp = Lp

        def __getitem__(self, key):
            return np.float64(memv[int(uni(self.Lp.cells[key], "cell address"))])

    def run_loop(reg: Region, f0, tn, sites):
        Lp = reg.loop
        carried_ = [v for v in Lp.order if Lp.phis[v].i in live_ids or (v in Lp.louts and Lp.louts[v].i in live_ids)]
        for v in carried_:
            val[Lp.phis[v].i] = V(Lp.init[v]) if Lp.phis[v].su else vec(Lp.init[v])
        cnt = None
        if Lp.count is not None:
            c = uni(Lp.count, "loop count")
            cnt = 0 if not c > 0 else int(min(c, 134217728.0))
        k = 0
        while cnt is None or k < cnt:
            if Lp.cond is not None and not _in_subtree(Lp.cond, Lp) and not _truthy(np.float64(uni(Lp.cond, "while condition"))):
                break
            if Lp.cond is not None and k >= (1 << 26):
                raise AssertionError("loop cap")
            if not run_items(reg, _CellCarry(Lp), f0, tn, sites):
                break
            for key, o in Lp.cell_out.items():
                a = int(uni(Lp.cells[key], "cell address"))
                memv[a] = vec(o)[tn - 1]
                fl = Lp.cell_flag.get(key)
                if fl is None or np.any(_truthy(vec(fl))[:tn]):
                    mem_high[0] = max(mem_high[0], a + 1)
            nxt = {v: V(Lp.next[v]) for v in carried_}
            for v in carried_:
                val[Lp.phis[v].i] = nxt[v]
            k += 1
        for v, lo in Lp.louts.items():
            if v in carried_:
                val[lo.i] = vec(Lp.phis[v])

    def address_pass(Lp: LoopInfo):
        """The kernel's check before a block: every per-trip cell address steps evenly and no two ever meet."""
        reg = self.regions[Lp.id]
        need = sorted({n.i: n for n in reg.nodes if n.uniform and n.kind == "op"}.values(), key=lambda n: n.i)
        for v in Lp.order:
            if Lp.phis[v].su:
                val[Lp.phis[v].i] = V(Lp.init[v])
        seqs: Dict[str, List[int]] = {key: [] for key in Lp.cells}
        cnt = None
        if Lp.count is not None:
            c = uni(Lp.count, "loop count")
            cnt = 0 if not c > 0 else int(min(c, 134217728.0))
        k = 0
        while cnt is None or k < cnt:
            stop = False
            for n in need:
                try:
                    val[n.i] = _np_op(n.op, [V(a) for a in n.args])
                except KeyError:
                    continue                  # (depends on a cell's value: not an address)
                if n is Lp.cond and not _truthy(np.float64(val[n.i])):
                    stop = True
                    break
            if stop or (Lp.cond is not None and not _in_subtree(Lp.cond, Lp) and not _truthy(np.float64(uni(Lp.cond, "cond")))):
                break
            for key, a in Lp.cells.items():
                seqs[key].append(int(uni(a, "cell address")))
            nxt = {v: V(Lp.next[v]) for v in Lp.order if Lp.phis[v].su}
            for v, x_ in nxt.items():
                val[Lp.phis[v].i] = x_
            k += 1
        desc = {}
        for key, s in seqs.items():
            if not s:
                desc[key] = (0, 1, 0, -1)
                continue
            st = s[1] - s[0] if len(s) > 1 else 1
            if any(b_ - a_ != st for a_, b_ in zip(s, s[1:])) or max(s) >= mcap:
                raise TparAbort(0, "a per-trip cell address does not step evenly through the trips (or leaves the arena)")
            desc[key] = (s[0], st, min(s), max(s))
        for key_, (_, _, lo_, hi_) in desc.items():
            if hi_ >= lo_:
                lbox[0], lbox[1] = min(lbox[0], lo_), max(lbox[1], hi_)
                if key_ in Lp.cell_out:
                    wbox[0], wbox[1] = min(wbox[0], lo_), max(wbox[1], hi_)
        keys = list(desc)
        for i1 in range(len(keys)):
            for i2 in range(i1 + 1, len(keys)):
                if (keys[i1] in Lp.cell_out or keys[i2] in Lp.cell_out) and not _sites_ok(*desc[keys[i1]], *desc[keys[i2]]):
                    raise TparAbort(0, "two per-trip cell addresses may name one cell")
            if any(desc[keys[i1]][2] <= a <= desc[keys[i1]][3] for a in cell_addr.values()):
                raise TparAbort(0, "a per-trip cell runs over a mem[] cell")

    with np.errstate(all="ignore"):
        for n in self.uniform:
            if n.kind in ("const", "hold"):
                continue
            if n.kind == "inv":
                val[n.i] = np.float64(inv_value(n.name))
            else:
                val[n.i] = _np_op(n.op, [V(a) for a in n.args])
        for gn in self.guards:
            if _truthy(np.float64(V(gn))):
                raise TparAbort(0, "a rare-event branch the lowering left out is due")
        cell_addr = {name: int(V(a)) for name, a in self.cells.items()}
        if len(set(cell_addr.values())) != len(cell_addr) or any(a >= len(memv) for a in cell_addr.values()):
            raise TparAbort(0, "mem[] cells alias each other or lie past the arena")
        for Lp in self.loops:
            if Lp.cells:
                address_pass(Lp)
        carry = {name: np.float64(inv_value(name)) for name in self.st}
        hcarry = {name: np.float64(inv_value(name)) for name in self.holdvars}
        y = np.zeros_like(x)
        final_vals: Dict[str, float] = {}
        f0, cut_to = 0, None
        self.fb_cuts = 0
        while f0 < frames:
            tn = min(WAVE, frames - f0) if cut_to is None else cut_to
            cut_to = None
            last = tn - 1
            for n in self.inputs:
                col = np.zeros(WAVE)
                col[:tn] = x[int(n.val), f0:f0 + tn].astype(np.float64)
                val[n.i] = col
            sites: Dict[int, dict] = {}
            try:
                run_items(top, carry, f0, tn, sites)
            except _Recut as rc:                        # the same chunk again, ending before the frame that reads its own writes
                cut_to = rc.e
                self.fb_cuts += 1
                continue
            except TparAbort:
                for st_ in reversed(self.stores):          # what this chunk's early writes replaced
                    if st_.mode == "early" and st_.j in sites and "old" in sites[st_.j]:
                        memv[sites[st_.j]["A"][:tn]] = sites[st_.j]["old"]
                raise
            for st_ in self.stores:                    # the chunk's writes land after all of its reads are resolved
                si = sites[st_.j]
                if st_.mode == "sparse":
                    Vv = vec(st_.value)
                    for t in np.flatnonzero(si["on"]):
                        memv[si["A"][t]] = Vv[t]
                        mem_high[0] = max(mem_high[0], int(si["A"][t]) + 1)
                    continue
                if not si["live"]:
                    continue
                if st_.mode == "late":
                    memv[si["A"][:tn]] = vec(st_.value)[:tn]
                mem_high[0] = max(mem_high[0], int(si["A"][:tn].max()) + 1)
            for ch in range(self.nch):
                y[ch, f0:f0 + tn] = vec(self.spl_out[ch])[:tn].astype(np.float32)
            for name in self.st:
                carry[name] = np.float64(vec(self.outs[name])[last])
            for name in self.holdvars:
                v = vec(self.outs[name])
                on = ~_is_hold(v)
                on[tn:] = False
                if on.any():
                    hcarry[name] = np.float64(v[np.flatnonzero(on)[-1]])
            if stream is not None:
                stream.end_chunk(int(carry[RNG_INDEX]))
            if f0 + tn >= frames:
                for name, o in list(self.outs.items()) + [(f"spl{ch}", self.spl_out[ch]) for ch in range(self.nch)]:
                    final_vals[name] = float(hcarry[name]) if name in hcarry else float(vec(o)[last])
            f0 += tn
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
                mem_high[0] = max(mem_high[0], cell_addr[name] + 1)
        elif k is not None:
            spl_after[k] = v
        else:
            vars_after[name] = v
    self.mem_after, self.mem_high_after = memv, mem_high[0]
    return y, vars_after, spl_after


Plan.simulate = _simulate
