"""One frame of @sample as a DAG: nodes, uniform loops, the passes over the syntax tree (rare-event guards, events, variables
whose incoming value is observable) and the walk that builds the graph in the reference emitter's order of evaluation."""
from __future__ import annotations

import math
import os
from typing import Dict, List, Optional, Tuple

import numpy as np

from .. import syntax as S
from ..emit import NOOP_CALLS, PURE_MATH1, PURE_MATH2, c_double
from ..program import Program, is_slider_name, is_spl_name

from .numeric import *



class Unsupported(Exception):
    """@sample uses a construct the time-parallel lowering does not handle; the leaf keeps the generic kernel only."""


class N:
    __slots__ = ("i", "kind", "op", "args", "val", "name", "uniform", "su", "extra", "loop", "ctx", "fb", "pred")

    def __init__(self, i, kind, op=None, args=(), val=None, name=None):
        self.i, self.kind, self.op, self.args, self.val, self.name = i, kind, op, tuple(args), val, name
        self.uniform = False
        self.su = False          # wave-uniform, known while walking: built from constants, variables @sample never assigns and
        #                          uniform loop counters
        self.pred = None         # ld: the path condition it stands under (if-conversion reads anyway; its run-time checks only count there)
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


GUARD_MIN_BODY = 48      # syntax nodes (through the functions it calls) from which a guard's body is worth keeping out of the frame


def _ast_size(prog: Program, x, seen=None, cap: int = 100000) -> int:
    """Syntax nodes of x, through the user functions it calls (each counted once)."""
    seen = set() if seen is None else seen
    n, todo = 0, [x]
    while todo and n < cap:
        y = todo.pop()
        n += 1
        if isinstance(y, S.Call) and y.fn in prog.fns and y.fn not in seen:
            seen.add(y.fn)
            todo.append(prog.fns[y.fn].body)
        todo.extend(S.children(y))
    return n


def split_guards(prog: Program):
    """Top-level statements of @sample of the form `G ? ( ... )` where G reads only variables that nothing else in @sample
    writes and the body assigns one of them -- the "rebuild when the sample rate changed / a table is dirty" idiom: running the
    body clears the condition. The lowering takes G to be false (the statements are dropped, so everything they assign stays
    an invariant); the kernel evaluates every G at the start of each block and hands a launch whose G holds to the serial code.
    Returns (remaining statements, guard conditions)."""
    items = list(prog.sections.get("sample", []))
    if len(items) == 1 and isinstance(items[0], S.Seq):
        items = list(items[0].items)
    # (a LIGHT body -- a handful of assignments, `block_event_arm ? ( latch the block's MIDI flags; block_event_arm = 0 )`, which
    #  @block re-arms every block -- is cheaper as an ordinary conditional of the frame than as a serial frame per block:
    #  Texture spent 22 % of its kernel in those; only bodies that are heavy or large stay out of the frame)
    cand = {k for k, st in enumerate(items)
            if isinstance(st, (S.Cond, S.If)) and (st.els is None or isinstance(st.els, S.Num)) and st.then is not None
            and _pure_scalar(prog, st.cond)
            and (_assigned_names(prog, [st.then]) & _read_names(prog, [st.cond]))
            and (_heavy(prog, st.then) or _ast_size(prog, st.then) > GUARD_MIN_BODY)}
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


def split_events(prog: Program, stmts, keep=frozenset(), origin=None, cache=None, lower_modes=True):
    """Statements `C ? ( ... )` (no else) of @sample -- at the top or inside other conditionals, not inside loops or functions --
    whose body is heavy (_heavy) and whose condition is a plain expression: the "every hop: run the FFT" / "buffer full: convolve
    a block" idiom. The lowering replaces each by a marker that keeps the condition; the kernel evaluates the conditions first in
    every chunk, lets the frames before the first one that holds take the parallel path, runs that one frame with the serial
    section code (zt_frame) and starts over behind it. Returns (rewritten statements, bodies dropped)."""
    dropped = []
    wsyn = _assigned_names(prog, stmts)

    def mode_switch(cond) -> bool:
        """The condition reads nothing @sample assigns (and no audio): constant over a block -- a mode switch (`tex_loaded > 0`), not
        a rare event. While it holds it holds in every frame, so as an event it would hand the whole block to the serial code;
        lowered, its body runs time-parallel, and a chunk in which it does not hold skips the loops it guards (emit_loop). What
        the lowering cannot take in such a body still becomes an event, further in (FrameGraph._or_event)."""
        return not any(nm in wsyn or is_spl_name(nm) is not None for nm in _read_names(prog, [cond]))

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
                    and _heavy(prog, x.then) and id(x) not in keep and not (lower_modes and mode_switch(x.cond))):
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

    key = (tuple(id(st) for st in stmts), frozenset(keep), lower_modes)
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
        ld.pred = self.pred
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



class _Replan(Exception):
    """The set of statements that run as events changed: build the plan again with it."""


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


__all__ = [_n for _n in dir() if not _n.startswith("__")]
