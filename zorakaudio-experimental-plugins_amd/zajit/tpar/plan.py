"""Analysis of the frame graph: liveness, regions, recurrences (strongly connected components) and their affine forms, delay
lines and feedback, the schedule of one chunk -- and the loop that settles which statements run as events."""
from __future__ import annotations

import math
import os
from typing import Dict, List, Optional, Tuple

import numpy as np

from .. import syntax as S
from ..emit import NOOP_CALLS, PURE_MATH1, PURE_MATH2, c_double
from ..program import Program, is_slider_name, is_spl_name

from .numeric import *
from .graph import *

# ----------------------------------------------------------------------------------------------------------------------
# 2. analysis: recurrences, affine forms, schedule
# ----------------------------------------------------------------------------------------------------------------------
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
MAX_COUPLED_STATES = 4     # states of one affine / switched recurrence (zt_scan1, zt_scan2, zt_scanN<3>, zt_scanN<4>)
MODE_LOWERING_MAX_NODES = 1500   # ... and beyond this the bodies of block-constant conditions stay out of the frame (events)


class _TooLarge(Unsupported):
    pass
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


def build_plan(prog: Program, nch: int) -> Plan:
    """Raises Unsupported when the leaf cannot take the time-parallel kernel."""
    if not prog.has("sample") or nch <= 0:
        raise Unsupported("no audio @sample")
    if os.environ.get("ZA_TPAR_NO_BLOCK") and prog.has("block"):
        raise Unsupported("@block present")
    event_ids: set = set()
    no_event: set = set()
    why: Dict[int, str] = {}
    why["#lower_modes"] = True
    for _round in range(24):
        try:
            plan = _build_plan(prog, nch, event_ids, no_event, why)
        except _Replan:
            continue
        except _TooLarge:
            if not why["#lower_modes"]:
                raise
            # with the bodies of its mode switches lowered the frame is too much for one kernel: they run as events, as
            # split_events takes every other heavy branch (a block in which one holds then goes to the serial code)
            event_ids.clear(); no_event.clear()
            why.clear(); why["#lower_modes"] = False
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
        stmts, ev_bodies = split_events(prog, stmts, keep=frozenset(no_event), origin=origin, cache=why.setdefault("#split", {}),
                                        lower_modes=bool(why.get("#lower_modes", True)))
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
    if (not written and not plan.stores and not plan.events and not plan.guards and not g.loops
            and all(o.kind == "in" and int(o.val) == ch for ch, o in enumerate(plan.spl_out))):
        # every channel goes out as it came in and nothing else happens per frame, rare heavy branches included (a guard's body runs with
        # the wavefront's lanes cooperating -- the FFT harness: 304 against 355 us per round trip): such a leaf is its @block (message-bus / gmem
        # bookkeeping), which the lane-per-instance kernel runs with the state in registers from block to block (3DPannerManager,
        # 256 instances x 48 000 frames: 1124 ms against 1570 ms on a time-parallel kernel with nothing to parallelise)
        raise Unsupported("@sample is empty: nothing to run time-parallel")

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
        if ld.pred is not None:
            ex.append(ld.pred)
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
        raise _TooLarge(f"the frame is too large for one kernel ({len(live)} nodes)")
    if len(live) > MODE_LOWERING_MAX_NODES and why.get("#lower_modes", True) and not os.environ.get("ZA_TPAR_ANY_SIZE"):
        raise _TooLarge(f"{len(live)} nodes with the mode switches' bodies in the frame")
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
                if ld.pred is not None:
                    keep.append(ld.pred)
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
    guards_ = _uniform_guards(plan, live, all_regions) if want != "0" else {}
    plan.node_guard = guards_ if (want == "1" or (want is None and _wants_uniform_guards(plan, guards_))) else {}

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


def _wants_uniform_guards(plan: "Plan", guards_: Dict[int, tuple]) -> bool:
    """Whether the arms of block-constant conditions run under wave-uniform branches. The branches pin values the straight-line
    form could sink to their uses (ERBTilt: 176 -> 346 registers), so they must buy something: a script with ONE mode switch has
    its guarded nodes on the arm its settings take, whatever they are (measured without gain: BedRock, DPT, ATTACK, ERBTilt,
    PsychoConvolver); a script with many independent switches -- per-band modes -- leaves a good part of its arms idle under any
    setting (TSEQ: 23 conditions over 39 % of its lane-parallel nodes, 156 -> 132 ms). The rule: at least four distinct
    conditions guarding at least a quarter of the frame's lane-parallel nodes."""
    par = sum(1 for it in plan.top.items if it[0] == "par" and it[1].kind == "op")
    conds = len({c.i for c, _ in guards_.values()})
    return conds >= 4 and len(guards_) * 4 >= max(1, par)


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
    if d > MAX_COUPLED_STATES:
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

        def guess_for(cond: N, numeric: bool = False) -> N:
            for k, x in enumerate(conds):
                if x is cond:
                    return gnodes[k]
            conds.append(cond)
            # (op "num": the placeholder stands for the VALUE of `cond` -- floor(...) of a state -- not for its truth)
            gn = g.mk("guess", op="num" if numeric else None, name=f"{ci}", val=len(gnodes))
            gn.loop = reg.loop
            gnodes.append(gn)
            return gn

        def pick(cnd: N, a, b):
            co = {k: g.sel(cnd, a[0].get(k, g.ZERO), b[0].get(k, g.ZERO)) for k in sorted(set(a[0]) | set(b[0]))}
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
                        co = {k: f(a[0].get(k, g.ZERO), b[0].get(k, g.ZERO)) for k in sorted(set(a[0]) | set(b[0]))}
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
                elif op in ("floor", "ceil") and allow_guess:
                    # a wrap written with floor (`ph -= floor(ph)`): piecewise constant in the states -- with its VALUE fixed per
                    # frame the recurrence is affine; the value the scanned states imply must reproduce the guess, like a switch
                    a = aff(n.args[0])
                    if a:
                        r = ({}, guess_for(n, numeric=True))
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



__all__ = [_n for _n in dir() if not _n.startswith("__")]
