"""numpy restatement of the staged algorithm, one array element per lane: pins the analysis -- classification, coefficients,
carries, partial chunks, loops, delay lines, feedback cuts -- on the CPU (tests/test_tpar.py)."""
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
from .plan import *

# ----------------------------------------------------------------------------------------------------------------------
# 4. numpy restatement of the staged algorithm (tests)
# ----------------------------------------------------------------------------------------------------------------------
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
    # Plans with events (or guards inside loops) run the frame an event falls on with the script's own section code -- device only.
    # The restatement still covers them as long as no event is due: the conditions are evaluated where the kernel evaluates
    # them, and the first one that holds ends the simulation with NotImplementedError.
    import sys
    _pkg = sys.modules[__package__]                # (the package's tunables as they stand now: tests set them on the package)
    SPEC_MAX, SPEC_TOL = _pkg.SPEC_MAX, _pkg.SPEC_TOL
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
                on_ = (_truthy(vec(n.pred)) if n.pred is not None else np.ones(WAVE, dtype=bool)) & (lane < tn)
                out = np.where(B < len(memv), memv[np.minimum(B, len(memv) - 1)], 0.0)
                best = np.full(WAVE, -1)
                for st_ in self.stores:
                    si = sites[st_.j]
                    if st_.mode == "sparse":
                        if np.any(on_ & (B >= si["lo"]) & (B <= si["hi"])):
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
                        if np.any(on_ & (tw >= 0)):
                            raise TparAbort(f0, "a delay-line read falls into another buffer's freshly written span")
                        continue
                    if st_.mode == "early":
                        late = (tw > lane) | ((tw == lane) & (not st_.seq < n.val))
                        if np.any(on_ & late):
                            raise TparAbort(f0, "a gather reads a cell that a later frame of the chunk has already overwritten")
                        continue
                    if st_.j in n.fb:
                        if np.any(on_ & (tw >= 0) & ((tw < lane) | ((tw == lane) & (st_.seq < n.val)))):
                            raise TparAbort(f0, "a feedback read would need a value of its own chunk (the cut should have ended it)")
                        continue
                    vis = (tw >= 0) & ((tw < lane) | ((tw == lane) & (st_.seq < n.val))) & (tw >= best)
                    Vv = vec(st_.value)
                    out = np.where(vis, Vv[np.clip(tw, 0, WAVE - 1)], out)
                    best = np.where(vis, tw, best)
                if any(np.any(on_ & (B == a)) for a in cell_addr.values()):
                    raise TparAbort(f0, "a delay-line read hits a mem[] cell")
                if np.any(on_ & (B >= wbox[0]) & (B <= wbox[1])):
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
                for ev_ in self.events:
                    if np.any(_truthy(vec(ev_))[:tn]):
                        raise NotImplementedError(f"an event is due in the chunk at frame {f0}: that frame runs the script's own section code (device only)")
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
                        out_ = []
                        for c, gn_ in zip(comp.conds, comp.gnodes):
                            cv_ = np.broadcast_to(loc[c.i] if c.i in loc else V(c), (WAVE,))
                            out_.append(np.array(cv_, dtype=np.float64) if gn_.op == "num" else _truthy(cv_))
                        return out_

                    prev = [np.full(WAVE, carry[nm]) for nm in comp.names]
                    gs = conds_from(prev)
                    converged, iters, still = False, 0, 0
                    while iters < SPEC_MAX:
                        iters += 1
                        loc = {gn.i: (gs[k] if gn.op == "num" else np.where(gs[k], 1.0, 0.0)) for k, gn in enumerate(comp.gnodes)}
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
        # (the cells the trips use: one only a guard reads is loaded by the guard's own evaluation, like in the kernel's pass)
        pass_cells = {key: a for key, a in Lp.cells.items() if (key in Lp.cin and Lp.cin[key].i in live_ids) or key in Lp.cell_out}
        seqs: Dict[str, List[int]] = {key: [] for key in pass_cells}
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
            gmemo: Dict[int, np.float64] = {}

            def gv(n: N):                       # a guard's nodes inside the loop: evaluated here only (as in the kernel's pass)
                if n.kind == "const":
                    return np.float64(n.val)
                if not _in_subtree(n, Lp) or n.kind == "phi":
                    return np.float64(uni(n, "guard operand"))
                if n.i in gmemo:
                    return gmemo[n.i]
                if n.kind == "lcin":
                    a_ = int(gv(self.g.lcell_addr[n.name]))
                    r_ = np.float64(memv[a_] if 0 <= a_ < mcap else 0.0)
                else:
                    r_ = np.float64(_np_op(n.op, [gv(a_) for a_ in n.args]))
                gmemo[n.i] = r_
                return r_

            for gc in Lp.guards:
                if _truthy(gv(gc)):
                    raise NotImplementedError(f"a statement of loop {Lp.id} that runs as an event is due (trip {k}): device only")
            for key, a in pass_cells.items():
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
        wr_cells = {nm: a for nm, a in cell_addr.items() if nm in self.outs}
        if (any(a >= len(memv) for a in wr_cells.values())
                or any(a == a2 for nm, a in wr_cells.items() for nm2, a2 in cell_addr.items() if nm2 != nm)):
            raise TparAbort(0, "a mem[] cell that is stored to aliases another or lies past the arena")
        for Lp in self.loops:
            if Lp.cells or Lp.guards:
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


__all__ = [_n for _n in dir() if not _n.startswith("__")]
