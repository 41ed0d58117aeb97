"""HIP text of one plan: the kernel `zab_<leaf>_tpar`, its section functions and serial tail (csrc/zart_tpar.h holds the
wavefront primitives)."""
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
        L.append("  za_fft_tables_once(st);      // (a host may reach this launch without a prepare of this module: zab_generic.hip.h)")
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
                # (cells the frame only READS may share an address -- x[0] and x[n - 1] with n = 1 -- and may lie past the arena,
                #  where a read yields 0: only a cell that is stored to must be alone and inside)
                wr = {p.cells[nm].i for nm in p.cells if nm in p.outs}
                clash = " || ".join([f"ca{a.i} >= mcap" for a in ca if a.i in wr]
                                    + [f"ca{a.i} == ca{b_.i}" for i_, a in enumerate(ca) for b_ in ca[i_ + 1:] if a.i in wr or b_.i in wr]) or "false"
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
    def serial_loop(self, reg: Region, comps: List[Component], ind: str, rest_test: bool = False):
        """64 uniform steps; leaves the state before each frame in k<st> of that frame's lane.
        rest_test: first ask, frame-parallel, whether ANY frame of the chunk would move a state that sits at its carried value.
        If none does, the states keep those values through the whole chunk (induction over the frames: frame 0 sees the carried
        values and leaves them, so frame 1 sees them too, ...) and the 64 steps are skipped -- an idle voice of a voice loop, a
        smoother that has arrived. Bit patterns are compared, so -0 / +0 and NaN count as moves."""
        L, ref = self.L, self.ref
        for c in comps:
            for nm in c.names:
                s = reg.st[nm].i
                L.append(f"{ind}double y{s} = {self.carry(reg, nm)}, k{s} = {self.carry(reg, nm)};")
        if rest_test and not os.environ.get("ZA_TPAR_NO_REST_TEST"):
            tag = reg.st[comps[0].names[0]].i
            L.append(f"{ind}bool zrest{tag};")
            L.append(f"{ind}{{")
            moves = []
            for c in comps:
                mem = {m.i for m in c.members}

                def pref(x: N, mem=mem) -> str:
                    if x.kind in ("st", "lcin") and x.i in mem:
                        return f"y{x.i}"
                    if x.i in mem:
                        return f"u{x.i}"
                    return ref(x)

                for m in c.members:
                    if m.kind not in ("st", "lcin"):
                        L.append(f"{ind}  const double u{m.i} = {_expr(m.op, [pref(x) for x in m.args])};")
                for nm in c.names:
                    moves.append(f"__double_as_longlong({pref(reg.outs[nm])}) != __double_as_longlong(y{reg.st[nm].i})")
            L.append(f"{ind}  zrest{tag} = __ballot(valid && ({' || '.join(moves)})) == 0ull;")
            L.append(f"{ind}}}")
            L.append(f"{ind}if (!zrest{tag})")
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
        # (a read under a condition is made anyway -- if-conversion -- but where its condition is false its address means nothing:
        #  the checks only count for the frames that take the branch)
        gate = f"za_truthy({ref(n.pred)}) && " if n.pred is not None else ""
        if forward and any(s.mode == "late" and ",".join(map(str, s.region)) == n.name and s.j not in n.fb for s in p.stores):
            L.append(f"{ind}int best = -1;")
        for st_ in p.stores:
            j = st_.j
            if st_.mode == "sparse":
                L.append(f"{ind}zt_badl |= {gate}({B} >= zlo{j} && {B} <= zhi{j});")
                continue
            on = f"zsu{j} && " if st_.pred is not None else ""
            same = ",".join(map(str, st_.region)) == n.name
            if not same:
                L.append(f"{ind}zt_badl |= {gate}({on}((uint64_t)({B} - zq0_{j}) < (uint64_t)zqk_{j} || (uint64_t)({B} - zq1_{j}) < (uint64_t)(tn - zqk_{j})));")
                continue
            L.append(f"{ind}{{ int tw = -1; const int64_t d0 = {B} - zq0_{j}, d1 = {B} - zq1_{j};")
            L.append(f"{ind}  if ((uint64_t)d0 < (uint64_t)zqk_{j}) tw = (int)d0;")
            L.append(f"{ind}  if ((uint64_t)d1 < (uint64_t)(tn - zqk_{j})) tw = zqk_{j} + (int)d1;")
            if st_.mode == "early":
                # memory already holds this chunk's values: right for frames at or before this one, wrong for later ones
                before = "false" if st_.seq < n.val else "true"
                L.append(f"{ind}  zt_badl |= {gate}{on}(tw > lane || (tw == lane && {before})); }}")
            elif st_.j in n.fb:
                # (the chunk was cut before the first frame that reads what an earlier one of its frames writes: nothing to see)
                before = "true" if st_.seq < n.val else "false"
                L.append(f"{ind}  zt_badl |= {gate}{on}(tw >= 0 && (tw < lane || (tw == lane && {before}))); }}")
            else:
                assert forward
                before = "true" if st_.seq < n.val else "false"
                L.append(f"{ind}  const bool vis = {on}valid && tw >= 0 && (tw < lane || (tw == lane && {before})) && tw >= best;")
                L.append(f"{ind}  if (__ballot(vis)) {{ const double fw = zt_bperm({ref(st_.value)}, tw); v = vis ? fw : v; best = vis ? tw : best; }} }}")
        if self.cell_addrs:
            L.append(f"{ind}zt_badl |= {gate}({B} >= cmin && {B} <= cmax);")
        if self.has_wbox:
            L.append(f"{ind}zt_badl |= {gate}({B} >= wmin && {B} <= wmax);     // (a per-trip cell some loop stores to: its value lives in LDS)")

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
                self.serial_loop(reg, it[1], ind, rest_test=True)
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
        elif len(c.names) > 2:
            d = len(c.names)
            sid = [reg.st[nm].i for nm in c.names]
            rows = ", ".join("{" + ", ".join(ref(c.A[r][k]) for k in range(d)) + "}" for r in range(d))
            L.append(f"{ind}ZtMapN<{d}> sm{sid[0]} = {{{{{rows}}}, {{{', '.join(ref(c.b[r]) for r in range(d))}}}}};   // {', '.join(c.names)}: {d} coupled affine states")
            L.append(f"{ind}zt_scanN<{d}>(sm{sid[0]});")
            L.append(f"{ind}const double sc{sid[0]}[{d}] = {{{', '.join(cn(nm) for nm in c.names)}}};")
            for r, nm in enumerate(c.names):
                L.append(f"{ind}const double n{sid[r]} = zt_shift1(zt_mapN_apply<{d}>(sm{sid[0]}, {r}, sc{sid[0]}), {cn(nm)});")
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
                L.append(f"{ind}{'double' if gn.op == 'num' else 'bool'} {gname[gn.i]};")

        def xref(x: N, loc: Dict[int, str]) -> str:
            if x.i in loc:
                return loc[x.i]
            if x.kind == "guess":
                return gname[x.i] if x.op == "num" else f"({gname[x.i]} ? 1.0 : 0.0)"
            return ref(x)

        def slice_eval(c: Component, ind2: str, out_prefix: str):
            loc = {reg.st[nm].i: f"s{reg.st[nm].i}" for nm in c.names}
            for m in c.slice:
                loc[m.i] = f"v{m.i}"
                L.append(f"{ind2}const double v{m.i} = {_expr(m.op, [xref(x, loc) for x in m.args])};")
            for k, (cnd, gn) in enumerate(zip(c.conds, c.gnodes)):
                if gn.op == "num":
                    L.append(f"{ind2}{out_prefix.replace('bool', 'double')}{gname[gn.i]} = {xref(cnd, loc)};")
                else:
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
            elif len(c.names) > 2:
                d = len(c.names)
                sid = [reg.st[nm].i for nm in c.names]
                rows = ", ".join("{" + ", ".join(xref(c.A[r][k], loc) for k in range(d)) + "}" for r in range(d))
                L.append(f"{ind}  ZtMapN<{d}> sm{sid[0]} = {{{{{rows}}}, {{{', '.join(xref(c.b[r], loc) for r in range(d))}}}}};")
                L.append(f"{ind}  zt_scanN<{d}>(sm{sid[0]});")
                L.append(f"{ind}  const double sc{sid[0]}[{d}] = {{{', '.join(cn(nm) for nm in c.names)}}};")
                for r, nm in enumerate(c.names):
                    L.append(f"{ind}  s{sid[r]} = zt_shift1(zt_mapN_apply<{d}>(sm{sid[0]}, {r}, sc{sid[0]}), {cn(nm)});")
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
        if any(it[1].kind == "ld" and it[1].pred is not None and _in_subtree(it[1].pred, Lp) for it in reg.items):
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
        if any(n.kind == "ld" and n.pred is not None and _in_subtree(n.pred, Lp) for n in nodes):
            return None                           # (a read under a condition of the trip: its checks name that condition)
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



__all__ = [_n for _n in dir() if not _n.startswith("__")]
