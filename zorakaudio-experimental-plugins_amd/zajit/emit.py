"""Program -> C++ (one translation-unit fragment that both hipcc and g++ accept).

Every JSFX construct becomes one C++ *expression* of type double (GNU statement-expressions give sequencing),
executed against `ZaState<NV>& s` from csrc/zart.h. Evaluation order is the reference emitter's, made explicit
wherever C++ leaves it unspecified:
  binary operands left then right ................. dsp_jsfx_aot.py:4329-4334
  assignment: value, then target address .......... dsp_jsfx_aot.py:4386-4492
  a[b]: base then index ........................... dsp_jsfx_aot.py:4062-4073
  call arguments left to right .................... dsp_jsfx_aot.py:5216-5220
  builtin dispatch order (specials, no-op gfx_/str, user fns, math) ... dsp_jsfx_aot.py:4495-5566
  sequences / if / while as values ................ dsp_jsfx_aot.py:5568-5588
"""
from __future__ import annotations

import math
import os
import re
from typing import Dict, List, Optional, Set

from . import syntax as S
from .program import Program, is_slider_name, is_spl_name

PURE_MATH1 = {"sin": "sin", "cos": "cos", "sqrt": "sqrt", "fabs": "fabs", "floor": "floor", "ceil": "ceil",
              "asin": "asin", "acos": "acos", "atan": "atan", "exp": "exp", "log": "log", "tan": "tan",
              "log10": "log10"}
PURE_MATH2 = {"pow": "pow", "atan2": "atan2"}
NOOP_CALLS = {"sprintf", "printf", "strcpy", "strcat", "strcmp", "strlen", "str_getchar", "str_setchar",
              "str_insert", "str_delete", "str_mid", "strncpy", "file_read", "file_write", "file_string"}
# builtins that need a host (MIDI, files, strings, messaging): reaching one on the device latches ZA_ERR_UNSUPPORTED.
HOST_ONLY = {"msg_peer_name", "msg_peer_uid", "sample_name", "sample_preview_read", "sample_preview_bins"}
# scalar message bus between the instances of one engine (csrc/zart_msg.h)
MSG_CALLS = {"msg_subscribe", "msg_unsubscribe", "msg_advertise", "msg_send", "msg_sendto", "msg_avail", "msg_kind", "msg_recv",
             "msg_length", "msg_dropped", "msg_clear", "msg_peer_count", "msg_peer_id", "msg_peer_caps", "msg_peer_alive",
             "msg_send_buf", "msg_sendto_buf", "msg_recv_buf"}
# file_*() over host-provided file slots (csrc/zart_file.h)
FILE_CALLS = {"file_open", "file_open_multi", "file_close", "file_rewind", "file_seek", "file_avail", "file_text", "file_mem",
              "file_multi_count", "file_multi_select", "file_var", "file_riff"}
# Identity / host-context builtins with a fixed answer in a batch engine (reference: src/DspJsfxRuntimeBuiltins.cpp:67-140):
#   there is no DAW track, so the track-name queries answer "none" (0), as the reference does for an empty name;
#   an engine is one communication domain with one gmem segment, so comm_join() / instance_set_name() succeed (1);
#   strings are opaque handles on the device, so instance_uid() / instance_get_name() cannot fill one and return 0.
HOST_CONST = {"track_name": 0, "host_track_name": 0, "track_name_available": 0, "host_track_name_available": 0,
              "track_name_seq": 0, "host_track_name_seq": 0, "comm_join": 1, "instance_set_name": 1, "instance_uid": 0,
              "instance_get_name": 0}
# MIDI: the batch engine has no MIDI ports. With an empty input queue the reference's midirecv*() return 0 and leave their
# outputs untouched (src/JSFXJuceProcessor.cpp:2239-2333); with no output queue midisend*() return 0 (:2335-2420).
MIDI_CALLS = {"midirecv", "midirecv_buf", "midirecv_str", "midisend", "midisend_buf", "midisend_str", "midisyx"}
GMEM_CALLS = {"gmem_attach", "gmem_attach_size", "gmem_size", "gmem_get", "gmem_put", "gmem_fill", "gmem_zero",
              "gmem_copy", "gmem_seq", "gmem_page"}
POOL_READ_CALLS = {"sample_pool_from_slot", "sample_pool_set_mode", "sample_pool_set_budget_mb", "sample_pool_commit",
                   "sample_pool_state", "sample_pool_selected", "sample_pool_loaded", "sample_pool_failed",
                   "sample_pool_ram_mb", "sample_pool_generation", "sample_get", "sample_len", "sample_channels",
                   "sample_srate", "sample_peak", "sample_rms", "sample_read", "sample_read_interp", "sample_read2",
                   "sample_read2_interp", "sample_export_mem", "sample_export_mem2"}
FFT_CALLS = {"fft", "ifft", "fft_real", "ifft_real", "fft_permute", "fft_ipermute"}


class EmitError(ValueError):
    pass


def c_double(v: float) -> str:
    if math.isnan(v):
        return "(0.0/0.0)"
    if math.isinf(v):
        return "(1.0/0.0)" if v > 0 else "(-1.0/0.0)"
    r = repr(float(v))
    if "e" in r or "E" in r or "." in r or "n" in r:
        return r if ("." in r or "e" in r or "E" in r) else r + ".0"
    return r + ".0"


def c_ident(name: str) -> str:
    return "".join(ch if (ch.isalnum() or ch == "_") else f"_x{ord(ch):02X}_" for ch in name)


class Emitter:
    UNROLL_MAX_NODES = 120           # loop bodies up to this many AST nodes (callees included) are written out four times
    UNROLL_MAX_LOOPS = 12            # ... at most this many times per program (everything is inlined: code size)

    def __init__(self, prog: Program):
        self.p = prog
        self.unroll = not os.environ.get("ZA_NO_UNROLL")
        self.unrolled = 0
        self.cur_sec = "sample"
        self.redirect: Dict[str, str] = {}
        self.coop = not os.environ.get("ZA_NO_COOP")
        self.defer = None                          # map loops: arena stores of the trip being emitted go to named slots (_map_loop)
        self._fn_nodes: Dict[str, int] = {}
        self.tmp = 0
        self.strings: List[str] = []
        self.scope: List[Set[str]] = []          # parameter names of the function being emitted
        self.features: Set[str] = set()
        self.used_spl: Set[int] = set()
        self.used_sl: Set[int] = set()
        self.dyn_spl = False
        self.dyn_sl = False
        self._purity: Dict[int, bool] = {}

    # -- helpers ---------------------------------------------------------------------------
    def t(self, stem="t") -> str:
        self.tmp += 1
        return f"{stem}{self.tmp}_"

    def intern(self, s: str) -> int:
        if s not in self.strings:
            self.strings.append(s)
        return self.strings.index(s)

    def is_const(self, n) -> bool:
        return isinstance(n, (S.Num, S.Str)) or (isinstance(n, S.Var) and n.name.startswith("$") and not self._is_param(n.name))

    def _is_param(self, name) -> bool:
        return bool(self.scope) and name in self.scope[-1]

    def pure(self, n) -> bool:
        """No writes, no state-dependent side effects: safe to evaluate in any order relative to other pure nodes."""
        k = id(n)
        if k in self._purity:
            return self._purity[k]
        if isinstance(n, (S.Num, S.Str, S.Var)):
            r = True
        elif isinstance(n, (S.Assign, S.Loop, S.While, S.If)):
            r = False
        elif isinstance(n, S.Call):
            fn = "fabs" if n.fn == "abs" else n.fn
            r = (fn not in self.p.fns and fn in (set(PURE_MATH1) | set(PURE_MATH2) | {"min", "max", "sqr", "sign", "invsqrt", "__memtop"})
                 and all(self.pure(a) for a in n.args))
        elif isinstance(n, S.Index):
            r = not self._is_gmem(n) and self.pure(n.base) and self.pure(n.index)
        else:
            r = all(self.pure(c) for c in S.children(n))
        self._purity[k] = r
        return r

    def ordered(self, nodes: List[S.Node]):
        """Return (prelude_statements, operand_strings) evaluating `nodes` strictly left to right."""
        exprs = [self.expr(n) for n in nodes]
        if len(nodes) <= 1 or all(self.pure(n) for n in nodes):
            return "", exprs
        pre, names = [], []
        for n, e in zip(nodes, exprs):
            if self.is_const(n):
                names.append(e)
            else:
                nm = self.t()
                pre.append(f"double {nm} = {e};")
                names.append(nm)
        return " ".join(pre), names

    def wrap(self, pre: str, value: str) -> str:
        return f"({{ {pre} {value}; }})" if pre else value

    @staticmethod
    def _is_gmem(n) -> bool:
        return isinstance(n, S.Index) and isinstance(n.base, S.Var) and n.base.name == "gmem"

    # -- lvalues / variables ------------------------------------------------------------------
    def var_ref(self, name: str) -> str:
        if self._is_param(name):
            return "p_" + c_ident(name)
        if name in self.redirect:              # accumulator of a loop shared by replica lanes (e_Loop): its partial sum
            return self.redirect[name]
        k = is_spl_name(name)
        if k is not None:
            if not 0 <= k < 64:
                raise EmitError(f"Invalid spl index: {name}")
            self.used_spl.add(k)
            return f"s.spl[{k}]"
        k = is_slider_name(name)
        if k is not None:
            if not 1 <= k <= 64:
                raise EmitError(f"Invalid slider index: {name}")
            self.used_sl.add(k - 1)
            return f"s.sl[{k - 1}]"
        if name == "srate":
            return "s.srate"
        if name == "samplesblock":
            return "s.samplesblock"
        if name == "midi_bus":
            return "s.midi_bus"
        if name == "ext_midi_bus":
            return "s.ext_midi_bus"
        if name in ("mem", "gmem"):
            raise EmitError(f"{name} has no address")
        if name not in self.p.vars:
            raise EmitError(f"Unknown variable {name!r} (not declared by analysis)")
        return f"s.v[{self.p.vars[name]}]"

    def var_value(self, name: str) -> str:
        if not self._is_param(name):
            if name == "mem":
                return "0.0"
            if name == "gmem":
                raise EmitError("gmem may only be used as gmem[index]")
            if name == "$pi":
                return c_double(math.pi)
            if name == "$phi":
                return c_double((1.0 + math.sqrt(5.0)) * 0.5)
            if name == "$e":
                return c_double(math.e)
            if name.startswith("$x") and len(name) > 2:
                try:
                    return c_double(float(int(name[2:], 16)))
                except ValueError:
                    pass
        return self.var_ref(name)

    # -- expressions --------------------------------------------------------------------------
    def expr(self, n) -> str:
        m = getattr(self, "e_" + type(n).__name__)
        return m(n)

    def e_Num(self, n):
        return c_double(n.value)

    def e_Str(self, n):
        return f"(ZA_STRING_BASE + {self.intern(n.value)}.0)"

    def e_Var(self, n):
        return self.var_value(n.name)

    def e_Index(self, n):
        if self._is_gmem(n):
            self.features.add("gmem")
            return f"za_gmem_load(s, {self.expr(n.index)})"
        pre, (b, i) = self.ordered([n.base, n.index])
        self.features.add("mem")
        return self.wrap(pre, f"za_ld(s, za_addr({b}, {i}))")

    def e_Unary(self, n):
        a = self.expr(n.a)
        if n.op == "+":
            return a
        if n.op == "-":
            return f"za_neg({a})"
        if n.op == "!":
            return f"za_not({a})"
        raise EmitError(f"Unsupported unary op {n.op}")

    _INFIX = {"+": "+", "-": "-", "*": "*", "/": "/"}
    _CMP = {"<": "<", "<=": "<=", ">": ">", ">=": ">=", "==": "=="}
    _FN2 = {"^": "pow", "|": "za_or", "&": "za_and", "<<": "za_shl", ">>": "za_shr", "%": "za_mod", "!=": "za_ne"}

    def binop(self, op: str, a: str, b: str) -> str:
        if op in self._INFIX:
            return f"({a} {self._INFIX[op]} {b})"
        if op in self._CMP:
            return f"za_b({a} {self._CMP[op]} {b})"
        if op in self._FN2:
            return f"{self._FN2[op]}({a}, {b})"
        if op == "~":
            return f"za_xor({a}, {b})"
        raise EmitError(f"Unsupported binary op {op}")

    def e_Binary(self, n):
        if n.op == "&&":
            return f"za_b(za_truthy({self.expr(n.l)}) && za_truthy({self.expr(n.r)}))"
        if n.op == "||":
            return f"za_b(za_truthy({self.expr(n.l)}) || za_truthy({self.expr(n.r)}))"
        pre, (a, b) = self.ordered([n.l, n.r])
        return self.wrap(pre, self.binop(n.op, a, b))

    def e_Cond(self, n):
        return f"(za_truthy({self.expr(n.cond)}) ? {self.expr(n.then)} : {self.expr(n.els)})"

    def e_Assign(self, n):
        tgt, op = n.target, n.op
        rhs = self.expr(n.value)
        bop = None if op == "=" else op[:-1]
        if isinstance(tgt, S.Var):
            if tgt.name in ("mem",):
                raise EmitError("Cannot assign to mem")
            ref = self.var_ref(tgt.name)
            if bop is None:
                return f"({ref} = {rhs})"
            r = self.t("r")
            return f"({{ double {r} = {rhs}; {ref} = {self.binop(bop, ref, r)}; }})"
        if self._is_gmem(tgt):
            self.features.add("gmem")
            r, i = self.t("r"), self.t("i")
            idx = self.expr(tgt.index)
            if bop is None:
                return f"({{ double {r} = {rhs}; double {i} = {idx}; za_gmem_store(s, {i}, {r}); }})"
            cur = f"za_gmem_load(s, {i})"
            return f"({{ double {r} = {rhs}; double {i} = {idx}; za_gmem_store(s, {i}, {self.binop(bop, cur, r)}); }})"
        if isinstance(tgt, S.Index):
            r, a = self.t("r"), self.t("a")
            pre, (b, i) = self.ordered([tgt.base, tgt.index])
            addr = f"{pre} int64_t {a} = za_addr({b}, {i});"
            self.features.add("mem")
            if self.defer is not None:             # a trip of a shared map loop: the store is made after the group's loads
                da, dv, do = self.t("da"), self.t("dv"), self.t("do")
                self.defer.append((da, dv, do))
                val = r if bop is None else self.binop(bop, f"za_ld(s, {a})", r)
                return f"({{ double {r} = {rhs}; {addr} {dv} = {val}; {da} = {a}; {do} = true; {dv}; }})"
            if bop is None:
                return f"({{ double {r} = {rhs}; {addr} za_st(s, {a}, {r}); }})"
            return f"({{ double {r} = {rhs}; {addr} za_st(s, {a}, {self.binop(bop, f'za_ld(s, {a})', r)}); }})"
        if isinstance(tgt, S.Call) and tgt.fn in ("slider", "spl") and len(tgt.args) == 1:
            arr, off = self._dyn(tgt.fn)
            r, i = self.t("r"), self.t("i")
            idx = self.expr(tgt.args[0])
            if bop is None:
                return f"({{ double {r} = {rhs}; double {i} = {idx}; za_dyn_st64({arr}, {i}, {off}, {r}); {r}; }})"
            o = self.t("o")
            cur = f"za_dyn_ld64({arr}, {i}, {off})"
            return (f"({{ double {r} = {rhs}; double {i} = {idx}; double {o} = {self.binop(bop, cur, r)}; "
                    f"za_dyn_st64({arr}, {i}, {off}, {o}); {o}; }})")
        raise EmitError("Invalid assignment target")

    def _dyn(self, which):
        if which == "slider":
            self.dyn_sl = True
            return "s.sl", 1
        self.dyn_spl = True
        return "s.spl", 0

    @staticmethod
    def _plain_arg(a) -> bool:
        """An argument that can be evaluated twice or once alike: numbers, variables and arithmetic on them."""
        if isinstance(a, (S.Num, S.Var)):
            return True
        if isinstance(a, S.Unary):
            return Emitter._plain_arg(a.a)
        if isinstance(a, S.Binary):
            return Emitter._plain_arg(a.l) and Emitter._plain_arg(a.r)
        return False

    def _fuse_fft_pairs(self, items):
        """fft(b, n); fft_permute(b, n)  ->  __fft_nat(b, n)   and   fft_ipermute(b, n); ifft(b, n)  ->  __ifft_nat(b, n)
        when both calls have the same side-effect-free arguments (zart_fft.h za_fft_nat: same bits, one pass fewer)."""
        out, i = [], 0
        while i < len(items):
            a, b = items[i], items[i + 1] if i + 1 < len(items) else None
            if (isinstance(a, S.Call) and isinstance(b, S.Call) and len(a.args) == 2 and len(b.args) == 2
                    and (a.fn, b.fn) in (("fft", "fft_permute"), ("fft_ipermute", "ifft"))
                    and repr(a.args) == repr(b.args) and all(self._plain_arg(x) for x in a.args)):
                out.append(S.Call("__fft_nat" if a.fn == "fft" else "__ifft_nat", a.args, line=a.line, col=a.col))
                i += 2
                continue
            # fft_real(b, n); fft_permute(b, m)  ->  __fft_real_nat(b, n, m);  fft_ipermute(b, m); ifft_real(b, n)  ->  __ifft_real_nat(b, n, m)
            # (the run time checks m = n / 2 and otherwise makes the two calls: za_fft_real_nat)
            if (isinstance(a, S.Call) and isinstance(b, S.Call) and len(a.args) == 2 and len(b.args) == 2
                    and (a.fn, b.fn) in (("fft_real", "fft_permute"), ("fft_ipermute", "ifft_real"))
                    and repr(a.args[0]) == repr(b.args[0]) and all(self._plain_arg(x) for x in a.args + b.args)):
                real, perm = (a, b) if a.fn == "fft_real" else (b, a)
                out.append(S.Call("__fft_real_nat" if a.fn == "fft_real" else "__ifft_real_nat",
                                  [real.args[0], real.args[1], perm.args[1]], line=a.line, col=a.col))
                i += 2
                continue
            out.append(a)
            i += 1
        return out

    def e_Seq(self, n):
        if not n.items:
            return "0.0"
        if len(n.items) > 1:
            n = S.Seq(self._fuse_fft_pairs(n.items), line=n.line, col=n.col)
        parts = []
        for it in n.items[:-1]:
            parts.append(self.stmt(it))
        last = n.items[-1]
        if isinstance(last, (S.If, S.While)):
            parts.append(self.stmt(last))
            parts.append("0.0;")
        else:
            parts.append(self.expr(last) + ";")
        return "({ " + " ".join(parts) + " })"

    def stmt(self, n) -> str:
        if isinstance(n, S.If):
            s = f"if (za_truthy({self.expr(n.cond)})) {{ {self.stmt(n.then)} }}"
            if n.els is not None:
                s += f" else {{ {self.stmt(n.els)} }}"
            return s
        if isinstance(n, S.While):
            c = self.t("w")
            shared = self._map_while(n)
            if shared:
                return shared
            return (f"{{ int64_t {c} = 0; while (za_truthy({self.expr(n.cond)})) {{ {self.stmt(n.body)} "
                    f"if (++{c} >= ZA_LOOP_CAP) {{ s.err |= ZA_ERR_LOOP_CAP; break; }} }} }}")
        return f"(void)({self.expr(n)});"

    def _map_while(self, n):
        """`while (v < X) ( body; v += <positive integer>; )` with X loop-invariant, as a map loop (_map_plan) when the trips
        turn out independent where it runs; the serial form otherwise."""
        hot = self.cur_sec in ("block", "sample")
        if not (hot and self.coop and not self.redirect) or self._nodes(n.body) > 400:
            return None
        if any(isinstance(x, (S.Loop, S.While)) for x in _walk(n.body)):
            return None
        cd = n.cond
        if not (isinstance(cd, S.Binary) and cd.op == "<" and isinstance(cd.l, S.Var)):
            return None
        items = n.body.items if isinstance(n.body, S.Seq) else [n.body]
        plan = self._map_plan(items, set(self.scope[-1]) if self.scope else set())
        if not plan:
            return None
        ind = plan[0]
        v = cd.l.name
        if ind.get(v, 0.0) <= 0.0:
            return None
        written = {x.target.name for it in items for x in _walk(it) if isinstance(x, S.Assign) and isinstance(x.target, S.Var)}
        for x in _walk(cd.r):                                # the bound must not move while the loop runs
            if isinstance(x, (S.Index, S.Assign, S.Call, S.Cond)) or (isinstance(x, S.Var) and x.name in written):
                return None
        c, w = self.t("n"), self.t("w")
        pro, okx, shared = self._map_loop(plan, c, n.body)
        serial = (f"{{ int64_t {w} = 0; while (za_truthy({self.expr(n.cond)})) {{ {self.stmt(n.body)} "
                  f"if (++{w} >= ZA_LOOP_CAP) {{ s.err |= ZA_ERR_LOOP_CAP; break; }} }} }}")
        count = f"za_map_trips({self.expr(cd.r)}, {self.var_ref(v)}, {c_double(ind[v])})"
        return (f"{{ bool z_ = false; if (ZA_COOP_ON(s)) {{ const int64_t {c} = {count}; "
                f"if ({c} >= za_coop_min(s) && {c} < ZA_LOOP_CAP) {{ {pro} z_ = {okx}; if (z_) {{ {shared} }} }} }} "
                f"if (!z_) {serial} }}")

    def e_If(self, n):
        return "({ " + self.stmt(n) + " 0.0; })"

    def e_While(self, n):
        return "({ " + self.stmt(n) + " 0.0; })"

    # -- loops shared by replica lanes -------------------------------------------------------
    _COOP_PURE = set(PURE_MATH1) | set(PURE_MATH2) | {"abs", "min", "max", "sqr", "sign", "invsqrt", "__memtop"}

    def _coop_plan(self, loop):
        """Is `loop` a read-only accumulation loop whose trips only depend on each other through `acc += expr` sums and
        `i += <integer literal>` counters? Returns (accumulators, {induction: step}) or None. Every other variable written in a
        trip (the callees' local()s included) must be assigned before it is read in that trip, unconditionally; no arena / gmem
        / spl / slider stores, no nested loops, only pure builtins."""
        items = loop.body.items if isinstance(loop.body, S.Seq) else [loop.body]
        acc, ind, written, bad = {}, {}, [], []
        seen_write = set()

        def fail(why):
            bad.append(why)

        def walk(node, params, cond, top):
            if bad:
                return
            if isinstance(node, (S.Num, S.Str)):
                return
            if isinstance(node, S.Var):
                nm = node.name
                if nm in params or nm.startswith("$"):
                    return
                reads.append(nm)
                if nm in temps_pending:
                    pass
                return
            if isinstance(node, S.Index):
                if self._is_gmem(node):
                    return fail("gmem")
                walk(node.base, params, cond, False); walk(node.index, params, cond, False)
                return
            if isinstance(node, (S.Loop, S.While, S.If, S.FuncDef)):
                return fail("control flow")
            if isinstance(node, S.Cond):
                walk(node.cond, params, cond, False)
                walk(node.then, params, True, False)
                if node.els is not None:
                    walk(node.els, params, True, False)
                return
            if isinstance(node, S.Binary):
                walk(node.l, params, cond, False)
                walk(node.r, params, cond or node.op in ("&&", "||"), False)
                return
            if isinstance(node, S.Unary):
                return walk(node.a, params, cond, False)
            if isinstance(node, S.Seq):
                for it in node.items:
                    walk(it, params, cond, False)
                return
            if isinstance(node, S.Call):
                for a in node.args:
                    walk(a, params, cond, False)
                if node.fn in self.p.fns:
                    f = self.p.fns[node.fn]
                    if len(stack) > 6 or node.fn in stack:
                        return fail("call depth")
                    stack.append(node.fn)
                    walk(f.body, set(f.params), cond, False)
                    stack.pop()
                    return
                if node.fn not in self._COOP_PURE:
                    return fail("builtin " + node.fn)
                return
            if isinstance(node, S.Assign):
                tgt = node.target
                if not isinstance(tgt, S.Var) or tgt.name in params or is_spl_name(tgt.name) is not None \
                        or is_slider_name(tgt.name) is not None or tgt.name not in self.p.vars:
                    return fail("store")
                nm = tgt.name
                if node.op in ("+=", "-=") and top and not cond:
                    if isinstance(node.value, S.Num) and float(node.value.value) == int(node.value.value) and nm not in acc:
                        if nm in ind or nm in seen_write:
                            return fail("induction written twice")
                        ind[nm] = (1.0 if node.op == "+=" else -1.0) * float(node.value.value)
                        seen_write.add(nm)
                        events.append(("ind", nm, len(reads)))
                        return
                    if nm not in ind and nm not in temps:
                        mark = len(reads)
                        walk(node.value, params, cond, False)
                        acc.setdefault(nm, []).append((mark, len(reads)))
                        seen_write.add(nm)
                        return
                    return fail("mixed update of " + nm)
                if node.op != "=" or cond:
                    return fail("conditional or compound write of " + nm)
                if nm in acc or nm in ind:
                    return fail("mixed update of " + nm)
                walk(node.value, params, cond, False)
                temps.setdefault(nm, len(reads))          # reads before this index saw the previous trip's value
                seen_write.add(nm)
                return
            return fail("node " + type(node).__name__)

        reads, events, temps, temps_pending, stack = [], [], {}, set(), []
        for it in items:
            walk(it, set(self.scope[-1]) if self.scope else set(), False, True)
        if bad or not acc:
            return None
        for nm, first_def in temps.items():                # def before use inside a trip
            if nm in reads[:first_def]:
                return None
        for nm in acc:                                       # accumulators are only ever accumulated
            if nm in reads or nm in temps or nm in ind:
                return None
        return sorted(acc), ind


    # -- elementwise ("map") loops shared by replica lanes ----------------------------------------
    class _Aff:
        """address = sum(coef[v] * v over induction variables) + const + sum(terms), terms = loop-invariant expressions"""
        __slots__ = ("coef", "const", "terms")

        def __init__(self, coef=None, const=0.0, terms=None):
            self.coef, self.const, self.terms = dict(coef or {}), float(const), list(terms or [])

        def add(self, o, sign=1.0):
            c = dict(self.coef)
            for k, v in o.coef.items():
                c[k] = c.get(k, 0.0) + sign * v
            t = self.terms + ([x for x in o.terms] if sign > 0 else [S.Unary("-", x) for x in o.terms])
            return Emitter._Aff(c, self.const + sign * o.const, t)

        def scale(self, f):
            return Emitter._Aff({k: v * f for k, v in self.coef.items()}, self.const * f,
                                [S.Binary("*", S.Num(float(f)), x) for x in self.terms])

    def _map_plan(self, items, scope_params):
        """Is one trip of a loop a map over mem[]: every arena access at an address that is affine in the loop's counters
        (`k += <integer>` as the trip's LAST statements) with loop-invariant rest, every other variable written in the trip
        assigned before it is read in that trip, no calls besides pure maths, no nested loops, no gmem / spl / slider
        stores? Returns (ind {name: step}, accesses [(kind, _Aff | (_Aff base, modulus ast))]) or None. Whether the trips
        are independent is decided where the loop runs (za_map_ok): from the accesses' first addresses and strides."""
        written = set()
        for it in items:
            for x in _walk(it):
                if isinstance(x, (S.Loop, S.While, S.FuncDef, S.Str)):
                    return None
                if isinstance(x, S.Call) and x.fn not in self._COOP_PURE:
                    return None
                if isinstance(x, S.Index) and self._is_gmem(x):
                    return None
                if isinstance(x, S.Assign):
                    t = x.target
                    if isinstance(t, S.Var):
                        if t.name in scope_params or is_spl_name(t.name) is not None or is_slider_name(t.name) is not None \
                                or t.name in ("mem", "gmem") or t.name.startswith("$"):
                            return None
                        written.add(t.name)
                    elif not isinstance(t, S.Index):
                        return None
        # counters: the trailing `v += n` / `v -= n` statements, each variable written nowhere else
        ind, tail = {}, len(items)
        while tail > 0:
            it = items[tail - 1]
            if (isinstance(it, S.Assign) and it.op in ("+=", "-=") and isinstance(it.target, S.Var) and isinstance(it.value, S.Num)
                    and float(it.value.value) == int(it.value.value) and int(it.value.value) != 0 and it.target.name not in ind):
                ind[it.target.name] = (1.0 if it.op == "+=" else -1.0) * float(it.value.value)
                tail -= 1
            else:
                break
        if not ind or tail == 0:
            return None
        body = items[:tail]
        for it in body:
            for x in _walk(it):
                if isinstance(x, S.Assign) and isinstance(x.target, S.Var) and x.target.name in ind:
                    return None
        temps = written - set(ind)

        def invariant(node):
            for x in _walk(node):
                if isinstance(x, S.Index) or (isinstance(x, S.Var) and (x.name in written)) or isinstance(x, (S.Assign, S.Call, S.Cond)):
                    return False
            return True

        env = {}                                            # temp -> _Aff while it holds an affine value

        def aff(node):
            if isinstance(node, S.Num):
                return self._Aff(const=float(node.value))
            if isinstance(node, S.Var):
                if node.name in ind:
                    return self._Aff({node.name: 1.0})
                if node.name in temps:
                    return env.get(node.name)
                if node.name == "mem":
                    return self._Aff()
                return self._Aff(terms=[node])
            if isinstance(node, S.Unary) and node.op in ("+", "-"):
                a = aff(node.a)
                return None if a is None else (a if node.op == "+" else a.scale(-1.0))
            if isinstance(node, S.Binary) and node.op in ("+", "-"):
                a, b = aff(node.l), aff(node.r)
                return None if a is None or b is None else a.add(b, 1.0 if node.op == "+" else -1.0)
            if isinstance(node, S.Binary) and node.op == "*":
                for u, v in ((node.l, node.r), (node.r, node.l)):
                    if isinstance(u, S.Num) and float(u.value) == int(u.value):
                        a = aff(v)
                        return None if a is None else a.scale(float(u.value))
            if invariant(node):
                return self._Aff(terms=[node])
            return None

        accesses, defined, ok = [], set(), [True]

        def access(node, kind):
            a = aff(S.Binary("+", node.base, node.index))
            if a is not None:
                accesses.append((kind, a))           # (list order = order within the trip)
                return
            # base[(affine) % M] with M invariant: a load somewhere in [base, base + M)
            ix = node.index
            if kind == 0 and isinstance(ix, S.Binary) and ix.op == "%" and invariant(ix.r):
                b = aff(node.base)
                if b is not None and not b.coef:
                    accesses.append((2, (b, ix.r)))
                    return
            ok[0] = False

        def walk(node, cond):
            if not ok[0]:
                return
            if isinstance(node, (S.Num,)):
                return
            if isinstance(node, S.Var):
                if node.name in temps and node.name not in defined:
                    ok[0] = False                            # would read the previous trip's value
                return
            if isinstance(node, S.Index):
                walk(node.base, cond); walk(node.index, cond)
                access(node, 0)
                return
            if isinstance(node, S.Assign):
                t = node.target
                walk(node.value, cond)
                if isinstance(t, S.Var):
                    if node.op != "=":
                        if t.name not in defined:
                            ok[0] = False
                        env.pop(t.name, None)
                        return
                    if cond:
                        if t.name not in defined:
                            ok[0] = False                    # a conditional write must follow an unconditional one in the same trip
                        env.pop(t.name, None)
                        return
                    a = aff(node.value)
                    defined.add(t.name)
                    if a is not None:
                        env[t.name] = a
                    else:
                        env.pop(t.name, None)
                    return
                walk(t.base, cond); walk(t.index, cond)
                if node.op != "=":
                    access(t, 0)                     # a compound store reads its cell first
                access(t, 1)
                return
            if isinstance(node, S.Cond):
                walk(node.cond, cond); walk(node.then, True)
                if node.els is not None:
                    walk(node.els, True)
                return
            if isinstance(node, S.If):
                walk(node.cond, cond); walk(node.then, True)
                if node.els is not None:
                    walk(node.els, True)
                return
            if isinstance(node, S.Binary):
                walk(node.l, cond); walk(node.r, cond or node.op in ("&&", "||"))
                return
            if isinstance(node, S.Unary):
                return walk(node.a, cond)
            if isinstance(node, S.Seq):
                for it in node.items:
                    walk(it, cond)
                return
            if isinstance(node, S.Call):
                for a in node.args:
                    walk(a, cond)
                return
            ok[0] = False

        for it in body:
            walk(it, False)
        if not ok[0] or not any(k == 1 for k, _ in accesses) or len(accesses) > 24:
            return None
        return ind, accesses

    def _map_guard(self, plan, c, start):
        """C++ that fills the access table of a planned map loop and asks za_map_ok whether its `c` trips are independent."""
        ind, accesses = plan
        rows = []
        for kind, a in accesses:
            if kind == 2:
                b, m = a
                base = " + ".join([c_double(b.const)] + [self.expr(t) for t in b.terms])
                rows.append(f"{{ {base}, {self.expr(m)}, 0, 2, {len(rows)} }}")
                continue
            a0 = " + ".join([c_double(a.const)] + [self.expr(t) for t in a.terms]
                            + [f"{c_double(cf)} * {start[v]}" for v, cf in a.coef.items() if cf != 0.0])
            sig = int(sum(cf * ind[v] for v, cf in a.coef.items()))
            rows.append(f"{{ {a0}, 0.0, {sig}, {kind}, {len(rows)} }}")
        tab = self.t("m")
        return f"const ZaMapAcc {tab}[] = {{ {', '.join(rows)} }};", f"za_map_ok({tab}, {len(rows)}, {c})"

    MAP_GROUP = int(os.environ.get("ZA_MAP_GROUP", "4"))      # (8 was measured: NeuroCV x1024 108 -> 178 ms, STFT x1024 17.0 -> 17.9 ms, PsychoConvolver with an impulse response unchanged)

    def _map_loop(self, plan, c, body_node, last_value=None):
        """The shared form of a planned map loop of `c` trips: lane r of the instance's R replica lanes runs trips r, r + R, ...
        of the first c - 1 (really storing: za_st only stores from the primary lane otherwise), the lanes exchange what
        those trips left behind (za_map_sync), and every lane runs the last trip, so that the script's temporaries, its
        counters and the loop's value end as they would serially. A lane's trips run in groups of MAP_GROUP whose arena stores
        are made after the whole group's loads: the device compiler must assume that a store aliases the next trip's loads, so
        trip by trip every trip would wait a full memory latency (the guard refuses a trip that reads back what it stored).
        Returns (prologue, guard expression, shared code)."""
        ind, _ = plan
        start = {v: self.t("i") for v in ind}
        k, rp, G = self.t("k"), self.t("q"), self.MAP_GROUP
        seti = lambda kk: " ".join(f"{self.var_ref(v)} = {start[v]} + (double)({kk}) * {c_double(st)};" for v, st in ind.items())
        table, okx = self._map_guard(plan, c, start)
        pro = " ".join(f"const double {start[v]} = {self.var_ref(v)};" for v in ind) + " " + table
        ints = " && ".join(f"za_coop_int({start[v]})" for v in ind)
        self.defer = []
        trips = [f"{{ {seti(f'{k} + {u} * (int64_t)s.rep_n')} (void)({self.expr(body_node)}); }}" for u in range(G)]
        slots, self.defer = self.defer, None
        decl = " ".join(f"int64_t {a} = 0; double {v} = 0.0; bool {o} = false;" for a, v, o in slots)
        flush = " ".join(f"if ({o}) za_st(s, {a}, {v});" for a, v, o in slots)
        body_expr = self.expr(body_node)
        shared = (f"const uint32_t {rp} = s.replica; s.replica = 0u; int64_t {k} = s.rep_i; "
                  f"for (; {k} + {G - 1} * (int64_t)s.rep_n < {c} - 1; {k} += {G} * (int64_t)s.rep_n) {{ {decl} {' '.join(trips)} {flush} }} "
                  f"for (; {k} < {c} - 1; {k} += s.rep_n) {{ {seti(k)} (void)({body_expr}); }} "
                  f"s.replica = {rp}; za_map_sync(s); {seti(c + ' - 1')} "
                  + (f"{last_value} = {body_expr};" if last_value else f"(void)({body_expr});"))
        # ("coopmap", not "coop": a map loop uses replica lanes where a leaf has them -- FFT builtins or accumulation loops on its
        #  audio path -- but is no reason to give a leaf thin wavefronts: NeuroCV x1024 went from 105 to 173 ms when it did)
        self.features.add("coopmap")
        return pro, f"{ints} && {okx}", shared

    def e_Loop(self, n):
        c, l, i = self.t("n"), self.t("l"), self.t("k")
        count = self.expr(n.count)
        head = (f"({{ int64_t {c} = za_loopcount({count}); double {l} = 0.0; "
                f"if ({c} > ZA_LOOP_CAP) {{ {c} = ZA_LOOP_CAP; s.err |= ZA_ERR_LOOP_CAP; }} ")
        inner = not any(isinstance(x, (S.Loop, S.While)) for x in _walk(n.body))
        hot = self.cur_sec in ("block", "sample")
        coop = ""
        if inner and hot and self.coop and not self.redirect and self._nodes(n.body) <= 400:
            plan = self._coop_plan(n)
            if plan:
                accs, ind = plan
                self.features.add("coop")
                part = {a: self.t("p") for a in accs}
                start = {v: self.t("i") for v in ind}
                self.redirect = dict(part)
                body_r = self.expr(n.body)
                self.redirect = {}
                k = self.t("k")
                seti = lambda kk: " ".join(f"{self.var_ref(v)} = {start[v]} + (double)({kk}) * {c_double(st)};" for v, st in ind.items())
                ok = " && ".join([f"{c} >= za_coop_min(s)"] + [f"za_coop_int({self.var_ref(v)})" for v in ind])
                # REPLICA LANES SHARE THE TRIPS (zab_generic.hip.h): lane r of the instance takes trips k = r, r + R, ...; the
                # partial sums meet in a fixed-order butterfly; every lane repeats the last trip (partials discarded) so that the
                # script's temporaries and the loop's value end as they would serially. Differs from the serial sum by rounding.
                coop = (f"if (ZA_COOP_ON(s) && {ok}) {{ "
                        + " ".join(f"const double {start[v]} = {self.var_ref(v)};" for v in ind)
                        + " " + " ".join(f"double {part[a]} = 0.0;" for a in accs)
                        # (one trip per loop iteration: four per iteration -- the group's loads in flight together, as the map loops
                        #  do -- was measured and is slower: DOT 147 -> 179 ms, TSEQ 2204 -> 2256 ms at 1024 x 48 000; again at one instance
                        #  per wavefront with the arena load's bounds check branch-free / as a branch, 1 / 4 trips, 12 000 frames: TSEQ 436 /
                        #  462 (shipped) / 463 / 489 ms, DOT 38.9 / 34.8 (shipped) / 39.2 / 34.9 ms)
                        + f" for (int64_t {k} = s.rep_i; {k} < {c}; {k} += s.rep_n) {{ {seti(k)} (void)({body_r}); }} "
                        + " ".join(f"{part[a]} = za_coop_sum(s, {part[a]});" for a in accs)
                        + " " + " ".join(f"{self.var_ref(a)} = {self.var_ref(a)} + {part[a]};" for a in accs)
                        + f" {{ " + " ".join(f"double {part[a]} = 0.0;" for a in accs) + f" {seti(c + ' - 1')} {l} = {body_r}; "
                        + " ".join(f"(void){part[a]};" for a in accs) + " } } else ")
        if not coop and inner and hot and self.coop and not self.redirect and self._nodes(n.body) <= 400:
            items = n.body.items if isinstance(n.body, S.Seq) else [n.body]
            plan = self._map_plan(items, set(self.scope[-1]) if self.scope else set())
            if plan:
                pro, okx, shared = self._map_loop(plan, c, n.body, last_value=l)
                coop = f"if (ZA_COOP_ON(s) && {c} >= za_coop_min(s) && ({{ {pro} bool z_ = {okx}; if (z_) {{ {shared} }} z_; }})) {{ }} else "
        body = self.expr(n.body)
        if (inner and hot and self.unroll and self.unrolled < self.UNROLL_MAX_LOOPS
                and self._nodes(n.body) <= self.UNROLL_MAX_NODES):
            self.unrolled += 1
            one = f"{l} = {body};"
            return (head + coop + f"{{ int64_t {i} = 0; for (; {i} + 4 <= {c}; {i} += 4) {{ {one} {one} {one} {one} }} "
                    f"for (; {i} < {c}; ++{i}) {{ {one} }} }} {l}; }})")
        return head + coop + f"{{ for (int64_t {i} = 0; {i} < {c}; ++{i}) {{ {l} = {body}; }} }} {l}; }})"

    def _nodes(self, node, depth=0) -> int:
        """AST size of an expression with user-function bodies counted at every call site (they are all inlined)."""
        total = 0
        for x in _walk(node):
            total += 1
            if isinstance(x, S.Call) and x.fn in self.p.fns and depth < 8:
                if x.fn not in self._fn_nodes:
                    self._fn_nodes[x.fn] = 0          # (specialisation forbids recursion; belt and braces)
                    self._fn_nodes[x.fn] = self._nodes(self.p.fns[x.fn].body, depth + 1)
                total += self._fn_nodes[x.fn]
        return total

    # -- calls ------------------------------------------------------------------------------
    def out_ptr(self, node, api) -> str:
        """C++ lvalue pointer for builtins with output arguments (variables or mem[] slots)."""
        if isinstance(node, S.Var) and node.name not in ("mem", "gmem"):
            return "&" + self.var_ref(node.name)
        if isinstance(node, S.Index) and not self._is_gmem(node):
            self.features.add("memptr")
            pre, (b, i) = self.ordered([node.base, node.index])
            self.features.add("mem")
            return self.wrap(pre, f"za_mem_ptr(s, za_addr({b}, {i}))")
        raise EmitError(f"{api} output arguments must be assignable variables or mem[] slots")

    def nargs(self, n, *counts):
        if len(n.args) not in counts:
            raise EmitError(f"{n.fn} expects {' or '.join(map(str, counts))} args")

    def call_rt(self, n, cname, with_state=True):
        pre, args = self.ordered(n.args)
        argl = ", ".join((["s"] if with_state else []) + args)
        return self.wrap(pre, f"{cname}({argl})")

    def e_Call(self, n):
        fn = n.fn
        if fn in ("slider", "spl"):
            self.nargs(n, 1)
            arr, off = self._dyn(fn)
            return f"za_dyn_ld64({arr}, {self.expr(n.args[0])}, {off})"
        if fn == "instance_id":
            self.nargs(n, 0)
            return "((double)s.instance_id)"
        if fn in GMEM_CALLS:
            self.features.add("gmem")
            return self.call_rt(n, "za_" + fn)
        if fn in HOST_ONLY:
            self.features.add("host:" + fn)
            pre, args = self.ordered([a for a in n.args])
            body = " ".join(f"(void)({a});" for a in args)
            return f"({{ {pre} {body} za_unsupported(s); }})"
        if fn in MSG_CALLS or (fn == "comm_join" and self.p.calls & MSG_CALLS):
            self.features.add("msg")
            if fn == "msg_recv":
                self.nargs(n, 7)
                pre, args = self.ordered(n.args[:1])
                outs = ", ".join(self.out_ptr(a, fn) for a in n.args[1:])
                return self.wrap(pre, f"za_msg_recv(s, {args[0]}, {outs})")
            if fn in ("msg_send_buf", "msg_sendto_buf", "msg_recv_buf"):
                self.features.add("msgbuf")
                self.features.add("mem")
            if fn == "msg_recv_buf":                # (channel, &src, &tag, dstBase, maxLen)
                self.nargs(n, 5)
                pre, args = self.ordered([n.args[0], n.args[3], n.args[4]])
                outs = ", ".join(self.out_ptr(a, fn) for a in n.args[1:3])
                return self.wrap(pre, f"za_msg_recv_buf(s, {args[0]}, {outs}, {args[1]}, {args[2]})")
            return self.call_rt(n, "za_" + fn)
        if fn in HOST_CONST:
            self.features.add("hostconst")
            vals = [a for a in n.args if not isinstance(a, (S.Var, S.Index)) or fn in ("comm_join", "instance_set_name")]
            pre, args = self.ordered(vals)
            return f"({{ {pre} {' '.join(f'(void)({a});' for a in args)} {float(HOST_CONST[fn])}; }})"
        if fn in MIDI_CALLS:
            self.features.add("midi")
            vals = [a for a in n.args if not (fn.startswith("midirecv") and isinstance(a, (S.Var, S.Index)))]
            pre, args = self.ordered(vals)                    # value arguments are still evaluated (side effects)
            return f"({{ {pre} {' '.join(f'(void)({a});' for a in args)} 0.0; }})"
        if fn in FILE_CALLS:
            self.features.add("file")
            if fn == "file_riff":
                self.nargs(n, 3)
                pre, args = self.ordered(n.args[:1])
                return self.wrap(pre, f"za_file_riff(s, {args[0]}, {self.out_ptr(n.args[1], fn)}, {self.out_ptr(n.args[2], fn)})")
            if fn == "file_var":
                self.nargs(n, 2)
                pre, args = self.ordered(n.args[:1])
                return self.wrap(pre, f"za_file_var(s, {args[0]}, {self.out_ptr(n.args[1], fn)})")
            if fn in ("file_open", "file_open_multi") and len(n.args) == 1:
                pre, args = self.ordered(n.args)
                return self.wrap(pre, f"za_{fn}(s, {args[0]}, 0.0)")
            return self.call_rt(n, "za_" + fn)
        if fn == "__memtop":
            self.nargs(n, 0)
            return c_double(float(self.p.memtop))
        if fn in POOL_READ_CALLS:
            self.features.add("pool")
            if fn in ("sample_read2", "sample_read2_interp"):
                self.nargs(n, 5)
                pre, args = self.ordered(n.args[:3])
                o1, o2 = self.out_ptr(n.args[3], fn), self.out_ptr(n.args[4], fn)
                return self.wrap(pre, f"za_{fn}(s, {', '.join(args)}, {o1}, {o2})")
            return self.call_rt(n, "za_" + fn)
        if fn.startswith("gfx_") or fn in NOOP_CALLS:
            if not n.args:
                return "0.0"
            return "({ " + " ".join(f"(void)({self.expr(a)});" for a in n.args) + " 0.0; })"
        if n.fn in self.p.fns:
            f = self.p.fns[n.fn]
            if len(n.args) != len(f.params):
                # the reference IR builder would reject a wrong-arity call; pad/truncate is not JSFX behaviour
                raise EmitError(f"{n.fn}: expected {len(f.params)} args, got {len(n.args)}")
            return self.call_rt(n, "fn_" + c_ident(n.fn))
        if fn == "abs":
            fn = "fabs"
        if fn in ("min", "max"):
            self.nargs(n, 2)
            return self.call_rt(n, "za_" + fn, with_state=False)
        if fn in ("sqr", "sign", "invsqrt"):
            self.nargs(n, 1)
            return f"za_{fn}({self.expr(n.args[0])})"
        if fn in PURE_MATH1:
            self.nargs(n, 1)
            return f"{PURE_MATH1[fn]}({self.expr(n.args[0])})"
        if fn in PURE_MATH2:
            self.nargs(n, 2)
            return self.call_rt(n, PURE_MATH2[fn], with_state=False)
        if fn == "rand":
            self.nargs(n, 0, 1)
            self.features.add("rand")
            return f"za_rand(s, {self.expr(n.args[0]) if n.args else '1.0'})"
        if fn == "freembuf":
            self.nargs(n, 1)
            return f"({{ (void)({self.expr(n.args[0])}); 0.0; }})"
        if fn in ("sliderchange", "slider_automate", "slider_show"):
            self.features.add("sliderchange")
            m = self._mask_arg(n.args[0]) if n.args else None
            if fn == "sliderchange":
                self.nargs(n, 1)
                return f"za_sliderchange(s, {m})"
            if fn == "slider_automate":
                self.nargs(n, 1, 2)
                end = self.expr(n.args[1]) if len(n.args) == 2 else "0.0"
                return f"({{ double m_ = {m}; double e_ = {end}; za_slider_automate(s, m_, e_); }})"
            self.nargs(n, 1, 2)
            if len(n.args) == 1:
                return f"za_slider_show1(s, {m})"
            return f"({{ double m_ = {m}; double e_ = {self.expr(n.args[1])}; za_slider_show2(s, m_, e_); }})"
        if fn == "slider_next_chg":
            self.nargs(n, 2)
            idx = self.expr(n.args[0])
            try:
                ptr = self.out_ptr(n.args[1], fn)
            except EmitError:
                return f"({{ double i_ = {idx}; (void)({self.expr(n.args[1])}); za_slider_next_chg(s, i_, (double*)0); }})"
            return f"({{ double i_ = {idx}; za_slider_next_chg(s, i_, {ptr}); }})"
        if fn == "memset":
            self.nargs(n, 3)
            self.features.add("mem")
            return self.call_rt(n, "za_memset")
        if fn == "memcpy":
            self.nargs(n, 3)
            self.features.add("mem")
            return self.call_rt(n, "za_memcpy")
        if fn in FFT_CALLS:
            self.nargs(n, 2)
            self.features.add("fft")
            return self.call_rt(n, "za_" + fn)
        if fn in ("__fft_nat", "__ifft_nat", "__fft_real_nat", "__ifft_real_nat"):          # fused pairs (_fuse_fft_pairs)
            self.features.add("fft")
            return self.call_rt(n, "za_" + fn[2:])
        if fn == "convolve_c":
            self.nargs(n, 3)
            self.features.add("fft")
            return self.call_rt(n, "za_convolve_c")
        raise EmitError(f"Unknown function call {n.fn}")

    def _mask_arg(self, a) -> str:
        if isinstance(a, S.Var) and not self._is_param(a.name):
            k = is_slider_name(a.name)
            if k is not None and 1 <= k <= 64 and a.name == f"slider{k}":
                return c_double(float(1 << (k - 1)))
        return self.expr(a)

    # -- top level --------------------------------------------------------------------------
    def function(self, name: str, f) -> str:
        self.scope.append(set(f.params))
        m = re.match(r"__fn__(init|slider|block|sample)__", name)      # specialised per calling section (program.py)
        self.cur_sec = m.group(1) if m else "sample"
        body = self.expr(f.body)
        self.scope.pop()
        params = "".join(f", double p_{c_ident(p)}" for p in f.params)
        return f"template <class S> ZA_UFN double fn_{c_ident(name)}(S& s{params}) {{ return {body}; }}"

    def section(self, sec: str) -> str:
        self.cur_sec = sec
        body = " ".join(self.stmt(st) for st in self._fuse_fft_pairs(list(self.p.sections.get(sec, []))))
        return f"template <class S> ZA_SECTION_FN void za_section_{sec}(S& s) {{ {body} }}"

    def emit(self) -> str:
        order = self._fn_order()
        protos = []
        for name in order:
            f = self.p.fns[name]
            params = "".join(f", double p_{c_ident(p)}" for p in f.params)
            protos.append(f"template <class S> ZA_UFN double fn_{c_ident(name)}(S& s{params});")
        fns = [self.function(name, self.p.fns[name]) for name in order]
        secs = [self.section(sec) for sec in ("init", "slider", "block", "sample")]
        return "\n".join(protos + fns + secs) + "\n"

    def _fn_order(self) -> List[str]:
        """Callees before callers (specialisation forbids cycles), so always_inline can resolve bottom-up."""
        seen, out = set(), []

        def visit(name):
            if name in seen:
                return
            seen.add(name)
            for node in _walk(self.p.fns[name].body):
                if isinstance(node, S.Call) and node.fn in self.p.fns:
                    visit(node.fn)
            out.append(name)

        for name in self.p.fns:
            visit(name)
        return out


def _walk(n):
    yield n
    for c in S.children(n):
        yield from _walk(c)
