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
HOST_ONLY = {"msg_send_buf", "msg_sendto_buf", "msg_recv_buf", "msg_peer_name", "msg_peer_uid",
             "sample_name", "sample_preview_read", "sample_preview_bins"}
# scalar message bus between the instances of one engine (csrc/zart_msg.h)
MSG_CALLS = {"msg_subscribe", "msg_unsubscribe", "msg_advertise", "msg_send", "msg_sendto", "msg_avail", "msg_kind", "msg_recv",
             "msg_length", "msg_dropped", "msg_clear", "msg_peer_count", "msg_peer_id", "msg_peer_caps", "msg_peer_alive"}
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
            if bop is None:
                self.features.add("mem")
                return f"({{ double {r} = {rhs}; {addr} za_st(s, {a}, {r}); }})"
            self.features.add("mem")
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

    def e_Seq(self, n):
        if not n.items:
            return "0.0"
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
            return (f"{{ int64_t {c} = 0; while (za_truthy({self.expr(n.cond)})) {{ {self.stmt(n.body)} "
                    f"if (++{c} >= ZA_LOOP_CAP) {{ s.err |= ZA_ERR_LOOP_CAP; break; }} }} }}")
        return f"(void)({self.expr(n)});"

    def e_If(self, n):
        return "({ " + self.stmt(n) + " 0.0; })"

    def e_While(self, n):
        return "({ " + self.stmt(n) + " 0.0; })"

    def e_Loop(self, n):
        c, l, i = self.t("n"), self.t("l"), self.t("k")
        count = self.expr(n.count)
        body = self.expr(n.body)
        head = (f"({{ int64_t {c} = za_loopcount({count}); double {l} = 0.0; "
                f"if ({c} > ZA_LOOP_CAP) {{ {c} = ZA_LOOP_CAP; s.err |= ZA_ERR_LOOP_CAP; }} ")
        # Small innermost bodies are written out four times per trip: the trip count differs per lane, so the device
        # compiler does not unroll these loops itself, and one iteration alone leaves it nothing to overlap the arena
        # loads of the next iteration with (a serial script pays a full memory latency per iteration otherwise).
        inner = not any(isinstance(x, (S.Loop, S.While)) for x in _walk(n.body))
        if (inner and self.unroll and self.unrolled < self.UNROLL_MAX_LOOPS and self.cur_sec in ("block", "sample")
                and self._nodes(n.body) <= self.UNROLL_MAX_NODES):
            self.unrolled += 1
            one = f"{l} = {body};"
            return (head + f"int64_t {i} = 0; for (; {i} + 4 <= {c}; {i} += 4) {{ {one} {one} {one} {one} }} "
                    f"for (; {i} < {c}; ++{i}) {{ {one} }} {l}; }})")
        return head + f"for (int64_t {i} = 0; {i} < {c}; ++{i}) {{ {l} = {body}; }} {l}; }})"

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
        body = " ".join(self.stmt(st) for st in self.p.sections.get(sec, []))
        return f"template <class S> ZA_FN void za_section_{sec}(S& s) {{ {body} }}"

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
