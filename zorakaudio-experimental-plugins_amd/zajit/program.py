"""JSFX script -> analysed program (sections, specialised user functions, variable table, I/O, capabilities).

Restates, with its own data flow, the front half of the reference AOT compiler so that the translator's results
are interchangeable with the reference's generated header:
  import expansion ...... dsp_jsfx_aot.py:839-949
  section split ......... dsp_jsfx_aot.py:951-974   (only @init/@slider/@block/@sample are compiled, :2283)
  function lowering ..... dsp_jsfx_aot.py:1821-2048 (per-caller-section specialisation; local() -> persistent
                          `__fnlocal__<sec>__<fn>__<name>` vars; instance()/this. -> `<namespace>.<name>` vars)
  variable table ........ dsp_jsfx_aot.py:1038-1099 (sorted names -> vars[] index)
  options / memtop ...... dsp_jsfx_aot.py:1138-1178
  I/O channel inference . dsp_jsfx_aot.py:1662-1801
  section validation .... dsp_jsfx_aot.py:1544-1606
  capability flags ...... dsp_jsfx_aot.py:1406-1542,1608-1660
"""
from __future__ import annotations

import math
import os
import re
from dataclasses import dataclass, field
from pathlib import Path
from typing import Dict, List, Optional, Set, Tuple

from . import syntax as S
from .sliders import parse_slider_decls, slider_aliases

SECTIONS = ("init", "slider", "block", "sample")
_SECTION_RE = re.compile(r"^\s*@([A-Za-z_][A-Za-z0-9_]*)\b.*$")
_IMPORT_RE = re.compile(r"^\s*import\s+(?:\"([^\"]+)\"|'([^']+)'|([^\s;]+))\s*;?\s*(?://.*)?$")
BUILTIN_VARS = {"mem", "gmem", "srate", "samplesblock", "midi_bus", "ext_midi_bus"}
DEFAULT_MEMTOP = 8 * 1024 * 1024


# ----------------------------------------------------------------------------------------------
# text level
# ----------------------------------------------------------------------------------------------

class _Bundle:
    def __init__(self):
        self.preamble: List[str] = []
        self.order: List[str] = []
        self.body: Dict[str, List[str]] = {}
        self.header: Dict[str, str] = {}

    def add(self, sec, lines, header=None):
        if sec not in self.body:
            self.body[sec] = []
            self.order.append(sec)
        if header is not None and sec not in self.header:
            self.header[sec] = header
        self.body[sec].extend(lines)


def _load_bundle(path: Path, stack: List[Path]) -> _Bundle:
    b = _Bundle()
    cur: Optional[str] = None
    for raw in path.read_text(encoding="utf-8", errors="replace").splitlines(True):
        mi, ms = _IMPORT_RE.match(raw), _SECTION_RE.match(raw)
        if mi:
            token = next((g for g in mi.groups() if g), "")
            if token:
                inc = (path.parent / token).resolve()
                if not inc.exists():
                    raise FileNotFoundError(f"Unable to resolve JSFX import {token!r} from {path}")
                if inc in stack:
                    raise ValueError("Cyclic JSFX import chain: " + " -> ".join(map(str, stack + [inc])))
                child = _load_bundle(inc, stack + [inc])
                if cur is None:
                    b.preamble.extend(child.preamble)
                    for sec in child.order:
                        b.add(sec, child.body[sec], child.header.get(sec))
                else:
                    b.add(cur, child.preamble)
                    for sec in child.order:
                        b.add(sec, child.body[sec], child.header.get(sec))
                continue
            (b.preamble if cur is None else b.body[cur]).append(raw)
            continue
        if ms:
            cur = ms.group(1)
            b.add(cur, [], raw)
            continue
        if cur is None:
            b.preamble.append(raw)
        else:
            b.body[cur].append(raw)
    return b


def expand_imports(path: Path) -> str:
    src = Path(path).resolve()
    b = _load_bundle(src, [src])
    out = list(b.preamble)
    for sec in b.order:
        h = b.header.get(sec, f"@{sec}\n")
        out.append(h if h.endswith("\n") else h + "\n")
        out.extend(b.body[sec])
        if out and not out[-1].endswith("\n"):
            out.append("\n")
    return "".join(out)


def split_sections(text: str) -> Dict[str, Tuple[str, int]]:
    """{name: (code, first_line_number)}; repeated markers of one section concatenate."""
    chunks: Dict[str, List[str]] = {}
    first: Dict[str, int] = {}
    cur = None
    for i, ln in enumerate(text.splitlines(True)):
        m = _SECTION_RE.match(ln)
        if m:
            cur = m.group(1)
            chunks.setdefault(cur, [])
            first.setdefault(cur, i + 2)
        elif cur is not None:
            chunks[cur].append(ln)
    return {k: ("".join(v), first.get(k, 1)) for k, v in chunks.items()}


def parse_options(text: str) -> Dict[str, str]:
    opts: Dict[str, str] = {}
    for ln in text.splitlines():
        m = re.match(r"^\s*options\s*:\s*(.*)$", ln, re.I)
        if not m:
            continue
        for tok in re.split(r"[\s,]+", m.group(1).strip()):
            if tok and "=" in tok:
                k, v = tok.split("=", 1)
                if k.strip():
                    opts[k.strip().lower()] = v.strip()
    return opts


def memtop_slots(opts: Dict[str, str]) -> int:
    raw = str(opts.get("maxmem", "") or "").strip()
    if not raw:
        return DEFAULT_MEMTOP
    try:
        n = int(float(raw))
    except Exception:
        return DEFAULT_MEMTOP
    return n if n > 0 else DEFAULT_MEMTOP


def pin_hints(text: str) -> Dict[str, Optional[int]]:
    saw = {"inputs": False, "outputs": False}
    cnt = {"inputs": 0, "outputs": 0}
    for ln in text.splitlines():
        ln = ln.split("//", 1)[0].split(";", 1)[0]
        m = re.match(r"^\s*(in_pin|out_pin)\s*:\s*(.*?)\s*$", ln, re.I)
        if not m:
            continue
        k = "inputs" if m.group(1).lower() == "in_pin" else "outputs"
        saw[k] = True
        if m.group(2).strip().lower() == "none":
            cnt[k] = 0
        else:
            cnt[k] += 1
    return {k: (cnt[k] if saw[k] else None) for k in cnt}


# ----------------------------------------------------------------------------------------------
# user-function lowering
# ----------------------------------------------------------------------------------------------

def _mangle(text: str) -> str:
    out = "".join(ch if (ch.isalnum() or ch == "_") else f"_x{ord(ch):02X}_" for ch in text) or "_"
    return "_" + out if out[0].isdigit() else out


def _uses_this(n: S.Node) -> bool:
    if isinstance(n, S.Var):
        return n.name == "this" or n.name.startswith("this.")
    if isinstance(n, S.Call) and (n.fn == "this" or n.fn.startswith("this.")):
        return True
    return any(_uses_this(c) for c in S.children(n))


def _rel_ns(prefix: str, cur: Optional[str]) -> Optional[str]:
    if prefix == "this":
        return cur
    if prefix.startswith("this."):
        suf = prefix[5:]
        if cur:
            return f"{cur}.{suf}" if suf else cur
        return suf or cur
    return prefix


class _Lowerer:
    def __init__(self, fn_defs: Dict[str, S.Node]):
        self.defs = fn_defs
        self.needs_ns = {k: bool(f.instances) or _uses_this(f.body) for k, f in fn_defs.items()}
        self.out: Dict[str, S.Node] = {}
        self.cache: Dict[tuple, str] = {}
        self.busy: Set[tuple] = set()

    def specialise(self, section: str, base: str, call_ns: Optional[str]) -> str:
        f = self.defs[base]
        ns = call_ns if self.needs_ns[base] else None
        if self.needs_ns[base] and not ns:
            ns = base
        key = (section, base, ns)
        if key in self.cache:
            return self.cache[key]
        if key in self.busy:
            raise ValueError(f"Recursive or cyclic user-function specialization detected for {base}")
        name = f"__fn__{_mangle(section)}__{_mangle(base)}" + (f"__ns__{_mangle(ns)}" if ns else "")
        self.cache[key] = name
        self.busy.add(key)
        ctx = dict(section=section, ns=ns, params=set(f.params),
                   locals={l: f"__fnlocal__{_mangle(section)}__{_mangle(base)}__{_mangle(l)}" for l in f.locals},
                   inst={v: f"{ns}.{v}" for v in f.instances} if ns else {})
        body = self.rewrite(f.body, ctx)
        self.out[name] = S.FuncDef(name, list(f.params), [], [], body, line=f.line, col=f.col)
        self.busy.discard(key)
        return name

    def rewrite(self, n: S.Node, ctx) -> S.Node:
        if isinstance(n, S.Var):
            nm = n.name
            if nm in ctx["params"]:
                return n
            if nm in ctx["locals"]:
                nm = ctx["locals"][nm]
            elif nm in ctx["inst"]:
                nm = ctx["inst"][nm]
            elif nm == "this":
                nm = ctx["ns"] or nm
            elif nm.startswith("this."):
                suf = nm[5:]
                nm = (f"{ctx['ns']}.{suf}" if suf else ctx["ns"]) if ctx["ns"] else (suf or nm)
            return n if nm == n.name else S.Var(nm, line=n.line, col=n.col)
        if isinstance(n, S.Call):
            fn = n.fn
            if fn in self.defs:
                fn = self.specialise(ctx["section"], fn, None)
            else:
                parts = fn.split(".")
                if len(parts) >= 2 and parts[-1] in self.defs:
                    fn = self.specialise(ctx["section"], parts[-1], _rel_ns(".".join(parts[:-1]), ctx["ns"]))
            return S.Call(fn, [self.rewrite(a, ctx) for a in n.args], line=n.line, col=n.col)
        if isinstance(n, S.FuncDef):
            raise TypeError("Unexpected nested FunctionDef during lowering")
        return S.rebuild(n, lambda c: self.rewrite(c, ctx))


def lower_functions(programs: Dict[str, List[S.Node]]):
    defs: Dict[str, S.Node] = {}
    stripped: Dict[str, List[S.Node]] = {}
    for sec, prog in programs.items():
        keep = []
        for n in prog:
            if isinstance(n, S.FuncDef):
                defs[n.name] = n          # last definition wins
            else:
                keep.append(n)
        stripped[sec] = keep
    if not defs:
        return stripped, {}
    lw = _Lowerer(defs)
    out = {}
    for sec, prog in stripped.items():
        ctx = dict(section=sec, ns=None, params=set(), locals={}, inst={})
        out[sec] = [lw.rewrite(n, ctx) for n in prog]
    return out, lw.out


# ----------------------------------------------------------------------------------------------
# analyses
# ----------------------------------------------------------------------------------------------

def is_spl_name(name: str) -> Optional[int]:
    if name.startswith("spl") and name[3:].isdigit():
        return int(name[3:])
    return None


def is_slider_name(name: str) -> Optional[int]:
    if name.startswith("slider") and name[6:].isdigit():
        return int(name[6:])
    return None


def collect_vars(programs, fns) -> Dict[str, int]:
    names: Set[str] = set()

    def walk(n, shadow):
        if isinstance(n, S.Var):
            nm = n.name
            if nm in shadow or nm in BUILTIN_VARS or nm.startswith("$"):
                return
            if is_spl_name(nm) is not None or is_slider_name(nm) is not None:
                return
            names.add(nm)
            return
        for c in S.children(n):
            walk(c, shadow)

    for prog in programs.values():
        for st in prog:
            walk(st, frozenset())
    for f in fns.values():
        walk(f.body, frozenset(f.params) | frozenset(f.locals))
    return {nm: i for i, nm in enumerate(sorted(names))}


_SECTION_ONLY_BLOCK = {
    # builtin -> message fragment; the reference rejects these outside @block (dsp_jsfx_aot.py:1544-1606)
}


def walk_all(programs, fns):
    for sec, prog in programs.items():
        for st in prog:
            yield from _walk(st)
    for f in fns.values():
        yield from _walk(f.body)


def _walk(n):
    yield n
    for c in S.children(n):
        yield from _walk(c)


def called_names(programs, fns) -> Set[str]:
    return {n.fn for n in walk_all(programs, fns) if isinstance(n, S.Call)}


def infer_io(programs, fns, hints) -> Dict[str, int]:
    """Channel counts from splN reads/writes, overridden by in_pin/out_pin declarations
    (same rules and fallbacks as dsp_jsfx_aot.py:1662-1801)."""
    reads: Set[int] = set()
    writes: Set[int] = set()

    def note(name, is_write):
        k = is_spl_name(name)
        if k is not None and re.fullmatch(r"spl[0-9]+", name) and 0 <= k < 64:
            (writes if is_write else reads).add(k)

    def walk(n, shadow, wctx=False):
        if isinstance(n, S.Var):
            if n.name not in shadow:
                note(n.name, wctx)
            return
        if isinstance(n, S.Assign):
            walk(n.target, shadow, True)
            if n.op != "=":
                walk(n.target, shadow, False)
            walk(n.value, shadow, False)
            return
        for c in S.children(n):
            walk(c, shadow, False)

    for prog in programs.values():
        for st in prog:
            walk(st, frozenset())
    for f in fns.values():
        walk(f.body, frozenset(f.params) | frozenset(f.locals))
    max_r = max(reads) if reads else -1
    max_w = max(writes) if writes else -1
    n_in, n_out = max_r + 1, max_w + 1
    d_in, d_out = hints.get("inputs"), hints.get("outputs")
    if d_in is not None:
        n_in = int(d_in)
    if d_out is not None:
        n_out = int(d_out)
    if d_in is None and d_out is None and n_in == 0 and n_out == 0:
        n_in = n_out = 2
    if d_in is None and n_in == 0 and n_out > 0:
        n_in = n_out
    if d_out is None and n_out == 0 and n_in > 0:
        n_out = n_in
    n_in, n_out = max(0, min(64, n_in)), max(0, min(64, n_out))
    return {"inputs": n_in, "outputs": n_out, "process": max(n_in, n_out), "max_read": max_r, "max_write": max_w}


# ----------------------------------------------------------------------------------------------
# named constants
# ----------------------------------------------------------------------------------------------
# JSFX has no `const`: a script names its record offsets, table sizes and enumerations by assigning them once at the top of
# @init (Sample: VC_ACTIVE = 0; VC_STRIDE = 96; MAX_VOICES = 16; ... several hundred of them) and every v[VC_ACTIVE] of the
# per-sample code is then a load of the variable table plus a floating-point add and a rounding for the address. A variable
#   * with exactly ONE write site in the whole program (sections and specialised functions; compound assignments and the
#     output arguments of builtins count as writes), which is
#   * a plain top-level statement of @init `name = <constant expression>` (numbers, + - * /, constants established by the
#     statements before it), and
#   * neither a slider alias nor one of the host's variables
# holds that value whenever @slider, @block or @sample run: @init has run to its end before any of them, nothing else stores
# to it, and a state image of the same script carries the same number. Reads of it in those sections (and in the functions
# specialised for them) become the literal; @init itself is left alone, so the table cell is written as before and reads
# that come before the assignment there still see 0. (IEEE double arithmetic folded here is the arithmetic EEL2 would do.)
_OUT_ARG_BUILTINS = ("midirecv", "midisyx", "msg_recv", "file_", "sample_read2", "slider_next_chg", "get_", "str", "sprintf",
                     "match", "gfx_")


# variables a host or the @gfx side may store to (this engine never does; left alone all the same)
_HOST_VARS = {"tempo", "play_state", "play_position", "beat_position", "ts_num", "ts_denom", "trigger", "num_ch", "pdc_delay",
              "pdc_bot_ch", "pdc_top_ch", "pdc_midi"}


def _writes(n, shadow, out: Dict[str, int]):
    if isinstance(n, S.Assign) and isinstance(n.target, S.Var) and n.target.name not in shadow:
        out[n.target.name] = out.get(n.target.name, 0) + 1
    if isinstance(n, S.Call) and n.fn.startswith(_OUT_ARG_BUILTINS):
        for a in n.args:
            if isinstance(a, S.Var) and a.name not in shadow:
                out[a.name] = out.get(a.name, 0) + 2          # (never a constant)
    for c in S.children(n):
        _writes(c, shadow, out)


def _const_value(n, known: Dict[str, float]) -> Optional[float]:
    if isinstance(n, S.Num):
        return float(n.value)
    if isinstance(n, S.Var):
        return known.get(n.name)
    if isinstance(n, S.Unary) and n.op == "-":
        a = _const_value(n.a, known)
        return None if a is None else -a
    if isinstance(n, S.Binary) and n.op in ("+", "-", "*", "/"):
        l, r = _const_value(n.l, known), _const_value(n.r, known)
        if l is None or r is None or (n.op == "/" and r == 0.0):
            return None
        v = l + r if n.op == "+" else l - r if n.op == "-" else l * r if n.op == "*" else l / r
        return v if math.isfinite(v) else None
    return None


def named_constants(programs, fns, aliases) -> Dict[str, float]:
    counts: Dict[str, int] = {}
    for prog in programs.values():
        for st in prog:
            _writes(st, frozenset(), counts)
    for f in fns.values():
        _writes(f.body, frozenset(f.params) | frozenset(f.locals), counts)
    skip = set(aliases.values()) | BUILTIN_VARS | _HOST_VARS
    known: Dict[str, float] = {}
    for st in programs.get("init", []):
        if not (isinstance(st, S.Assign) and st.op == "=" and isinstance(st.target, S.Var)):
            continue
        nm = st.target.name
        if counts.get(nm) != 1 or nm in skip or nm.startswith(("$", "#", "gfx_", "mouse_", "ext_", "pdc_")) or is_spl_name(nm) is not None \
                or is_slider_name(nm) is not None:
            continue
        v = _const_value(st.value, known)
        if v is not None:
            known[nm] = v
    # What is worth a literal is what ends up in addresses, trip counts and comparisons of counters: whole numbers. A fractional
    # coefficient (eps, ln10, a default gain) gains nothing as a literal -- the kernels held it in a register as an invariant -- and
    # costs a 64-bit constant materialised where it is used (SOMA +6 %, DOT +4 % with every constant folded, round 4 sweep).
    if not os.environ.get("ZA_CONSTS_ALL"):
        known = {nm: v for nm, v in known.items() if v == math.floor(v) and abs(v) < 2.0 ** 31}
    return known


def fold_named_constants(programs, fns, consts: Dict[str, float]):
    if not consts:
        return programs, fns

    def sub(n, shadow):
        if isinstance(n, S.Var):
            if n.name in consts and n.name not in shadow:
                return S.Num(consts[n.name], line=n.line, col=n.col)
            return n
        return S.rebuild(n, lambda c: sub(c, shadow))

    out = {sec: (prog if sec == "init" else [sub(st, frozenset()) for st in prog]) for sec, prog in programs.items()}
    out_f = {}
    for name, f in fns.items():
        m = re.match(r"__fn__(init|slider|block|sample)__", name)
        if m and m.group(1) != "init":
            shadow = frozenset(f.params) | frozenset(f.locals)
            f = S.FuncDef(f.name, f.params, f.locals, f.instances, sub(f.body, shadow), line=f.line, col=f.col)
        out_f[name] = f
    return out, out_f


@dataclass
class Program:
    name: str
    text: str
    sections: Dict[str, List[S.Node]]
    fns: Dict[str, S.Node]
    vars: Dict[str, int]
    options: Dict[str, str]
    memtop: int
    io: Dict[str, int]
    slider_decls: dict
    aliases: Dict[int, str]
    calls: Set[str]
    strings: List[str] = field(default_factory=list)

    @property
    def nvars(self) -> int:
        return max(1, len(self.vars))

    def has(self, sec: str) -> bool:
        return bool(self.sections.get(sec))

    def uses(self, *names) -> bool:
        return any(n in self.calls for n in names)


# Section validity of host-coupled builtins: the rule table and the message text of the reference's
# validate_builtin_sections (dsp_jsfx_aot.py:1544-1605), applied like there to the calls written directly in a section
# (after user functions were specialised, so calls inside function bodies are not visited). The reference's own comm and
# sample-pool tests pin the messages (scripts/run_dsp-jsfx_commtests.py:65-66, run_dsp-jsfx_sample_pool_tests.py:60).
_BLOCK_ONLY = {"msg_send", "msg_sendto", "msg_recv", "msg_send_buf", "msg_sendto_buf", "msg_recv_buf", "msg_avail", "msg_kind",
               "msg_length", "msg_dropped", "msg_clear", "msg_peer_count", "msg_peer_id", "msg_peer_name", "msg_peer_uid",
               "msg_peer_caps", "msg_peer_alive", "gmem_get", "gmem_put", "gmem_fill", "gmem_zero", "gmem_copy",
               "sample_export_mem", "sample_export_mem2"}
_SETUP = {"comm_join", "msg_subscribe", "msg_unsubscribe", "msg_advertise", "instance_set_name", "instance_get_name",
          "instance_uid", "gmem_attach", "gmem_attach_size", "track_name", "track_name_available", "track_name_seq",
          "host_track_name", "host_track_name_available", "host_track_name_seq", "sample_pool_from_slot",
          "sample_pool_set_mode", "sample_pool_set_budget_mb", "sample_pool_commit", "instance_id"}
_POOL_RUNTIME = {"sample_pool_state", "sample_pool_selected", "sample_pool_loaded", "sample_pool_failed", "sample_pool_ram_mb",
                 "sample_pool_generation", "sample_get", "sample_len", "sample_channels", "sample_srate", "sample_peak",
                 "sample_rms", "sample_preview_bins", "sample_read", "sample_read_interp", "sample_read2",
                 "sample_read2_interp", "sample_preview_read", "sample_name"}


def validate_builtin_sections(programs) -> None:
    def visit(section, n):
        if isinstance(n, S.Call):
            fn = n.fn
            if fn in _BLOCK_ONLY and section != "block":
                raise S.JsfxSyntaxError(f"{fn}() is only valid in @block at {n.line}:{n.col}")
            if fn in _SETUP and section not in ("init", "slider", "block"):
                raise S.JsfxSyntaxError(f"{fn}() is only valid in @init, @slider, or @block at {n.line}:{n.col}")
            if fn in _POOL_RUNTIME and section not in ("init", "slider", "block", "sample"):
                raise S.JsfxSyntaxError(f"{fn}() is only valid in @init, @slider, @block, or @sample at {n.line}:{n.col}")
        for c in S.children(n):
            visit(section, c)

    for section, nodes in programs.items():
        for node in nodes:
            visit(section, node)


def analyse(text: str, name: str = "jsfx") -> Program:
    secs = split_sections(text)
    progs = {}
    for sec in SECTIONS:
        if sec in secs:
            code, line0 = secs[sec]
            progs[sec] = S.parse_section(code, line0)
        else:
            progs[sec] = []
    progs, fns = lower_functions(progs)
    validate_builtin_sections(progs)
    vars_ = collect_vars(progs, fns)
    opts = parse_options(text)
    decls = parse_slider_decls(text)
    consts = {} if os.environ.get("ZA_NO_CONSTS") else named_constants(progs, fns, slider_aliases(decls))
    progs, fns = fold_named_constants(progs, fns, consts)
    return Program(name=name, text=text, sections=progs, fns=fns, vars=vars_, options=opts,
                   memtop=memtop_slots(opts), io=infer_io(progs, fns, pin_hints(text)),
                   slider_decls=decls, aliases=slider_aliases(decls), calls=called_names(progs, fns))


def analyse_file(path) -> Program:
    p = Path(path)
    return analyse(expand_imports(p), name=p.stem)
