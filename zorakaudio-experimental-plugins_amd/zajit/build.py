"""Build driver: JSFX leaves -> gfx950 plugin modules, plus the host runtime library.

Plays the role scripts/build.py + dsp_jsfx_aot.py play in the reference (scripts/build.py:343-438: per leaf,
run the AOT translator, then compile), with the same leaf discovery contract (plugins/<Category>/<Key>/plugin.json,
`pluginType` jsfx, `entry` -> src/*.jsfx; scripts/pluginlib.py:105-240) and CLI spelling (--only, --list).

    python -m zajit.build --plugins-root /root/reference/plugins --only DDT [--correctness-check]
    python -m zajit.build --list

--only takes what scripts/build.py takes (scripts/pluginlib.py:243-257): a case-insensitive substring of the category, key, slug,
name, path, bundleId or clapId. --correctness-check (scripts/build.py:556): after a leaf is built it is run on the device
against the reference shadow VM's recorded results (zajit/check.py) and the build fails above the reference's tolerances.

Outputs (all git-ignored, all travel to the GPU box):
    <pkg>/lib/libzabatch.so        host runtime + C ABI (include/zabatch.h)
    <pkg>/lib/libzab_<Key>.so      one module per leaf (generic kernels [+ hand-written leaf kernel])
    <pkg>/lib/<Key>.json           metadata (vars table, io, slider declarations) for Python hosts/tests
    <pkg>/_gen/<Key>_module.hip    generated source (kept for inspection)
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import subprocess
import sys
import time
from pathlib import Path
from typing import Dict, List, Optional

from . import codegen, program

PKG = Path(__file__).resolve().parent.parent
CSRC = PKG / "csrc"
LIB = PKG / "lib"
GEN = PKG / "_gen"
ARCH = "gfx950"
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -ffp-contract=off: the reference's IR carries no fast-math/contract flags, so a*b+c is two roundings.
HIP_FLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
             "-fno-fast-math", "-Wno-unused-value", "-Wno-unused-variable", "-Wno-parentheses-equality"]
HIP_FLAGS += os.environ.get("ZA_EXTRA_HIP_FLAGS", "").split()      # experiments (e.g. -DZA_LD_BRANCH)

# leaves with a hand-written kernel: name -> header under csrc/kernels/
FAST_KERNELS = {"DDT": "kernels/ddt_ring2.hip.h"}
FAST_KERNEL_DEPS = {"DDT": ["kernels/ddt_fast.hip.h"]}     # headers the hand-written kernel file includes
# Build variants of the repo's own test fixtures (no catalog leaf is named here: what used to be per-leaf choices -- the arena
# load's bounds check as a branch for four delay-line scripts, replica lanes for two scripts' elementwise loops, uniform branches
# for one script's mode switches -- are now rules over what the translator finds in a script: zajit/codegen.py make_unit,
# zajit/tpar/plan.py _wants_uniform_guards; DESIGN.md section 4.1 has the measurements the rules came from).
LEAF_FLAGS = {"fx_dynkat_s1": ["-DZT_SPEC_MAX=1"],       # test variant: switched recurrences mostly fall back to their serial loop
              # FFT builtins with the whole 4096-point transform in LDS (zart_fft.h: ZA_FFT_LDS_POINTS; default 1024 + slicing)
              "fx_fftkat_full": ["-DZA_FFT_LDS_POINTS=4096"], "fx_fftbench_full": ["-DZA_FFT_LDS_POINTS=4096"]}
# Compiler-hazard rule (DESIGN.md "Compiler hazards"): twice the parity tests caught hipcc 7.2 storing ONE script variable from a
# register pair whose high half a temporary had taken over, both times in a kernel at the 512-register ceiling with kilobytes of
# spills (RTT in round 2; TextureXY in round 4: `attack_sc` came back as 0x00000000_b9d438c5 for 0x3fc272ad_b9d438c5, low dword right,
# high dword zero, while every value computed FROM it was right; tests/test_tpar.py ...long_run...[TextureXY+IR]). The same text
# compiled with the scheduler's GCN pressure trackers is correct. That is a perturbation, not a fix -- a scheduler option cannot
# repair an allocator -- so it is not on everywhere (measured on every module: 690 tests green, NeuroCV, Texture, ERBTilt 6-9 %
# faster, TSEQ 11 %, SOMA, BedRock, DPT 5-6 % slower, 3DPanner's time-parallel kernel past the long-branch limit); it goes where the
# fault was seen: a module whose generic process kernel sits at the ceiling and spills HAZARD_SPILL_BYTES or more per lane is
# compiled again with it (TextureXY 8.9 KB, Texture 10.5 KB today; the next heaviest, 3DPanner, 7.3 KB). The guard stays what it was: every variable of
# every leaf is compared after processing.
GCN_TRACKERS = ["-mllvm", "-amdgpu-use-amdgpu-trackers=1"]
HAZARD_SPILL_BYTES = int(os.environ.get("ZA_HAZARD_SPILL_BYTES", "8192"))


def ceiling_spill_bytes(so: Path) -> int:
    """Private segment (spills + scratch arrays, bytes per lane) of a built module's generic process kernel if it uses all 512
    registers (256 VGPRs + 256 AGPRs); 0 otherwise."""
    import re
    import tempfile
    llvm = Path("/opt/rocm/lib/llvm/bin")
    try:
        with tempfile.TemporaryDirectory() as d:
            fat, co = os.path.join(d, "fat.bin"), os.path.join(d, "dev.co")
            subprocess.run([str(llvm / "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", str(so), fat], check=True)
            subprocess.run([str(llvm / "clang-offload-bundler"), "--unbundle", "--type=o", f"--input={fat}", f"--output={co}",
                            f"--targets=hipv4-amdgcn-amd-amdhsa--{ARCH}"], check=True, capture_output=True)
            notes = subprocess.run([str(llvm / "llvm-readelf"), "--notes", co], capture_output=True, text=True, check=True).stdout
    except (OSError, subprocess.CalledProcessError) as ex:          # (no disassembly tools: the rule cannot be evaluated -- say so, build on)
        print(f"warning: {so.name}: kernel resources not readable ({ex}); hazard rule skipped", file=sys.stderr)
        return 0
    worst = 0
    for m in re.finditer(r"\.name:\s+(\S+).*?\.private_segment_fixed_size:\s+(\d+).*?\.vgpr_count:\s+(\d+)", notes, re.S):
        if m.group(1).endswith("_process") and int(m.group(3)) >= 512:      # the generic process kernel: where the fault was seen
            worst = max(worst, int(m.group(2)))
    return worst


# leaves that could take a time-parallel kernel but keep the generic one, with the reason (none at present: a leaf whose @sample
# does nothing is recognised by the lowering itself)
NO_TPAR: Dict[str, str] = {}
LONG_BRANCH_LIMIT = int(os.environ.get("ZA_LONG_BRANCH_LIMIT", "32"))
# leaves whose state the hand-written kernel wants contiguous per instance
INSTANCE_MAJOR = {"DDT"}


def discover(plugins_root: Path) -> Dict[str, dict]:
    """{Key: {category, json, entry}} for every plugins/<Category>/<Key>/plugin.json (exactly two levels deep)."""
    out = {}
    for pj in sorted(Path(plugins_root).glob("*/*/plugin.json")):
        try:
            meta = json.loads(pj.read_text(encoding="utf-8"))
        except Exception:
            continue
        key = pj.parent.name
        ptype = str(meta.get("pluginType", "jsfx")).lower()
        entry = meta.get("entry")
        src = None
        if entry:
            cand = pj.parent / entry
            if cand.exists():
                src = cand
        if src is None:
            pat = "*.dsp" if ptype == "faust" else "*.jsfx"
            found = sorted((pj.parent / "src").glob(pat))
            src = found[0] if found else None
        out[key] = {"category": pj.parent.parent.name, "type": ptype, "meta": meta, "entry": src, "dir": pj.parent}
    return out


def matches(key: str, leaf: dict, needle: str) -> bool:
    """plugin_matches of the reference (scripts/pluginlib.py:243-257): case-insensitive substring of category, slug, name, key,
    directory, bundleId or clapId. A needle that equals a key exactly selects that leaf only (DDT, not also "DDT-anything")."""
    q = needle.strip().lower()
    if not q:
        return True
    meta = leaf.get("meta", {}) or {}
    hay = [leaf.get("category", ""), str(meta.get("slug", "")), str(meta.get("name", "")), key,
           str(Path(leaf.get("category", "")) / key), str(leaf.get("dir", "")), str(meta.get("bundleId", "")), str(meta.get("clapId", ""))]
    return any(q in str(h).lower() for h in hay)


def select(leaves: Dict[str, dict], needles: List[str]) -> List[str]:
    if not needles:
        return list(leaves)
    out = []
    for q in needles:
        exact = [k for k in leaves if k.lower() == q.strip().lower()]
        for k in (exact or [k for k, v in leaves.items() if matches(k, v, q)]):
            if k not in out:
                out.append(k)
    return out


def _run(cmd: List[str]):
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        tail = "\n".join((r.stdout + r.stderr).splitlines()[-40:])
        raise RuntimeError(f"command failed ({r.returncode}): {' '.join(cmd[:6])} ...\n{tail}")
    return r


def build_runtime(force=False) -> Path:
    LIB.mkdir(exist_ok=True)
    so = LIB / "libzabatch.so"
    srcs = [CSRC / "zabatch.hip", CSRC / "zab_module.h", PKG.parent / "include" / "zabatch.h"]
    if not force and so.exists() and all(so.stat().st_mtime >= s.stat().st_mtime for s in srcs):
        return so
    _run([HIPCC] + HIP_FLAGS + ["-o", str(so), str(CSRC / "zabatch.hip"), "-ldl"])
    return so


def _hot_calls(prog) -> set:
    """Builtins reachable from @sample or @block (through user functions, which are specialised per caller section)."""
    from . import syntax as S
    seen, out, todo = set(), set(), list(prog.sections.get("sample", [])) + list(prog.sections.get("block", []))
    while todo:
        n = todo.pop()
        if isinstance(n, S.Call):
            out.add(n.fn)
            if n.fn in prog.fns and n.fn not in seen:
                seen.add(n.fn)
                todo.append(prog.fns[n.fn])
        todo.extend(S.children(n))
    return out


def tpar_plan(unit: codegen.Unit):
    """Time-parallel plan of the leaf's @sample (zajit/tpar.py), or (None, reason). Leaves with a hand-written kernel keep it."""
    from . import tpar
    if os.environ.get("ZA_NO_TPAR"):
        return None, "disabled (ZA_NO_TPAR)"
    if unit.prog.name in FAST_KERNELS:
        return None, "hand-written kernel"
    if unit.prog.name in NO_TPAR:
        return None, NO_TPAR[unit.prog.name]
    memo = getattr(unit, "_tpar_plan", None)           # (asked for by module_source and by the metadata: one analysis per unit)
    if memo is None:
        memo = unit._tpar_plan = tpar.try_plan(unit.prog, int(unit.defines["ZA_NCH"]))
    return memo


def long_branches(so: Path, kernel: str) -> int:
    """Long-branch expansions (s_getpc / s_setpc through a scratch register pair) inside one kernel of a built module. A branch
    needs them when its target is more than 2^17 bytes away, i.e. in very large kernels. CMD's time-parallel kernel, while its
    address pass wrote out 325 pair tests inline (90 000 instructions, 490 expansions, 12 000 scalar-register spills), faulted
    on the device with an access beyond the largest legal address -- same source text as the 8 000-instruction form that runs;
    TSEQ's runs with 4 - 17 of them and passes every parity test. Kernels with more than LONG_BRANCH_LIMIT are not trusted: the
    leaf is rebuilt without its time-parallel kernel (build_module) and says so in its metadata."""
    import re
    import tempfile
    llvm = Path("/opt/rocm/lib/llvm/bin")
    with tempfile.TemporaryDirectory() as d:
        fat, co = os.path.join(d, "fat.bin"), os.path.join(d, "dev.co")
        subprocess.run([str(llvm / "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", str(so), fat], check=True)
        subprocess.run([str(llvm / "clang-offload-bundler"), "--unbundle", "--type=o", f"--input={fat}", f"--output={co}",
                        f"--targets=hipv4-amdgcn-amd-amdhsa--{ARCH}"], check=True, capture_output=True)
        dis = subprocess.run([str(llvm / "llvm-objdump"), "-d", f"--disassemble-symbols={kernel}", co], capture_output=True, text=True).stdout
    return sum(1 for m in re.finditer(r"s_setpc_b64 s\[(\d+):", dis) if m.group(1) != "30")


def module_source(unit: codegen.Unit) -> str:
    p = unit.prog
    name = p.name
    nch = int(unit.defines["ZA_NCH"])
    plan, _why = tpar_plan(unit)
    alias = " ".join(f"X({k}, {p.vars[v]})" for k, v in sorted(p.aliases.items()) if v in p.vars)
    from .emit import FFT_CALLS
    fft_hot = bool(_hot_calls(p) & set(FFT_CALLS)) or (unit.defines.get("ZA_USES_COOP") == "1" and "gmem" not in unit.features)   # replica lanes
    lines = [f"// generated by zajit.build for leaf {name}; do not edit", unit.preamble()
             + ("#define ZA_MEM_STRIDE1 1\n" if fft_hot else ""),      # arenas contiguous per instance, always (zab_generic.hip.h)
             f"#define ZA_KERNEL(x) zab_{_cid(name)}_##x",
             f"#define ZA_FOR_CH(X) {' '.join(f'X({c})' for c in range(nch))}",
             f"#define ZA_FOR_ALIAS(X) {alias}",
             '#include "zart.h"']
    if "fft" in unit.features:
        lines.append('#include "zart_fft.h"')
    if "gmem" in unit.features:
        lines.append('#include "zart_gmem.h"')
    if "pool" in unit.features:
        lines.append('#include "zart_pool.h"')
    if "file" in unit.features:
        lines.append('#include "zart_file.h"')
    if "msg" in unit.features:
        lines.append('#include "zart_msg.h"')
    lines.append(unit.code)
    lines.append('#include "zab_generic.hip.h"')
    if name in FAST_KERNELS:      # hand-written kernels address state by name
        for vn, vi in sorted(p.vars.items(), key=lambda kv: kv[1]):
            lines.append(f"#define ZA_VAR_{_cid(vn)} {vi}")
    fast = FAST_KERNELS.get(name)
    if fast:
        lines.append(f'#include "{fast}"')
    elif plan is not None:
        from . import tpar
        lines.append('#include "zart_tpar.h"')
        lines.append(tpar.emit_hip(plan, p))
        lines.append(f'#define ZA_FAST_KERNEL_NAME "zab_{_cid(name)}_tpar"')
        fast = "tpar"
    names = sorted(p.vars.items(), key=lambda kv: kv[1])
    arr = ", ".join(json.dumps(n) for n, _ in names) or '""'
    lines.append(f"static const char* const za_var_names[] = {{ {arr} }};")
    lines.append("static const ZabModule za_module = {")
    lines.append(f"  ZAB_MODULE_ABI, {json.dumps(name)}, ZA_NV, ZA_NCH, {p.io['inputs']}, {p.io['outputs']},")
    lines.append("  ZA_HAS_INIT, ZA_HAS_SLIDER, ZA_HAS_BLOCK, ZA_HAS_SAMPLE,")
    # leaves with FFT builtins on the audio path (@sample / @block): the wave-cooperative transforms stream one instance's
    # buffer with all lanes, so it must be contiguous (a leaf that only transforms in @init/@slider stays interleaved)
    # (2 = instance-major AND thin wavefronts with replica lanes while the batch is small, see zabatch.hip / zab_generic)
    # leaves whose time-parallel kernel reads delay lines: 64 consecutive frames of ONE instance per access, so its arena must be
    # contiguous (interleaved, those 64 reads would touch 64 cache lines)
    tp_mem = plan is not None and bool(plan.loads or plan.stores or any(L.cells for L in plan.loops))
    # (3 = thin wavefronts like 2, but the kernel has no replica lanes: gmem users, zab_generic.hip.h ZA_REPLICAS)
    lines.append(f"  {(3 if 'gmem' in unit.features else 2) if fft_hot else (1 if (name in INSTANCE_MAJOR or tp_mem) else 0)}, 65536, za_var_names, {2 * 32768 if 'fft' in unit.features else 0}, {(2 if p.options.get('gmem') else 1) if 'gmem' in unit.features else 0}, {1 if 'pool' in unit.features else 0}, {1 if 'file' in unit.features else 0}, {(2 if 'msgbuf' in unit.features else 1) if 'msg' in unit.features else 0},")
    lines.append("  za_launch_prepare, za_launch_process, za_launch_slider,")
    if fast:
        lines.append("  za_fast_applies, za_launch_fast, ZA_FAST_KERNEL_NAME,")
    else:
        lines.append("  nullptr, nullptr, nullptr,")
    lines.append(f'  "zab_{_cid(name)}_process", za_launch_section, {"za_launch_msg_flush" if "msg" in unit.features else "nullptr"}, ZA_USES_LMEM }};')
    lines.append('extern "C" const ZabModule* zab_module_get(void) { return &za_module; }')
    return "\n".join(lines) + "\n"


def _cid(name: str) -> str:
    s = "".join(ch if ch.isalnum() else "_" for ch in name)
    return "_" + s if s[0].isdigit() else s


def build_module(jsfx_path, name: Optional[str] = None, force=False, verbose=False) -> Path:
    prog = program.analyse_file(jsfx_path)
    if name:
        prog.name = name
    unit = codegen.make_unit(prog)
    LIB.mkdir(exist_ok=True)
    GEN.mkdir(exist_ok=True)
    src = GEN / f"{prog.name}_module.hip"
    so = LIB / f"libzab_{prog.name}.so"
    leaf_flags = list(LEAF_FLAGS.get(prog.name, []))
    if unit.defines.get("ZA_USES_FFT") == "1" and unit.defines.get("ZA_OUTLINE_FNS") != "1" and not os.environ.get("ZA_FFT_CALLS"):
        # FFT leaves: the transform code inlined into its kernels, so that the kernels' register cap (ZA_OCC: two wavefronts per
        # SIMD) covers it -- a function that is called keeps its own, larger allocation and the kernel inherits it
        leaf_flags.append("-DZA_INLINE_ALL")
    base_sha = hashlib.sha1((module_source(unit) + " ".join(leaf_flags)).encode()).hexdigest()
    tr_note = LIB / f"{prog.name}.trackers"        # "<bytes>\n<sha1 of the module text the hazard rule fired on>"
    if os.environ.get("ZA_GCN_TRACKERS") or (tr_note.exists() and tr_note.read_text().split()[-1:] == [base_sha]):
        leaf_flags += GCN_TRACKERS
    text = module_source(unit) + (f"// leaf build flags: {' '.join(leaf_flags)}\n" if leaf_flags else "")
    deps = [CSRC / "zart.h", CSRC / "zab_generic.hip.h", CSRC / "zab_module.h"]
    if "zart_tpar.h" in text:
        deps.append(CSRC / "zart_tpar.h")
    if prog.name in FAST_KERNELS:
        deps.append(CSRC / FAST_KERNELS[prog.name])
        deps += [CSRC / d for d in FAST_KERNEL_DEPS.get(prog.name, [])]
    for extra in ("zart_fft.h", "zart_gmem.h", "zart_pool.h", "zart_file.h", "zart_msg.h"):
        if f'#include "{extra}"' in text:      # (only the leaves that include a runtime header are stale when it changes)
            deps.append(CSRC / extra)
    lb_note = LIB / f"{prog.name}.longbranch"      # "<count>\n<sha1 of the module text whose time-parallel kernel was refused>"
    allow_lb = bool(os.environ.get("ZA_TPAR_ALLOW_LONG_BRANCHES"))
    deps_newer = lambda: any(so.stat().st_mtime < d.stat().st_mtime for d in deps)
    text_sha = hashlib.sha1(text.encode()).hexdigest()
    refused = None
    if lb_note.exists() and not allow_lb and not force:
        note = lb_note.read_text().split()
        if len(note) == 2 and note[1] == text_sha:
            refused = note[0]           # the same text was compiled, counted and refused before: do not compile it again
    if refused is None:
        stale = force or not so.exists() or not src.exists() or src.read_text() != text or deps_newer()
        if stale:
            src.write_text(text)
            t0 = time.time()
            _run([HIPCC] + HIP_FLAGS + leaf_flags + ["-I", str(CSRC), "-o", str(so), str(src)])
            if verbose:
                print(f"  hipcc {prog.name}: {time.time() - t0:.1f}s")
            if "-amdgpu-use-amdgpu-trackers=1" not in leaf_flags and prog.name not in FAST_KERNELS:
                spill = ceiling_spill_bytes(so)
                if spill >= HAZARD_SPILL_BYTES:            # the hazard rule (above): once more, with the GCN trackers
                    tr_note.write_text(f"{spill}\n{base_sha}")
                    leaf_flags += GCN_TRACKERS
                    text = module_source(unit) + f"// leaf build flags: {' '.join(leaf_flags)}\n"
                    text_sha = hashlib.sha1(text.encode()).hexdigest()
                    src.write_text(text)
                    _run([HIPCC] + HIP_FLAGS + leaf_flags + ["-I", str(CSRC), "-o", str(so), str(src)])
                    if verbose:
                        print(f"  hipcc {prog.name}: {spill} B of spills at the register ceiling -> rebuilt with the GCN trackers")
            lb_note.unlink(missing_ok=True)
            if "ZA_FAST_KERNEL_NAME \"zab_" in text and "_tpar\"" in text and prog.name not in NO_TPAR:
                nlb = long_branches(so, f"zab_{_cid(prog.name)}_tpar")
                if nlb > LONG_BRANCH_LIMIT and not allow_lb:      # see long_branches(): the generic kernel stays the leaf's only one
                    lb_note.write_text(f"{nlb}\n{text_sha}")
                    refused = str(nlb)
    if refused is not None:
        NO_TPAR[prog.name] = f"the device compiler needed {refused} long-branch expansions in the time-parallel kernel (not trusted)"
        text = module_source(unit)
        if force or not so.exists() or not src.exists() or src.read_text() != text or deps_newer():
            src.write_text(text)
            _run([HIPCC] + HIP_FLAGS + leaf_flags + ["-I", str(CSRC), "-o", str(so), str(src)])
    meta = unit.meta()
    meta["sliders"] = {str(i): {"default": d.default, "min": d.vmin, "max": d.vmax, "step": d.step,
                                "is_choice": d.is_choice, "is_string": d.is_string, "var": d.var_name, "label": d.label}
                       for i, d in prog.slider_decls.items()}
    from .sliders import default_slider_values
    meta["default_sliders"] = [float(x) for x in default_slider_values(prog.slider_decls)]
    plan, why = tpar_plan(unit)
    meta["fast_path"] = prog.name in FAST_KERNELS or plan is not None
    meta["tpar"] = plan.stats if plan is not None else {"unsupported": why}
    (LIB / f"{prog.name}.json").write_text(json.dumps(meta))
    return so


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(description="Build MI355X plugin modules from JSFX leaves")
    ap.add_argument("--plugins-root", default="/root/reference/plugins")
    ap.add_argument("--only", action="append", default=[],
                    help="build only matching leaves: category, key, slug, name, path, bundleId or clapId (repeatable / comma separated)")
    ap.add_argument("--correctness-check", action="store_true",
                    help="run every built leaf on the device against the reference shadow VM's recorded results (needs a GPU)")
    ap.add_argument("--list", action="store_true")
    ap.add_argument("--force", action="store_true")
    ap.add_argument("--keep-going", action="store_true")
    args = ap.parse_args(argv)
    leaves = discover(Path(args.plugins_root))
    if args.list:
        for k, v in leaves.items():
            print(f"{v['category']}/{k}\t{v['type']}\t{v['entry']}")
        return 0
    want = select(leaves, [w for arg in args.only for w in arg.split(",") if w])
    if args.only and not want:
        print(f"no leaf matches {args.only}", file=sys.stderr)
        return 2
    build_runtime(force=args.force)
    rc = 0
    checked: List[dict] = []
    for key, leaf in leaves.items():
        if key not in want:
            continue
        if leaf["entry"] is None:
            continue
        try:
            if leaf["type"] == "faust":             # hand-written HIP restatement + mydsp adapter (zajit/faust.py)
                from . import faust
                so = faust.build_faust_module(leaf["entry"], key, force=args.force, verbose=True)
                faust.adapter_header(key, leaf["entry"])
            else:
                so = build_module(leaf["entry"], name=key, force=args.force, verbose=True)
            print(f"built {so}")
            if args.correctness_check and leaf["type"] != "faust":
                from . import check
                rows = check.check_leaf(key)
                checked += rows
                if not all(r["ok"] for r in rows):
                    raise RuntimeError(f"{key}: correctness check above the reference's tolerances (audio 1e-5, vars / mem 1e-8)")
            elif args.correctness_check:
                print(f"  correctness {key}: Faust leaf -- the reference has no shadow runtime for FaustJuceProcessor; nothing to compare")
        except Exception as ex:  # noqa: BLE001
            print(f"FAILED {key}: {str(ex)[:2000]}", file=sys.stderr)
            rc = 1
            if not args.keep_going:
                return rc
    if args.correctness_check:
        bad = [r for r in checked if not r["ok"]]
        print(f"correctness check: {len(checked) - len(bad)} of {len(checked)} (case, kernel) runs within the reference's tolerances")
    return rc


if __name__ == "__main__":
    sys.exit(main())
