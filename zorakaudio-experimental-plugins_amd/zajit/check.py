"""`--correctness-check` of the build driver: after a leaf is built, run it on the device against the reference shadow
runtime's recorded results and report what the reference's in-plugin monitor reports.

The reference compiles its WDL/EEL2 shadow VM into the plugin when scripts/build.py is given --correctness-check
(scripts/build.py:556,647; cmake/plugin/CMakeLists.txt:233) and compares compiled code and VM in lock step inside
processBlock (src/JSFXJuceProcessor.cpp:3589-3678): outputs within 1e-5 on float-cast samples, sliders / vars / touched
mem[] within 1e-8 (src/JSFXCorrectnessCheck.h:34-38), reported as max / RMS delta. A batch engine cannot host that VM next to
its kernels, so the same comparison runs here against the VM's recorded runs -- the golden vectors under tests/golden/
(inputs regenerated from their seed, outputs / final vars / touched mem[] / write high-water mark produced by the
reference's VM built from its own sources, tests/golden/make_golden.py) -- once per kernel the leaf has (the generated
time-parallel or hand-written kernel, and the generic one). This module reads fixtures and drives the engine; it does not
import the CPU oracles (tests/test_oracle_cpu.py compares against a LIVE VM where oracle/_ref is built: the fixtures are re-derived
there, and config C1 runs at its full 480 000 frames).
"""
from __future__ import annotations

import math
import os
from pathlib import Path
from typing import Dict, List

import numpy as np

AUDIO_EPS = 1.0e-5      # src/JSFXCorrectnessCheck.h:34
SCALAR_EPS = 1.0e-8     # :35

ROOT = Path(__file__).resolve().parent.parent.parent


def golden_dir() -> Path:
    return Path(os.environ.get("ZA_GOLDEN_DIR", ROOT / "tests" / "golden"))


def cases_of(leaf: str) -> List[str]:
    """Fixture cases '<leaf>_<case>.npz' of one leaf ('DDT_far_extreme' belongs to DDT; 'fx_stft4k_default' to fx_stft4k, not
    to fx_stft: the leaf of a fixture leaf is its first two words)."""
    out = []
    for p in sorted(golden_dir().glob(f"{leaf}_*.npz")):
        parts = p.stem.split("_")
        owner = "_".join(parts[:2]) if parts[0] == "fx" else parts[0]
        if owner == leaf:
            out.append(p.stem)
    return out


def dbfs(x: float) -> float:
    return 20.0 * math.log10(max(x, 1e-300))


def _state_delta(names, got, want):
    """Worst |delta| over the variables the VM knows (NaN in the fixture = the VM never created it; names that differ only in
    case are one variable in the VM and two in the compiled path: skipped, as in the parity tests)."""
    lower: Dict[str, int] = {}
    for n in names:
        lower[n.lower()] = lower.get(n.lower(), 0) + 1
    worst, who = 0.0, ""
    for n, a, b in zip(names, got, want):
        if lower[n.lower()] > 1 or np.isnan(b):
            continue
        if np.isnan(a) or np.isinf(a) or np.isinf(b):
            d = 0.0 if ((np.isnan(a) and np.isnan(b)) or a == b) else float("inf")
        else:
            d = abs(float(a) - float(b))
        if d > worst:
            worst, who = d, n
    return worst, who


def check_leaf(leaf: str, instances: int = 3, verbose: bool = True) -> List[dict]:
    """One row per (fixture case, kernel). Raises RuntimeError when the leaf has no fixture at all."""
    import zabatch
    from . import noise
    cases = cases_of(leaf)
    if not cases:
        raise RuntimeError(f"{leaf}: no reference-VM fixture under {golden_dir()} (tests/golden/make_golden.py makes them)")
    meta = zabatch.leaf_meta(leaf)
    rows = []
    for case in cases:
        g = np.load(golden_dir() / f"{case}.npz", allow_pickle=False)
        x1 = noise.white_noise([int(g["seed_instance"])], int(g["frames"]), channels=int(g["nch"]))[0]
        x = np.repeat(x1[None], instances, axis=0)
        mem_high = int(g["mem_high"])
        paths = [("generic", zabatch.ZAB_PATH_GENERIC)]
        if meta.get("fast_path"):
            paths.insert(0, ("fast", zabatch.ZAB_PATH_FAST))
        for label, sel in paths:
            with zabatch.Engine(leaf, instances, srate=float(g["srate"]), path=sel, mem_cap=max(65536, mem_high + 64)) as e:
                e.set_sliders(g["sliders"]); e.prepare()
                names = e.var_names()
                y = e.process_host(x, block=int(g["block"]))
                v = e.read_vars()
                kernel = e.last_kernel_name()
                high = e.mem_high()
                small = 0 < mem_high <= (1 << 22)
                mem = e.read_mem(0, mem_high) if small else None
            d = y.astype(np.float64) - g["out"].astype(np.float64)[None]
            row = {"leaf": leaf, "case": case[len(leaf) + 1:], "path": label, "kernel": kernel,
                   "max_dbfs": dbfs(float(np.abs(d).max())), "rms_dbfs": dbfs(float(np.sqrt((d * d).mean()))),
                   "audio_max": float(np.abs(d).max())}
            row["vars_worst"], row["vars_worst_name"] = max((_state_delta(names, v[i], g["vars"]) for i in (0, instances - 1)), key=lambda t: t[0])
            if mem is not None:
                want = np.zeros(mem_high); want[g["mem_idx"]] = g["mem_val"]
                row["mem_worst"] = float(np.abs(mem - want[None]).max())
            else:
                row["mem_worst"] = None                      # not compared (no arena use, or one too large to read back here)
            row["mem_high_ok"] = bool((high == mem_high).all())
            row["ok"] = bool(row["audio_max"] <= AUDIO_EPS and row["vars_worst"] <= SCALAR_EPS
                             and (row["mem_worst"] is None or row["mem_worst"] <= SCALAR_EPS) and row["mem_high_ok"])
            rows.append(row)
            if verbose:
                print(f"  correctness {leaf:>18s} {row['case']:<18s} {label:<7s} {kernel:<28s} audio max {row['max_dbfs']:8.1f} dBFS rms "
                      f"{row['rms_dbfs']:8.1f} dBFS | vars {row['vars_worst']:.2e} ({row['vars_worst_name'] or '-'}) | mem "
                      f"{'n/a (not compared)' if row['mem_worst'] is None else format(row['mem_worst'], '.2e')} | high-water {'ok' if row['mem_high_ok'] else 'DIFFERS'} | {'PASS' if row['ok'] else 'FAIL'}")
    return rows
