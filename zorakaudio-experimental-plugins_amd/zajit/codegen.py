"""Translation-unit assembly: Program -> generated C++ text + metadata (shared by the HIP module build and the
CPU port build under oracle/)."""
from __future__ import annotations

import hashlib
import json
import sys
from dataclasses import dataclass, field
from typing import Dict, List

from .emit import Emitter
from .program import Program

sys.setrecursionlimit(max(sys.getrecursionlimit(), 20000))


@dataclass
class Unit:
    prog: Program
    code: str                 # template functions: fn_* and za_section_{init,slider,block,sample}
    defines: Dict[str, str]
    features: List[str]
    used_spl: List[int]
    used_sl: List[int]
    strings: List[str]

    def preamble(self) -> str:
        return "".join(f"#define {k} {v}\n" for k, v in self.defines.items())

    def meta(self) -> dict:
        p = self.prog
        return {
            "name": p.name,
            "nvars": p.nvars,
            "vars": p.vars,
            "io": p.io,
            "nch": int(self.defines["ZA_NCH"]),
            "memtop": p.memtop,
            "options": p.options,
            "has": {s: p.has(s) for s in ("init", "slider", "block", "sample")},
            "features": self.features,
            "aliases": {str(k): v for k, v in p.aliases.items()},
            "alias_var_index": {str(k): p.vars[v] for k, v in p.aliases.items() if v in p.vars},
            "strings": self.strings,
            "vars_sha1": hashlib.sha1(json.dumps(sorted(p.vars.items())).encode()).hexdigest(),
        }


# Scripts whose only shareable loops are elementwise ("map") loops get replica lanes exactly when they have a time-parallel kernel:
# there the wavefront IS the instance, so the section code it runs between the frames (@block, event frames) shares such loops
# over 64 lanes that would otherwise idle (measured: NeuroCV 78.8 -> 72.4 ms, Contour's generic kernel 108 -> 82 ms), while on
# the lane-per-instance kernel alone thin wavefronts for map loops did not pay (NeuroCV 109 -> 174 ms, TextureXY 478 -> 515 ms).


def make_unit(prog: Program) -> Unit:
    em = Emitter(prog)
    code = em.emit()
    from . import tpar
    import os
    nch_ = max(0, min(64, int(prog.io["process"])))
    tp_plan = (None, "disabled (ZA_NO_TPAR)") if os.environ.get("ZA_NO_TPAR") else tpar.try_plan(prog, nch_)
    used_spl = list(range(64)) if em.dyn_spl else sorted(em.used_spl)
    used_sl = list(range(64)) if em.dyn_sl else sorted(em.used_sl)
    # aliased sliders are written back to vars by the host sequence, so they must be resident too
    for k in prog.aliases:
        if k not in used_sl:
            used_sl.append(k)
    used_sl.sort()
    nch = max(1, min(64, int(prog.io["process"]))) if prog.io["process"] > 0 else 0
    for ch in range(nch):
        if ch not in used_spl:
            used_spl.append(ch)
    used_spl.sort()
    defines = {
        "ZA_NV": str(prog.nvars),
        "ZA_NCH": str(nch),
        "ZA_HAS_INIT": "1" if prog.has("init") else "0",
        "ZA_HAS_SLIDER": "1" if prog.has("slider") else "0",
        "ZA_HAS_BLOCK": "1" if prog.has("block") else "0",
        "ZA_HAS_SAMPLE": "1" if prog.has("sample") else "0",
        "ZA_FOR_USED_SPL(X)": " ".join(f"X({k})" for k in used_spl),
        "ZA_FOR_USED_SL(X)": " ".join(f"X({k})" for k in used_sl),
        "ZA_USES_RAND": "1" if "rand" in em.features else "0",
        "ZA_USES_SLIDERCHANGE": "1" if "sliderchange" in em.features else "0",
        "ZA_USES_GMEM": "1" if "gmem" in em.features else "0",
        "ZA_GMEM_AUTOATTACH": "1" if ("gmem" in em.features and prog.options.get("gmem")) else "0",
        "ZA_USES_POOL": "1" if "pool" in em.features else "0",
        "ZA_USES_FILE": "1" if "file" in em.features else "0",
        "ZA_USES_MSG": "1" if "msg" in em.features else "0",
        "ZA_USES_FFT": "1" if "fft" in em.features else "0",
        # has accumulation loops shared by replica lanes (emit.py) -- or only elementwise loops, where that was measured to pay
        "ZA_USES_COOP": "1" if ("coop" in em.features or ("coopmap" in em.features and tp_plan[0] is not None)) else "0",
        # leaves whose mem[] is touched only by zart.h's load/store/memset/memcpy can keep its low part in LDS (zart.h)
        "ZA_USES_LMEM": "1" if "mem" in em.features and not em.features & {"fft", "gmem", "pool", "file", "msg"} else "0",
        # very large scripts (Sample: 754 specialised functions, 1 MB of expressions) cannot be flattened into one kernel body
        # by the device compiler in reasonable time or memory: their user functions become real calls
        "ZA_OUTLINE_FNS": "1" if len(prog.fns) > 256 else "0",
        # a variable table this large lives in memory anyway: its section functions are real calls (csrc/zart.h ZA_SECTION_FN).
        # Not on top of outlined user functions (Sample): there the sections are small already, and the extra frames cost it
        # 10 % (3858 -> 4245 ms) and 37 KB of private segment per lane (the mixed-leaf run's scratch allocation failed)
        "ZA_BIG_STATE": "1" if prog.nvars > 1000 and len(prog.fns) <= 256 else "0",
        "ZA_MEMTOP": f"{float(prog.memtop)!r}",
    }
    unit = Unit(prog=prog, code=code, defines=defines, features=sorted(em.features), used_spl=used_spl,
                used_sl=used_sl, strings=list(em.strings))
    unit._tpar_plan = tp_plan          # (zajit/build.py tpar_plan: one analysis per unit)
    return unit
