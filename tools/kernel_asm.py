"""rebuild a leaf module and dump the device asm of one kernel: python tools/kernel_asm.py DDT zab_ddt_fast /tmp/k1.s"""
import sys, subprocess
sys.path.insert(0, str(__import__('pathlib').Path(__file__).resolve().parent.parent / 'zorakaudio-experimental-plugins_amd'))
from zajit import build
from pathlib import Path
leaf, kern, out = sys.argv[1], sys.argv[2], sys.argv[3]
# (Faust leaves: their kernels are templates, so `kern` is matched as a substring of the mangled name)
from zajit import faust
l = build.discover(Path("/root/reference/plugins")); l.update({p.stem if p.stem[:3] == "fx_" else "fx_" + p.stem: {"entry": p} for p in (Path(__file__).resolve().parent.parent / "tests/fixtures").glob("*.jsfx")})
if leaf in faust.FAUST_LEAVES:
    faust.build_faust_module(l[leaf]['entry'], leaf, force=True)
else:
    build.build_module(l[leaf]['entry'], name=leaf, force=True)
flags = [f for f in build.HIP_FLAGS if f not in ("-shared", "-fPIC")]
r = subprocess.run([build.HIPCC] + flags + ["-I", str(build.CSRC), "--cuda-device-only", "-S", "-o", "/tmp/_all.s",
                    str(build.GEN / f"{leaf}_module.hip")], capture_output=True, text=True)
assert r.returncode == 0, r.stderr[-2000:]
L = open("/tmp/_all.s").read().splitlines()
i0 = [i for i, x in enumerate(L) if x.startswith(kern + ":") or (kern in x.split(":")[0] and ":" in x and not x.lstrip().startswith((".", ";")))][0]
i1 = [i for i, x in enumerate(L) if i > i0 and "s_endpgm" in x][0]
open(out, "w").write("\n".join(L[i0:i1 + 1]))
print(out, i1 - i0, "lines")
