"""Print register / scratch / occupancy of a leaf module's kernels: python tools/kernel_resources.py DDT [filter]"""
import sys, re, subprocess
sys.path.insert(0, str(__import__('pathlib').Path(__file__).resolve().parent.parent / 'zorakaudio-experimental-plugins_amd'))
from zajit import build
leaf = sys.argv[1]; filt = sys.argv[2] if len(sys.argv) > 2 else ''
cmd = [build.HIPCC] + build.HIP_FLAGS + ["-I", str(build.CSRC), "-Rpass-analysis=kernel-resource-usage", "-o", "/tmp/kres_t.so",
       str(build.GEN / f"{leaf}_module.hip")]
r = subprocess.run(cmd, capture_output=True, text=True)
if r.returncode: print(r.stderr[-3000:]); sys.exit(1)
for b in r.stderr.split("Function Name:")[1:]:
    name = b.split()[0]
    if filt in name:
        g = lambda k: re.search(k + r": (\d+)", b).group(1)
        print(name, "VGPR", g("VGPRs"), "AGPR", g("AGPRs"), "SGPR", g("SGPRs"), "scratch", g(r"ScratchSize \[bytes/lane\]"),
              "occ", g(r"Occupancy \[waves/SIMD\]"))
