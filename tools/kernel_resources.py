"""Registers and scratch of every kernel in the built leaf modules: python tools/kernel_resources.py [leaf ...]
(reads the code objects inside the built .so files; a process kernel with scratch > 0 is worth a look)"""
import sys, subprocess, re, tempfile, os
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "zorakaudio-experimental-plugins_amd"))
from zajit import build
LLVM = Path("/opt/rocm/lib/llvm/bin")
want = set(sys.argv[1:])
rows = []
for so in sorted(build.LIB.glob("libzab_*.so")):
    leaf = so.stem[len("libzab_"):]
    if want and leaf not in want: continue
    with tempfile.TemporaryDirectory() as d:
        fat, co = os.path.join(d, "fat.bin"), os.path.join(d, "dev.co")
        subprocess.run([str(LLVM / "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", str(so), fat], check=True)
        r = subprocess.run([str(LLVM / "clang-offload-bundler"), "--unbundle", "--type=o", f"--input={fat}", f"--output={co}",
                            "--targets=hipv4-amdgcn-amd-amdhsa--gfx950"], capture_output=True, text=True)
        if r.returncode: print(leaf, "unbundle failed", r.stderr[-200:]); continue
        notes = subprocess.run([str(LLVM / "llvm-readelf"), "--notes", co], capture_output=True, text=True).stdout
    for m in re.finditer(r"\.name:\s+(\S+).*?\.private_segment_fixed_size:\s+(\d+).*?\.vgpr_count:\s+(\d+)", notes, re.S):
        rows.append((leaf, m.group(1), int(m.group(2)), int(m.group(3))))
for leaf, k, scr, vg in rows:
    if "process" in k or "fast" in k or "wave" in k or "tpar" in k:
        print(f"{leaf:24s} {k:44s} scratch {scr:6d} B  vgprs {vg}")
