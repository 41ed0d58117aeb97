"""Registers and scratch of every kernel in the built leaf modules: python tools/kernel_resources.py [leaf ...]
(reads the code objects inside lib/libzab_<leaf>.so; a process kernel whose scratch is far above a few hundred bytes either
spills -- very large scripts -- or keeps the state object in memory)"""
import sys, subprocess, re, tempfile, os
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "zorakaudio-experimental-plugins_amd"))
from zajit import build
LLVM = Path("/opt/rocm/lib/llvm/bin")


def kernel_resources(so):
    """[(kernel name, scratch bytes per lane, VGPRs + AGPRs)] of the gfx950 code object inside a built module."""
    with tempfile.TemporaryDirectory() as d:
        fat, co = os.path.join(d, "fat.bin"), os.path.join(d, "dev.co")
        subprocess.run([str(LLVM / "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", str(so), fat], check=True)
        subprocess.run([str(LLVM / "clang-offload-bundler"), "--unbundle", "--type=o", f"--input={fat}", f"--output={co}",
                        "--targets=hipv4-amdgcn-amd-amdhsa--gfx950"], check=True, capture_output=True)
        notes = subprocess.run([str(LLVM / "llvm-readelf"), "--notes", co], capture_output=True, text=True, check=True).stdout
    return [(m.group(1), int(m.group(2)), int(m.group(3)))
            for m in re.finditer(r"\.name:\s+(\S+).*?\.private_segment_fixed_size:\s+(\d+).*?\.vgpr_count:\s+(\d+)", notes, re.S)]


if __name__ == "__main__":
    want = set(sys.argv[1:])
    for so in sorted(build.LIB.glob("libzab_*.so")):
        leaf = so.stem[len("libzab_"):]
        if want and leaf not in want:
            continue
        for k, scr, vg in kernel_resources(so):
            if any(w in k for w in ("process", "fast", "wave", "tpar", "wide")):
                print(f"{leaf:24s} {k:44s} scratch {scr:6d} B  vgprs {vg}")
