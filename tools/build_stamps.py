"""Build <LEAF>_stamps: the leaf with the in-kernel phase clock of its time-parallel kernel compiled in (ZA_TPAR_STAMPS).
usage: python tools/build_stamps.py LEAF [LEAF ...]     LEAF = a catalog key (needs /root/reference) or fx_<fixture>
Read the clock with tools/tpar_stamps.py <LEAF>_stamps. The _stamps modules are scratch: delete them from lib/ afterwards."""
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT / "zorakaudio-experimental-plugins_amd"), str(ROOT)]
os.environ["ZA_TPAR_STAMPS"] = "1"
from zajit import build as zb

leaves = None
for leaf in sys.argv[1:]:
    if leaf.startswith("fx_"):
        src = ROOT / "tests" / "fixtures" / f"{leaf[3:]}.jsfx"
    else:
        leaves = leaves or zb.discover(Path(os.environ.get("ZA_PLUGINS_ROOT", "/root/reference/plugins")))
        if leaf in leaves:
            src = leaves[leaf]["entry"]
        else:                                  # (a leaf the reference ships disabled: plugin.json.bak)
            src = sorted(Path(os.environ.get("ZA_PLUGINS_ROOT", "/root/reference/plugins")).glob(f"*/{leaf}/src/*.jsfx"))[0]
    print(zb.build_module(src, name=f"{leaf}_stamps", force=True))
