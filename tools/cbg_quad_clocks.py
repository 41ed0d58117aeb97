"""Role clock of the four-wavefront ClickBeGoneSG kernel: builds the module with -DZF_QUAD_CLOCKS as leaf ClickBeGoneSG_clk
(scratch: delete lib/*ClickBeGoneSG_clk* afterwards) and runs one launch; workgroup 5 prints, per wavefront role
(0 = recursion 1, 1 = trigger / hold / mix, 2 and 3 = feed-forward front), the cycles it worked between the barriers.
usage: python tools/cbg_quad_clocks.py build | run [instances] [frames]"""
import json
import os
import shutil
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT / "zorakaudio-experimental-plugins_amd"), str(ROOT)]
KEY = os.environ.get("ZF_CLK_KEY", "ClickBeGoneSG_clk")


def build():
    from zajit import build as zb, faust as zf
    src = zb.GEN / f"{KEY}_module.hip"
    src.write_text(zf.module_source("ClickBeGoneSG").replace('"ClickBeGoneSG"', json.dumps(KEY)))
    so = zb.LIB / f"libzab_{KEY}.so"
    zb._run([zb.HIPCC] + zb.HIP_FLAGS + ["-DZF_QUAD_CLOCKS", "-I", str(zb.CSRC), "-o", str(so), str(src)])
    meta = json.loads((zb.LIB / "ClickBeGoneSG.json").read_text())
    meta["name"] = KEY
    (zb.LIB / f"{KEY}.json").write_text(json.dumps(meta))
    print(so)


def run(n=1024, frames=48000):
    import numpy as np
    os.environ["ZAB_CBG_KERNEL"] = "quad"
    import zabatch
    from zajit import noise
    x = noise.white_noise(range(8), frames, channels=2)
    x = np.tile(x, ((n + 7) // 8, 1, 1))[:n]
    with zabatch.Engine(KEY, n, path=zabatch.ZAB_PATH_FAST) as e:
        e.prepare()
        e.process_host(x, block=512)


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build()
    else:
        run(*[int(v) for v in sys.argv[2:4]])
