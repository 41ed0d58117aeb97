import sys; sys.path.insert(0,'zorakaudio-experimental-plugins_amd'); sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np, zabatch
from conftest import load_golden, golden_input
case = sys.argv[1]; n = int(sys.argv[2])
g = load_golden(case); x = np.repeat(golden_input(g)[None], n, axis=0)
outs = {}
for name, path in (("generic", 1), ("fast", 2)):
    with zabatch.Engine("DDT", n, path=path) as e:
        e.set_sliders(g["sliders"]); e.prepare()
        outs[name] = e.process_host(x, block=int(g["block"]))
for i in range(n):
    d = np.abs(outs["fast"][i].astype(np.float64) - outs["generic"][i])
    bad = np.argwhere(d > 1e-6)
    print(i, 'max diff', d.max(), 'first bad', bad[:3].tolist(), 'last bad', bad[-3:].tolist(), 'count', len(bad))
