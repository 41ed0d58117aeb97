import sys; sys.path.insert(0,'zorakaudio-experimental-plugins_amd'); sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np, zabatch
from conftest import load_golden, golden_input
for case in sys.argv[1:]:
    g = load_golden(case); x = golden_input(g)[None]
    outs = {}
    for name, path in (("generic", 1), ("fast", 2)):
        with zabatch.Engine("DDT", 1, path=path) as e:
            e.set_sliders(g["sliders"]); e.prepare()
            outs[name] = e.process_host(x, block=int(g["block"]))
            v = e.read_vars()[0]; names = e.var_names()
            if name == "generic":
                print(case, {k: v[names.index(k)] for k in ("tapN","splitSamp","a_dir","a_early","a_late","wetp","dryp","out_gain")})
                tapN = int(v[names.index("tapN")])
                m = e.read_mem(32768, 64*6)[0]
                print(' dL', m[0:tapN].astype(int)); print(' dR', m[64:64+tapN].astype(int)); print(' D0', m[256:256+tapN].astype(int))
    d = np.abs(outs["fast"].astype(np.float64) - outs["generic"])[0]
    bad = np.argwhere(d > 1e-6)
    print(' max diff', d.max(), 'first bad', bad[:5].tolist(), 'count', len(bad), 'of', d.size)
    if len(bad):
        t = bad[0][1]; print(' around', t, outs["fast"][0,:,t-2:t+3], outs["generic"][0,:,t-2:t+3])
