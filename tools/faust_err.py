import sys; sys.path.insert(0,'tests'); sys.path.insert(0,'.'); sys.path.insert(0,'zorakaudio-experimental-plugins_amd')
import numpy as np, test_faust as T, zabatch
fr = T._ref()
for leaf in ("ModTilt","VAR","RED","GTS","ClickBeGoneSG"):
    for path, pn in ((zabatch.ZAB_PATH_GENERIC,"generic"),(zabatch.ZAB_PATH_FAST,"fast")):
        n, frames = 16, 3000
        x = T._input(leaf, list(range(40, 40+n)), frames)
        rows = np.zeros((n,64)); rows[:, :len(T.FAUST[leaf])] = T.FAUST[leaf]
        rows[:,0] += np.linspace(0,1,n)
        try:
            with zabatch.Engine(leaf, n, path=path) as e:
                e.set_sliders(rows); e.prepare()
                y = e.process_host(x, block=512)
        except Exception as ex:
            print(leaf, pn, "skip", ex); continue
        worst = 0.0; nz = 0
        for i in range(0, n, 4 if leaf=="GTS" else 1):
            r = fr.FaustRef(leaf, 48000)
            want = r.compute(x[i], rows[i,:8].astype(np.float32))
            d = np.abs(y[i].astype(np.float64) - want)
            worst = max(worst, d.max()); nz += int((d>0).sum())
        print(f"{leaf:14s} {pn:8s} max err {worst:.3e}  samples differing {nz}")
