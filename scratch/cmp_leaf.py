import sys
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/zorakaudio-experimental-plugins_amd')
from oracle import port, eel_oracle
from zajit import program, sliders, noise
import numpy as np, pathlib, glob
name = sys.argv[1]; frames = int(sys.argv[2]) if len(sys.argv)>2 else 4096
f = glob.glob(f'/root/reference/plugins/*/{name}/src/*.jsfx')[0]
port.build_port(f, name=name)
txt = program.expand_imports(pathlib.Path(f)); prog = program.analyse(txt, name)
dv = sliders.default_slider_values(prog.slider_decls)
nch = max(1, prog.io['process'])
x = noise.white_noise([0], frames, channels=nch)[0]
P = port.Port(name, 48000.0, mem_cap=1<<23); P.set_sliders(dv); P.prepare()
O = eel_oracle.EelOracle(txt, prog.aliases); O.set_sliders(dv); O.prepare(48000.0)
pv = P.vars()
bad=[(n, pv[i], O.var(n)) for n,i in P.meta['vars'].items() if O.var(n) is not None and not (abs(O.var(n)-pv[i])<=1e-8 or (np.isnan(pv[i]) and np.isnan(O.var(n))))]
print('after prepare: bad vars', len(bad), bad[:6], 'high', P.mem_high, O.mem_high)
yp = P.process(x, 512); yo = O.process(x, 512)
pv = P.vars()
bad=[(n, pv[i], O.var(n)) for n,i in P.meta['vars'].items() if O.var(n) is not None and not (abs(O.var(n)-pv[i])<=1e-8 or (np.isnan(pv[i]) and np.isnan(O.var(n))))]
d = np.abs(yp.astype(float)-yo)
print('dy', d.max(), 'first bad frame', (np.argwhere(d>1e-6)[:1]).tolist(), 'bad vars', len(bad), bad[:8])
hi=max(P.mem_high,O.mem_high,1); md=np.abs(P.mem(0,hi)-O.mem(0,hi)); print('dmem', md.max(), 'first', np.argwhere(md>1e-8)[:5].ravel().tolist(), 'high', P.mem_high, O.mem_high, 'err', P.err)
