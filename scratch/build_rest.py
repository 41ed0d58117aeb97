import sys, time; sys.path.insert(0,'zorakaudio-experimental-plugins_amd'); sys.path.insert(0,'.')
from zajit import build
from pathlib import Path
from oracle import port
l = build.discover(Path('/root/reference/plugins'))
rest = ['IPCProbeA','IPCProbeB','GesturePad','3DPannerManager','PsychoConvolver','CMD','Contour','TextureXY','3DPanner','Texture','Sample']
for k in rest:
    t=time.time()
    try:
        build.build_module(l[k]['entry'], name=k); t1=time.time()-t
        t=time.time(); port.build_port(l[k]['entry'], k)
        print(k, f'hip {t1:.0f}s port {time.time()-t:.0f}s', flush=True)
    except Exception as ex:
        print(k, 'FAILED', type(ex).__name__, str(ex)[-1500:], flush=True)
