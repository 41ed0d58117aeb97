import sys, os; sys.path[:0]=["zorakaudio-experimental-plugins_amd","."]
import zabatch, numpy as np
for ipw in (64, 16, 8, 4):
    os.environ["ZAB_IPW"]=str(ipw)
    out=[]
    n, size, K = 2048, 4096, 4
    with zabatch.Engine("fx_fftbench", n, mem_cap=1<<17) as e:
        row = np.zeros(64); row[0]=size; row[1]=K; row[2]=15
        e.set_sliders(row); e.prepare()
        frames = 64; nb = n*e.nch*frames*4
        di, do = e.device_alloc(nb), e.device_alloc(nb); e.device_noise(di, frames)
        for _ in range(2): e.process_device(di, do, frames, block=64); e.sync()
        out.append(f"fftbench {e.last_timing()[0]/K:.2f} ms/trip")
    for leaf, n, frames in (("fx_stft4k", 1024, 8192), ("DOT", 1024, 8192), ("PsychoConvolver", 1024, 8192)):
        with zabatch.Engine(leaf, n, mem_cap=(1<<22) if leaf=="PsychoConvolver" else 0) as e:
            e.set_sliders(zabatch.leaf_meta(leaf)["default_sliders"]); e.prepare()
            nb = n*2*frames*4
            di, do = e.device_alloc(nb), e.device_alloc(nb); e.device_noise(di, frames)
            for _ in range(2): e.process_device(di, do, frames); e.sync()
            out.append(f"{leaf} {e.last_timing()[0]:.1f} ms")
    print(f"ipw={ipw}: " + " | ".join(out), flush=True)
