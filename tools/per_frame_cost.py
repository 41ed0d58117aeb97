import sys; from pathlib import Path
ROOT = Path("/root/repo") if Path("/root/repo").exists() else Path.cwd()
sys.path[:0] = [str(ROOT / "zorakaudio-experimental-plugins_amd"), str(ROOT)]
import zabatch, numpy as np
for leaf, n, sl in (("fx_opbench", 256, [0, 0]), ("fx_opbench", 1024, [0, 0]), ("fx_stft", 1024, None), ("fx_stft", 256, None)):
    for block in (512, 16384):
        with zabatch.Engine(leaf, n, mem_cap=1 << 15 if leaf == "fx_opbench" else 0, max_block=block) as e:
            row = np.array(zabatch.leaf_meta(leaf)["default_sliders"], dtype=np.float64)
            if sl is not None: row[:len(sl)] = sl
            e.set_sliders(row); e.prepare()
            frames = 16384
            nb = n * e.nch * frames * 4
            di, do = e.device_alloc(nb), e.device_alloc(nb)
            e.device_noise(di, frames)
            for _ in range(2): e.process_device(di, do, frames, block=block); e.sync()
            ms, launches = e.last_timing()
            print(f"{leaf} n={n} block={block}: {ms:.2f} ms, {launches} launches, ipw {e.launch_shape()[0]}, {ms / frames * 1e3:.3f} us per frame", flush=True)
