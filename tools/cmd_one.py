import sys
sys.path[:0]=['zorakaudio-experimental-plugins_amd','.']
import numpy as np, zabatch
from zajit import noise
leaf=sys.argv[1]
meta = zabatch.leaf_meta(leaf)
x = noise.white_noise([3], 2048)
with zabatch.Engine(leaf, 1, mem_cap=1 << 20) as e:
    e.set_sliders(meta["default_sliders"]); e.prepare()
    y = e.process_host(x, block=512)
    print(leaf, 'ok', float(np.abs(y).max()), e.last_kernel_name())
