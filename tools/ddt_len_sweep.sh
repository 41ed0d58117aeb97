#!/bin/bash
# zab_ddt_fast vs zab_ddt_wide against launch length
for n in 1024 4096; do for fr in 9600 48000 96000 144000 192000 288000; do ZAB_DDT_KERNEL=fast python tools/quick_bench.py $n $fr fast; ZAB_DDT_KERNEL=wide python tools/quick_bench.py $n $fr wide; done; done > gpurun_out/len_sweep.log 2>&1
