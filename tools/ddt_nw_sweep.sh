#!/bin/bash
# zab_ddt_fast: waves per instance (NW) against batch size
for n in 256 1024 2048 4096 8192; do for nw in 1 2 4 8; do ZAB_DDT_NW=$nw python tools/quick_bench.py $n 96000 nw$nw; done; done > gpurun_out/nw_sweep.log 2>&1
