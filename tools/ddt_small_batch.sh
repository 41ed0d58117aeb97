#!/bin/bash
for n in 512 1024 2048 4096; do python tools/quick_bench.py $n 480000 auto; done > gpurun_out/ss.log 2>&1
for nw in 2 4 8; do ZAB_DDT_NW=$nw python tools/quick_bench.py 512 480000 nw$nw; done >> gpurun_out/ss.log 2>&1
