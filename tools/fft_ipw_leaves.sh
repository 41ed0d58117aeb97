#!/bin/bash
for ipw in 1 2 4; do echo "== ZAB_IPW=$ipw"; ZAB_IPW=$ipw python tools/fft_bench.py 2>&1 | grep -v "_full"; done > gpurun_out/ipw1.log 2>&1
