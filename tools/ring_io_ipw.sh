#!/bin/bash
# per-sample arena access cost at 1 and 2 instances per wavefront, 256 and 1024 instances
for n in 256 1024; do for ipw in 1 2; do echo "== instances $n ZAB_IPW=$ipw"; ZAB_IPW=$ipw python tools/ring_io.py $n; done; done > gpurun_out/ring_io_ipw.log 2>&1
