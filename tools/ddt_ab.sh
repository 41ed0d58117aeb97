#!/bin/bash
# DDT kernel check + timing over the waves-per-instance variants (one gpurun step); optional second module build in build_ab/libzab_DDT_b.so
python -m pytest tests/test_ddt_gpu.py -m gpu -x -q > gpurun_out/ddt.log 2>&1 || { tail -n 30 gpurun_out/ddt.log; exit 1; }
for nw in 1 2 4; do ZAB_DDT_NW=$nw python tools/quick_bench.py 4096 480000 a_nw$nw; done > gpurun_out/qb.log 2>&1
python tools/quick_bench.py 1024 480000 a_auto >> gpurun_out/qb.log 2>&1
if [ -f build_ab/libzab_DDT_b.so ]; then
  cp lib/libzab_DDT.so /tmp/a.so && cp build_ab/libzab_DDT_b.so lib/libzab_DDT.so
  python -m pytest tests/test_ddt_gpu.py -m gpu -x -q -k fast > gpurun_out/ddt_b.log 2>&1 || tail -n 30 gpurun_out/ddt_b.log
  for nw in 1 2 4; do ZAB_DDT_NW=$nw python tools/quick_bench.py 4096 480000 b_nw$nw; done >> gpurun_out/qb.log 2>&1
  python tools/quick_bench.py 1024 480000 b_auto >> gpurun_out/qb.log 2>&1
  cp /tmp/a.so lib/libzab_DDT.so
fi
