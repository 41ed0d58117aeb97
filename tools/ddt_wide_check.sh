#!/bin/bash
python -m pytest tests/test_ddt_gpu.py tests/test_fullsize_gpu.py -m gpu -x -q > gpurun_out/ddt.log 2>&1 || { tail -n 30 gpurun_out/ddt.log; exit 1; }
for n in 1024 4096; do for fr in 9600 48000 480000; do ZAB_DDT_KERNEL=wide python tools/quick_bench.py $n $fr wide; done; done > gpurun_out/qb.log 2>&1
python tools/quick_bench.py 4096 480000 auto >> gpurun_out/qb.log 2>&1
