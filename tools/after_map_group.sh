#!/bin/bash
# map loops with 8 instead of 4 trips per group: FFT-leaf harness, catalog rows with map loops, parity
set -e
python tools/fft_bench.py 2>&1 | grep -v fftbench > gpurun_out/mg8_fft_bench.log
python tools/catalog_sweep.py --only TSEQ,Texture,TextureXY,Contour,DOT,PsychoConvolver,NeuroCV,CMD > gpurun_out/mg8_sweep.log 2>&1
python -m pytest tests/test_catalog_gpu.py tests/test_fft_builtins.py -m gpu -q -x > gpurun_out/mg8_tests.log 2>&1
