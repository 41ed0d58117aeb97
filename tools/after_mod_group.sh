#!/bin/bash
# after the modulo shortcut and the grouped accumulation loops: ring i/o cost, coop leaves, FFT harness, parity
set -e
python tools/ring_io.py 1024 > gpurun_out/mg_ring_io.log 2>&1
python tools/catalog_sweep.py --only TSEQ,DOT,SpectralStabilizer,Texture,ERBTilt,EasyExpander > gpurun_out/mg_sweep.log 2>&1
python tools/fft_bench.py > gpurun_out/mg_fft_bench.log 2>&1
python -m pytest tests/test_catalog_gpu.py tests/test_fft_builtins.py tests/test_fullsize_gpu.py -m gpu -q -x > gpurun_out/mg_tests.log 2>&1
