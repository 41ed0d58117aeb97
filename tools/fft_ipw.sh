#!/bin/bash
# FFT builtins: GPU known answers, the leaves that use them, then the throughput harness
python -m pytest tests/test_fft_builtins.py tests/test_catalog_gpu.py -m gpu -x -q > gpurun_out/fft_tests.log 2>&1 || { tail -n 20 gpurun_out/fft_tests.log; exit 1; }
python tools/fft_bench.py > gpurun_out/fft_bench.log 2>&1
