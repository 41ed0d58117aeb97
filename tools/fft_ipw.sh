#!/bin/bash
# FFT builtins and the leaves that use them against the wavefront target of replica-lane FFT leaves (ZAB_FFT_WAVES)
python -m pytest tests/test_fft_builtins.py -m gpu -x -q > gpurun_out/fft_tests.log 2>&1 || { tail -n 20 gpurun_out/fft_tests.log; exit 1; }
for w in 256 1024 2048 4096; do echo "== ZAB_FFT_WAVES=$w"; ZAB_FFT_WAVES=$w python tools/fft_bench.py 2>&1 | grep -v "_full"; done > gpurun_out/fft_bench.log 2>&1
python tools/fft_bench.py 2>&1 | grep "_full" >> gpurun_out/fft_bench.log
