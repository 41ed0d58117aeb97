#!/bin/bash
# round 4, GPU session 13: FFT staging with unconditional (clamped) loads -- tests, harness, leaves.
O=gpurun_out; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_fft_builtins.py tests/test_catalog_gpu.py -m gpu -q --maxfail=20 -p no:cacheprovider > $O/s13_tests.log 2>&1; echo "tests rc=$?" | tee $O/s13_summary.txt
tail -4 $O/s13_tests.log
timeout -k 10 500 python tools/fft_bench.py > $O/s13_fft_bench.txt 2>&1; echo "fft bench rc=$?" | tee -a $O/s13_summary.txt
cat $O/s13_fft_bench.txt
timeout -k 10 300 python tools/stft_parts.py 1024 > $O/s13_stft_parts.txt 2>&1; cat $O/s13_stft_parts.txt
