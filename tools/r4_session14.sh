#!/bin/bash
# round 4, GPU session 14: fft_real; fft_permute / fft_ipermute; ifft_real fused -- the new VM fixture on both kernels, the FFT and
# catalog suites, PsychoConvolver's times.
O=gpurun_out; mkdir -p $O
timeout -k 10 1000 python -m pytest tests/test_fft_builtins.py tests/test_catalog_gpu.py tests/test_tpar.py tests/test_file_slots.py -m gpu -q --maxfail=20 -p no:cacheprovider > $O/s14_tests.log 2>&1; echo "tests rc=$?" | tee $O/s14_summary.txt
tail -4 $O/s14_tests.log
timeout -k 10 500 python tools/fft_bench.py > $O/s14_fft_bench.txt 2>&1; echo "fft bench rc=$?" | tee -a $O/s14_summary.txt
tail -6 $O/s14_fft_bench.txt
