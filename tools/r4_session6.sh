#!/bin/bash
# round 4, sixth GPU session: the two-stage sliced transforms (zart_fft.h): known answers, bits against the in-LDS build, the
# harness' times and its PMC traffic, the FFT-using leaves.
O=gpurun_out; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_fft_builtins.py -m gpu -q --maxfail=20 -p no:cacheprovider > $O/s6_fft_tests.log 2>&1; echo "fft tests rc=$?" | tee $O/s6_summary.txt
tail -5 $O/s6_fft_tests.log
FFT_BENCH_ONLY_KERNELS=1 timeout -k 10 300 python tools/fft_bench.py > $O/s6_fft_bench.txt 2>&1; echo "fft bench rc=$?" | tee -a $O/s6_summary.txt
cat $O/s6_fft_bench.txt
timeout -k 10 400 bash tools/fft_traffic.sh s6 > $O/s6_fft_traffic.log 2>&1; echo "traffic rc=$?" | tee -a $O/s6_summary.txt
tail -12 $O/s6_fft_traffic.log
timeout -k 10 400 python tools/fft_bench.py > $O/s6_fft_bench_full.txt 2>&1; echo "fft bench leaves rc=$?" | tee -a $O/s6_summary.txt
tail -6 $O/s6_fft_bench_full.txt
