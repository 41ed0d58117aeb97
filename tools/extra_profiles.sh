#!/bin/bash
# Round evidence beside the headline: rocprofv3 kernel stats of the FFT harness (config C3 ii), bench lines of C3 (i) and C5
TAG=${1:-rXX}
R=$(pwd); OUT=$R/gpurun_out
python -m pytest tests/test_ddt_gpu.py tests/test_group_gpu.py -m gpu -q -x > $OUT/${TAG}_ddt_group.log 2>&1 || exit 1
python bench.py --leaf fx_stft --instances-total 1024 --frames 16384 > $OUT/${TAG}_bench_stft.json 2> $OUT/${TAG}_bench_stft.err
python bench.py --leaf ClickBeGoneSG --instances-total 1024 --frames 48000 > $OUT/${TAG}_bench_cbg.json 2> $OUT/${TAG}_bench_cbg.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_fft_trace -- python3 $R/tools/leaf_scaling.py fx_fftbench 2048 --frames 64 > $OUT/${TAG}_fft_trace.log 2>&1
cd $R
f=$(ls $OUT/${TAG}_fft_trace/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp $f $OUT/${TAG}_fft_kernel_stats.csv
