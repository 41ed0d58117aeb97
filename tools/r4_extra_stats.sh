#!/bin/bash
# rocprofv3 --kernel-trace --stats of the other configurations' bench commands (C5 per-GPU batch and at 8192 instances, C3 i)
O=$(pwd)/gpurun_out; R=$(pwd); mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for cfg in "cbg1024 --leaf ClickBeGoneSG --instances-total 1024 --frames 48000" "cbg8192 --leaf ClickBeGoneSG --instances-total 8192 --frames 48000" "stft --leaf fx_stft --instances-total 1024 --frames 16384"; do
  set -- $cfg; tag=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/x_$tag -- python3 $R/bench.py --no-cpu-baseline "$@" > $O/x_${tag}_bench.json 2> $O/x_${tag}.err
  f=$(ls $O/x_$tag/*/*kernel_stats.csv | head -1); cp $f $O/x_${tag}_kernel_stats.csv; head -3 $f | cut -c1-140
done
