#!/bin/bash
# ClickBeGoneSG: the four-wavefront kernel against the wave-per-instance kernel (its own choice of G) over the batch size.
O=gpurun_out; mkdir -p $O; rm -f $O/cbg_sweep.txt
for n in 256 1024 1022 2048 4096 8192 16384; do for k in quad wave; do
  echo "cbg N=$n $k: $(ZAB_CBG_KERNEL=$k timeout -k 10 120 python bench.py --leaf ClickBeGoneSG --instances-total $n --frames 48000 --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["roofline"]["kernel"], round(d["roofline"]["kernel_ms"],3), "ms", round(d["roofline"]["frac"],4), "null", d["null_test_dbfs"])')" | tee -a $O/cbg_sweep.txt
done; done
