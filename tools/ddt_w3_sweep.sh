#!/bin/bash
# DDT fast kernel: waves per SIMD (ZAB_DDT_MINW) x waves per instance (ZAB_DDT_NW) over the batch size; 96 000 frames unless FR is set.
O=gpurun_out; mkdir -p $O; out=$O/ddt_w3_sweep.txt; rm -f $out
FR=${FR:-96000}
for n in ${NS:-256 1024 2048 4096 8192}; do for mw in 2 3; do for nw in 2 4 8; do
  echo "DDT N=$n frames=$FR mw=$mw nw=$nw: $(ZAB_DDT_MINW=$mw ZAB_DDT_NW=$nw timeout -k 10 200 python bench.py --no-cpu-baseline --instances-total $n --frames $FR --steps 8 --warmup 2 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["roofline"]["kernel"], round(d["roofline"]["kernel_ms"],3), "ms", round(d["roofline"]["frac"],4))')" | tee -a $out
done; done; done
echo "default policy:" | tee -a $out
for n in ${NS:-256 1024 2048 4096 8192}; do
  echo "DDT N=$n frames=$FR auto: $(timeout -k 10 200 python bench.py --no-cpu-baseline --instances-total $n --frames $FR --steps 8 --warmup 2 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["roofline"]["kernel"], round(d["roofline"]["kernel_ms"],3), "ms", round(d["roofline"]["frac"],4))')" | tee -a $out
done
