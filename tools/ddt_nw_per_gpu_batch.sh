#!/bin/bash
# the per-GPU batches of the strong-scaling bench (4096 / N instances, 480 000 frames): waves per instance
for n in 512 1024 2048; do for nw in auto 2 4 8; do
  if [ $nw = auto ]; then unset ZAB_DDT_NW; else export ZAB_DDT_NW=$nw; fi
  python bench.py --instances-total $n --no-cpu-baseline 2>/dev/null | python -c "
import json, sys; d = json.loads(sys.stdin.read()); print('N=$n nw=$nw', d['config']['kernel'], round(d['ms_per_step'], 3), 'ms', round(d['value'] / 1e3, 1), 'Gsamples/s', round(d['roofline']['frac'], 3))"
done; done > gpurun_out/ddt_nw_per_gpu_batch.log 2>&1
