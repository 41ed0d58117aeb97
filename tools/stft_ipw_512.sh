#!/bin/bash
# is one instance per wavefront better for an FFT leaf when every wavefront is resident? (512 instances)
for ipw in 1 2; do for leaf in fx_stft DOT PsychoConvolver; do
  echo "== $leaf ZAB_IPW=$ipw"; ZAB_IPW=$ipw python bench.py --leaf $leaf --instances-total 512 --frames 16384 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"
done; done > gpurun_out/stft_ipw_512.log 2>&1
