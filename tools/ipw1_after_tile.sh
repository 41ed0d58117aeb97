#!/bin/bash
# one vs two instances per wavefront after the replica-lane tile went dynamic: FFT leaves (tools/fft_bench.py sizes) and the
# accumulation-loop leaves of the catalog, plus parity at ZAB_IPW=1
for ipw in 1 2; do echo "== ZAB_IPW=$ipw"; ZAB_IPW=$ipw python tools/fft_bench.py 2>&1 | grep -v "_full"; done > gpurun_out/ipw1_fft.log 2>&1
for ipw in 1 2; do echo "== ZAB_IPW=$ipw"; ZAB_IPW=$ipw python tools/catalog_sweep.py --only TSEQ,SpectralStabilizer,Texture,Contour,DOT,PsychoConvolver 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        r = json.loads(l); print(r['leaf'], r['instances'], round(r.get('kernel_ms', 0), 2), 'ms  ipw', r.get('instances_per_wave'), 'lds', r.get('lds_mem_words'))
"; done > gpurun_out/ipw1_sweep.log 2>&1
ZAB_IPW=1 python -m pytest tests/test_catalog_gpu.py tests/test_fft_builtins.py -m gpu -q > gpurun_out/ipw1_tests.log 2>&1
