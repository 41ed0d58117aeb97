#!/bin/bash
# replica-lane leaves at 256 instances: 1 vs 2 instances per wavefront
for ipw in 1 2; do echo "== ZAB_IPW=$ipw"; ZAB_IPW=$ipw python tools/catalog_sweep.py --instances 256 --frames 16384 --only DOT,PsychoConvolver,TSEQ,SpectralStabilizer,Texture 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        r = json.loads(l); print(r['leaf'], round(r.get('kernel_ms', 0), 2), 'ms  ipw', r.get('instances_per_wave'))
"; done > gpurun_out/ipw_small.log 2>&1
