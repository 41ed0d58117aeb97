#!/bin/bash
# round 4, second GPU session: re-check the time-parallel kernels after the load gating / package split / rule changes, block-size
# sensitivity of the new leaves, lanes per instance of the message-bus leaves, ClickBeGoneSG after its shorter serial chain (G sweep).
O=gpurun_out; mkdir -p $O
R=$(pwd)
timeout -k 10 900 python -m pytest tests/test_tpar.py tests/test_faust.py tests/test_file_slots.py tests/test_shim.py tests/test_fft_builtins.py -m gpu -q --maxfail=40 -p no:cacheprovider > $O/s2_tests.log 2>&1; echo "tests rc=$?" | tee $O/s2_summary.txt
tail -8 $O/s2_tests.log
timeout -k 10 600 python tools/catalog_sweep.py --only Alias,Contour,Texture,TextureXY,TSEQ,NeuroCV,DOT,DPT,Roomalizer --cpu-seconds 1 --out $O/s2_sweep.json > $O/s2_sweep.log 2>&1; echo "sweep rc=$?" | tee -a $O/s2_summary.txt
cut -c1-330 $O/s2_sweep.log
timeout -k 10 400 python tools/catalog_sweep.py --only Alias,Contour,Texture,TextureXY --cpu-seconds 0 --block 48000 --out $O/s2_sweep_oneblock.json > $O/s2_sweep_oneblock.log 2>&1; echo "oneblock rc=$?" | tee -a $O/s2_summary.txt
cut -c1-330 $O/s2_sweep_oneblock.log
for ipw in 1 4 16; do
  ZAB_IPW=$ipw timeout -k 10 300 python tools/catalog_sweep.py --only 3DPannerManager,3DPanner --cpu-seconds 0 --out $O/s2_ipw$ipw.json > $O/s2_ipw$ipw.log 2>&1; echo "ipw $ipw rc=$?" | tee -a $O/s2_summary.txt
  cut -c1-260 $O/s2_ipw$ipw.log
done
for n in 1024 4096 8192; do for g in 1 2 4 8 16; do
  echo "cbg N=$n G=$g: $(ZAB_CBG_G=$g timeout -k 10 120 python bench.py --leaf ClickBeGoneSG --instances-total $n --frames 48000 --no-cpu-baseline --no-null-test --steps 20 --warmup 3 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["roofline"]["kernel"], round(d["roofline"]["kernel_ms"],3), "ms", round(d["roofline"]["frac"],4))')" | tee -a $O/s2_cbg_g_sweep.txt
done; done
