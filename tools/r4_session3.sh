#!/bin/bash
# round 4, third GPU session: TextureXY / Texture after the cell-alias fix, the four-wavefront ClickBeGoneSG kernel, phase clocks of
# the new leaves, the whole GPU suite.
O=gpurun_out; mkdir -p $O
R=$(pwd)
timeout -k 10 900 python -m pytest tests/test_tpar.py tests/test_faust.py -m gpu -q --maxfail=40 -p no:cacheprovider > $O/s3_tests.log 2>&1; echo "tests rc=$?" | tee $O/s3_summary.txt
tail -8 $O/s3_tests.log
for n in 1024 1022 4096 8192; do for k in wave quad; do
  echo "cbg N=$n $k: $(ZAB_CBG_KERNEL=$k timeout -k 10 120 python bench.py --leaf ClickBeGoneSG --instances-total $n --frames 48000 --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["roofline"]["kernel"], round(d["roofline"]["kernel_ms"],3), "ms", round(d["roofline"]["frac"],4), "null", d["null_test_dbfs"])')" | tee -a $O/s3_cbg_quad.txt
done; done
timeout -k 10 600 python tools/catalog_sweep.py --only Texture,TextureXY,3DPanner,Contour --cpu-seconds 1 --out $O/s3_sweep.json > $O/s3_sweep.log 2>&1; echo "sweep rc=$?" | tee -a $O/s3_summary.txt
cut -c1-330 $O/s3_sweep.log
for l in Texture Contour TextureXY Alias; do
  cap=0; [ $l = Texture ] && cap=33554432; [ $l = TextureXY ] && cap=33554432; [ $l = Contour ] && cap=16777216; [ $l = Alias ] && cap=524288
  n=192; [ $l = Alias ] && n=1024; [ $l = Contour ] && n=384
  timeout -k 10 300 python tools/tpar_stamps.py ${l}_stamps $n 48000 $cap >> $O/s3_stamps.txt 2>&1
done
cat $O/s3_stamps.txt
timeout -k 10 1500 python -m pytest tests -m gpu -q --maxfail=60 -p no:cacheprovider --deselect tests/test_tpar.py --deselect tests/test_faust.py > $O/s3_all.log 2>&1; echo "all rc=$?" | tee -a $O/s3_summary.txt
tail -6 $O/s3_all.log
