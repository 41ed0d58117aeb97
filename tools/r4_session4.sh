#!/bin/bash
# round 4, fourth GPU session: coupled-state scans, lowered mode switches / light guards (Texture, TextureXY, Contour + loaded runs),
# phase clocks, the sweep rows with the larger arena budget, SQ counters of the ClickBeGoneSG kernels.
O=gpurun_out; mkdir -p $O
R=$(pwd)
timeout -k 10 900 python -m pytest tests/test_tpar.py -m gpu -q --maxfail=40 -p no:cacheprovider > $O/s4_tpar.log 2>&1; echo "tpar rc=$?" | tee $O/s4_summary.txt
tail -6 $O/s4_tpar.log
timeout -k 10 900 python -m pytest tests/test_catalog_gpu.py tests/test_faust.py -m gpu -q --maxfail=40 -p no:cacheprovider > $O/s4_catalog.log 2>&1; echo "catalog rc=$?" | tee -a $O/s4_summary.txt
tail -4 $O/s4_catalog.log
rm -f $O/s4_stamps.txt
for l in Texture TextureXY Contour; do
  cap=33554432; [ $l = Contour ] && cap=16777216
  n=192; [ $l = Contour ] && n=384
  timeout -k 10 300 python tools/tpar_stamps.py ${l}_stamps $n 48000 $cap >> $O/s4_stamps.txt 2>&1
done
cat $O/s4_stamps.txt
timeout -k 10 900 python tools/catalog_sweep.py --only Texture,TextureXY,Contour,Alias,3DPanner,TSEQ,NeuroCV --cpu-seconds 1 --out $O/s4_sweep.json > $O/s4_sweep.log 2>&1; echo "sweep rc=$?" | tee -a $O/s4_summary.txt
cut -c1-360 $O/s4_sweep.log
ZAB_CBG_KERNEL=quad tools/sq_pass.sh s4_sq_cbg_quad ClickBeGoneSG 1024 48000 fast > $O/s4_sq_cbg_quad.log 2>&1; echo "sq quad rc=$?" | tee -a $O/s4_summary.txt
ZAB_CBG_KERNEL=wave tools/sq_pass.sh s4_sq_cbg_wave ClickBeGoneSG 1024 48000 fast > $O/s4_sq_cbg_wave.log 2>&1; echo "sq wave rc=$?" | tee -a $O/s4_summary.txt
cat $O/s4_sq_cbg_quad/summary.txt $O/s4_sq_cbg_wave/summary.txt | cut -c1-200
