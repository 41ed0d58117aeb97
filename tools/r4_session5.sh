#!/bin/bash
# round 4, fifth GPU session: rest test of serial recurrences (TextureXY's idle voices), the file-handle window (TextureXY with a
# texture loaded), the software-pipelined recursion of the four-wavefront ClickBeGoneSG kernel and its role clock.
O=gpurun_out; mkdir -p $O
for k in quad wave; do
  echo "cbg N=1024 $k: $(ZAB_CBG_KERNEL=$k timeout -k 10 120 python bench.py --leaf ClickBeGoneSG --instances-total 1024 --frames 48000 --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["roofline"]["kernel"], round(d["roofline"]["kernel_ms"],3), "ms", round(d["roofline"]["frac"],4), "null", d["null_test_dbfs"])')" | tee -a $O/s5_cbg.txt
done
timeout -k 10 120 python tools/cbg_quad_clocks.py run 1024 48000 > $O/s5_cbg_clocks.txt 2>&1; echo "clocks rc=$?" | tee $O/s5_summary.txt
cat $O/s5_cbg_clocks.txt | cut -c1-200
timeout -k 10 900 python -m pytest tests/test_faust.py tests/test_file_slots.py -m gpu -q --maxfail=40 -p no:cacheprovider > $O/s5_faust.log 2>&1; echo "faust+files rc=$?" | tee -a $O/s5_summary.txt
tail -4 $O/s5_faust.log
timeout -k 10 1000 python -m pytest tests/test_tpar.py -m gpu -q --maxfail=40 -p no:cacheprovider > $O/s5_tpar.log 2>&1; echo "tpar rc=$?" | tee -a $O/s5_summary.txt
tail -6 $O/s5_tpar.log
timeout -k 10 900 python tools/catalog_sweep.py --only Texture,TextureXY,Contour,Alias,3DPanner --cpu-seconds 1 --out $O/s5_sweep.json > $O/s5_sweep.log 2>&1; echo "sweep rc=$?" | tee -a $O/s5_summary.txt
cut -c1-360 $O/s5_sweep.log
