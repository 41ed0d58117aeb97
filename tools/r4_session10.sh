#!/bin/bash
# round 4, tenth GPU session: the meter window on ONE box -- 45 056 frames (lib) against 57 344 (build_ab/libzab_DDT_b.so) --
# then the whole GPU suite.
O=gpurun_out; mkdir -p $O
L=zorakaudio-experimental-plugins_amd/lib
for rep in 1 2; do python tools/quick_bench.py 4096 480000 a_45056_$rep; done > $O/s10_ab.txt 2>&1
cp $L/libzab_DDT.so /tmp/a.so && cp build_ab/libzab_DDT_b.so $L/libzab_DDT.so
for rep in 1 2; do python tools/quick_bench.py 4096 480000 b_57344_$rep; done >> $O/s10_ab.txt 2>&1
cp /tmp/a.so $L/libzab_DDT.so
python tools/quick_bench.py 4096 480000 a_45056_3 >> $O/s10_ab.txt 2>&1
python tools/quick_bench.py 512 480000 a_shard512 >> $O/s10_ab.txt 2>&1
python tools/quick_bench.py 1024 480000 a_1024 >> $O/s10_ab.txt 2>&1
cat $O/s10_ab.txt
timeout -k 10 1100 python -m pytest tests -m gpu -q --maxfail=30 -p no:cacheprovider > $O/s10_gpu_tests.log 2>&1; echo "gpu tests rc=$?" | tee $O/s10_summary.txt
tail -8 $O/s10_gpu_tests.log
