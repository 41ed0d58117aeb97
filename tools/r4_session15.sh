#!/bin/bash
# DDT headline on one box: module text without named constants (lib) against with (build_ab); 10 timed steps each
O=gpurun_out; mkdir -p $O
L=zorakaudio-experimental-plugins_amd/lib
qb() { python bench.py --steps 10 --warmup 2 --no-cpu-baseline --frames 480000 --instances-per-gpu $1 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']; print('$2', '$1', r['kernel'], round(r['kernel_ms'],3), 'min', round(r['kernel_ms_min'],3), 'copy', round(r['device_copy_gbs']))"; }
qb 4096 noconst > $O/s15_ab.txt; qb 4096 noconst >> $O/s15_ab.txt
cp $L/libzab_DDT.so /tmp/a.so && cp build_ab/libzab_DDT_b.so $L/libzab_DDT.so
qb 4096 consts >> $O/s15_ab.txt; qb 4096 consts >> $O/s15_ab.txt
cp /tmp/a.so $L/libzab_DDT.so
qb 4096 noconst >> $O/s15_ab.txt
ZAB_DDT_MINW=2 qb 4096 noconst_minw2 >> $O/s15_ab.txt
rocm-smi --showclocks 2>/dev/null | grep -i "sclk\|mclk" | head -4 >> $O/s15_ab.txt
cat $O/s15_ab.txt
