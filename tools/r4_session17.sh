#!/bin/bash
# round 4, GPU session 17: six waves per instance at three waves per SIMD (zab_ddt_fast_nw6w3) for two instances per CU.
O=gpurun_out; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_ddt_gpu.py -m gpu -q -p no:cacheprovider > $O/s17_ddt_tests.log 2>&1; echo "ddt tests rc=$?" | tee $O/s17_summary.txt
tail -3 $O/s17_ddt_tests.log
qb() { python bench.py --steps 10 --warmup 2 --no-cpu-baseline --frames 480000 --instances-per-gpu $1 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']; print('$2', '$1', r['kernel'], round(r['kernel_ms'],3), 'min', round(r['kernel_ms_min'],3), 'Gs/s', round(j['value']/1000,1), 'null', j['null_test_dbfs'])"; }
{ qb 512 auto; ZAB_DDT_NW=4 qb 512 nw4; ZAB_DDT_NW=4 ZAB_DDT_MINW=3 qb 512 nw4w3; qb 512 auto; qb 384 auto; ZAB_DDT_NW=4 qb 384 nw4; qb 257 auto; ZAB_DDT_NW=6 qb 1024 nw6; qb 1024 auto; qb 4096 auto; } > $O/s17_nw6.txt 2>&1
cat $O/s17_nw6.txt
