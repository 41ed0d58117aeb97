#!/bin/bash
# last GPU call of round 4: the whole GPU suite and smoke() on the final binaries, then the lines that changed since parts A-C
# (sweep, mixed run, bench at 1024 instances with its PMC file in place).
O=gpurun_out; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/r4f_gpu_suite.log 2>&1; echo "suite rc=$?"; tail -2 $O/r4f_gpu_suite.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
python bench.py > $O/r4f_bench.json 2> $O/r4f_bench.err; echo "bench rc=$?"; cut -c1-220 $O/r4f_bench.json
python bench.py --instances-total 1024 --no-cpu-baseline > $O/r4f_bench_1024.json 2>/dev/null; echo "bench1024 rc=$?"
python tools/catalog_sweep.py --cpu-seconds 2 --out $O/r4f_catalog_sweep.json > $O/r4f_catalog_sweep.log 2>&1; echo "sweep rc=$?"
python tools/catalog_mixed.py --out $O/r4f_catalog_mixed.json > $O/r4f_catalog_mixed.log 2>&1; echo "mixed rc=$?"; grep wall_s $O/r4f_catalog_mixed.log | cut -c1-200
