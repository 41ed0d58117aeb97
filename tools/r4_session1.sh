#!/bin/bash
# round 4, first GPU session: the new time-parallel kernels (statements as events, feedback cuts, shared delay lines) against the VM
# fixtures and the generic kernel; then the whole GPU suite; then the sweep rows of the leaves that moved and a kernel trace of the
# message-bus leaves.
O=gpurun_out; mkdir -p $O
R=$(pwd)
timeout -k 10 900 python -m pytest tests/test_tpar.py -m gpu -q --maxfail=40 -p no:cacheprovider > $O/s1_tpar.log 2>&1; echo "tpar rc=$?" | tee -a $O/s1_summary.txt
tail -5 $O/s1_tpar.log
timeout -k 10 1500 python -m pytest tests -m gpu -q --maxfail=60 -p no:cacheprovider --deselect tests/test_tpar.py > $O/s1_all.log 2>&1; echo "all rc=$?" | tee -a $O/s1_summary.txt
tail -5 $O/s1_all.log
timeout -k 10 900 python tools/catalog_sweep.py --only Alias,Contour,Texture,TextureXY,IPCProbeA,3DPanner,3DPannerManager,Sample --cpu-seconds 1 --out $O/s1_sweep.json > $O/s1_sweep.log 2>&1; echo "sweep rc=$?" | tee -a $O/s1_summary.txt
cat $O/s1_sweep.log | cut -c1-400
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/s1_trace_ipc -- python3 $R/tools/catalog_sweep.py --only IPCProbeA,3DPanner,3DPannerManager --cpu-seconds 0 > $R/$O/s1_trace_ipc.log 2>&1; echo "trace rc=$?" | tee -a $R/$O/s1_summary.txt
cd $R
find $O/s1_trace_ipc -name "*kernel_stats.csv" | head -1 | xargs -r head -12
