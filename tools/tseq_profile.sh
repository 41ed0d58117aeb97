#!/bin/bash
# rocprofv3 kernel stats of the generic kernel VERDICT round 1 cited (TSEQ x1024 x 12 000 frames)
R=$(pwd); OUT=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/tseq_trace -- python3 $R/tools/leaf_scaling.py TSEQ 1024 --frames 12000 > $OUT/tseq_trace.log 2>&1
cd $R
f=$(ls $OUT/tseq_trace/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp $f $OUT/tseq_kernel_stats.csv
