#!/bin/bash
# Round profile: the default bench line, its rocprofv3 kernel-trace summary, and the two PMC passes (FETCH_SIZE, WRITE_SIZE
# -- separate runs, no trace domains mixed in) that give the HBM traffic of the dominant kernel. Run on the GPU box:
#     bash tools/profile_round.sh r01          -> gpurun_out/r01_*   (copy what is to be judged into profiles/)
set -e -o pipefail
TAG=${1:-rXX}
R=$(pwd)
OUT=$R/gpurun_out
mkdir -p $OUT
python3 bench.py > $OUT/${TAG}_bench.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_trace -- python3 $R/bench.py --no-cpu-baseline > $OUT/${TAG}_bench_under_rocprof.json
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_pmc_fetch -- python3 $R/bench.py --no-cpu-baseline --steps 2 --warmup 1 > /dev/null
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_pmc_write -- python3 $R/bench.py --no-cpu-baseline --steps 2 --warmup 1 > /dev/null
cd $R
python3 tools/pmc_summarise.py $OUT ${TAG}
