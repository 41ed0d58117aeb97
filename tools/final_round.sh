#!/bin/bash
# End-of-round evidence in one gpurun call (tag = rNN): bench + rocprofv3 stats + PMC passes (profile_round.sh), the one-process
# multi-GPU driver at N = 1, SQ passes (headline kernel; an @block leaf on its generic and on its time-parallel kernel), the
# catalog sweep with CPU columns, the mixed-leaf run, the FFT harness, the other configs' bench lines. Steps are joined so that a
# failing GPU step ends the call. tools/collect_round.sh <tag> <round> copies what is to be judged into profiles/.
TAG=${1:-rXX}
O=gpurun_out
bash tools/profile_round.sh $TAG > $O/${TAG}_profile_round.log 2>&1 || exit 1
python bench.py --group --no-cpu-baseline > $O/${TAG}_bench_group.json 2> $O/${TAG}_bench_group.err || exit 1
ZAB_DDT_NW=2 tools/sq_pass.sh ${TAG}_sq_ddt DDT 4096 480000 auto > $O/${TAG}_sq_ddt.log 2>&1 || exit 1
tools/sq_pass.sh ${TAG}_sq_erb_tpar ERBTilt 1024 48000 fast > $O/${TAG}_sq_erb_tpar.log 2>&1 || exit 1
tools/sq_pass.sh ${TAG}_sq_erb_generic ERBTilt 1024 48000 generic > $O/${TAG}_sq_erb_generic.log 2>&1 || exit 1
python tools/catalog_sweep.py --cpu-seconds 2 --out $O/${TAG}_catalog_sweep.json > $O/${TAG}_catalog_sweep.log 2>&1 || exit 1
python tools/catalog_mixed.py --out $O/${TAG}_catalog_mixed.json > $O/${TAG}_catalog_mixed.log 2>&1 || exit 1
python tools/fft_bench.py > $O/${TAG}_fft_bench.log 2>&1 || exit 1
python bench.py --instances-total 1024 --no-cpu-baseline > $O/${TAG}_bench_1024.json 2> $O/${TAG}_bench_1024.err || exit 1
bash tools/extra_profiles.sh $TAG
