#!/bin/bash
# End-of-round evidence in one gpurun call: bench + rocprofv3 stats + PMC passes (profile_round.sh), SQ passes, sweep, FFT harness
TAG=${1:-rXX}
bash tools/profile_round.sh $TAG > gpurun_out/${TAG}_profile_round.log 2>&1 || exit 1
ZAB_DDT_NW=2 tools/sq_pass.sh ${TAG}_sq_ddt DDT 4096 480000 auto > gpurun_out/${TAG}_sq_ddt.log 2>&1
tools/sq_pass.sh ${TAG}_sq_sp_tpar SaliencePush 1024 48000 fast > gpurun_out/${TAG}_sq_sp_tpar.log 2>&1
tools/sq_pass.sh ${TAG}_sq_sp_generic SaliencePush 1024 48000 generic > gpurun_out/${TAG}_sq_sp_generic.log 2>&1
python tools/catalog_sweep.py --out gpurun_out/${TAG}_catalog_sweep.json > gpurun_out/${TAG}_catalog_sweep.log 2>&1
python tools/fft_bench.py > gpurun_out/${TAG}_fft_bench.log 2>&1
python bench.py --instances-total 1024 --no-cpu-baseline > gpurun_out/${TAG}_bench_1024.json 2> gpurun_out/${TAG}_bench_1024.err
