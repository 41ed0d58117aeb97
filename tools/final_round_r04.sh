#!/bin/bash
# End-of-round evidence of round 4, in gpurun calls of <= 20 minutes each (tools/final_round_r04.sh <part> <tag>):
#   part A: bench + rocprofv3 stats + PMC passes of the headline (tools/profile_round.sh), the one-process multi-GPU driver at N = 1,
#           the bench lines of the other configs, per-kernel PMC passes (tools/pmc_pass.sh): ClickBeGoneSG, the STFT fixture, one
#           generated time-parallel kernel (Alias), one generic kernel (Sample)
#   part B: the catalog sweep with CPU columns, the mixed-leaf run, the FFT harness
#   part C: SQ passes (headline; a round-4 leaf on both of its kernels; the two ClickBeGoneSG kernels), per-GPU batches of the headline
PART=${1:?A|B|C}; TAG=${2:-r4f}
O=gpurun_out; mkdir -p $O
case $PART in
A)
  bash tools/profile_round.sh $TAG > $O/${TAG}_profile_round.log 2>&1 || exit 1
  python bench.py --group --no-cpu-baseline > $O/${TAG}_bench_group.json 2> $O/${TAG}_bench_group.err || exit 1
  python bench.py --instances-total 1024 --no-cpu-baseline > $O/${TAG}_bench_1024.json 2> $O/${TAG}_bench_1024.err || exit 1
  python bench.py --leaf fx_stft --instances-total 1024 --frames 16384 > $O/${TAG}_bench_stft.json 2> $O/${TAG}_bench_stft.err || exit 1
  python bench.py --leaf ClickBeGoneSG --instances-total 1024 --frames 48000 > $O/${TAG}_bench_cbg.json 2> $O/${TAG}_bench_cbg.err || exit 1
  python bench.py --leaf ClickBeGoneSG --instances-total 8192 --frames 48000 --no-cpu-baseline > $O/${TAG}_bench_cbg_8192.json 2> $O/${TAG}_bench_cbg_8192.err || exit 1
  bash tools/pmc_pass.sh ${TAG}_cbg --leaf ClickBeGoneSG --instances-total 1024 --frames 48000 > $O/${TAG}_pmc_cbg.log 2>&1 || exit 1
  bash tools/pmc_pass.sh ${TAG}_stft --leaf fx_stft --instances-total 1024 --frames 16384 > $O/${TAG}_pmc_stft.log 2>&1 || exit 1
  bash tools/pmc_pass.sh ${TAG}_alias --leaf Alias --instances-total 1024 --frames 48000 --no-null-test --mem-cap 524288 > $O/${TAG}_pmc_alias.log 2>&1 || exit 1
  bash tools/pmc_pass.sh ${TAG}_sample --leaf Sample --instances-total 256 --frames 12000 --no-null-test --mem-cap 524288 > $O/${TAG}_pmc_sample.log 2>&1 || exit 1
  ;;
B)
  python tools/catalog_sweep.py --cpu-seconds 2 --out $O/${TAG}_catalog_sweep.json > $O/${TAG}_catalog_sweep.log 2>&1 || exit 1
  python tools/catalog_mixed.py --out $O/${TAG}_catalog_mixed.json > $O/${TAG}_catalog_mixed.log 2>&1 || exit 1
  python tools/fft_bench.py > $O/${TAG}_fft_bench.log 2>&1 || exit 1
  ;;
C)
  tools/sq_pass.sh ${TAG}_sq_ddt DDT 4096 480000 auto 0 ddt > $O/${TAG}_sq_ddt.log 2>&1 || exit 1
  tools/sq_pass.sh ${TAG}_sq_alias_tpar Alias 1024 48000 fast 524288 > $O/${TAG}_sq_alias_tpar.log 2>&1 || exit 1
  tools/sq_pass.sh ${TAG}_sq_alias_generic Alias 1024 48000 generic 524288 > $O/${TAG}_sq_alias_generic.log 2>&1 || exit 1
  ZAB_CBG_KERNEL=quad tools/sq_pass.sh ${TAG}_sq_cbg_quad ClickBeGoneSG 1024 48000 fast 0 zf_cbg > $O/${TAG}_sq_cbg_quad.log 2>&1 || exit 1
  ZAB_CBG_KERNEL=wave tools/sq_pass.sh ${TAG}_sq_cbg_wave ClickBeGoneSG 1024 48000 fast 0 zf_cbg > $O/${TAG}_sq_cbg_wave.log 2>&1 || exit 1
  bash tools/ddt_nw_per_gpu_batch.sh > $O/${TAG}_ddt_per_gpu_batch.txt 2>&1 || exit 1
  python -m pytest tests/test_ddt_gpu.py tests/test_group_gpu.py -m gpu -q -x > $O/${TAG}_ddt_group.log 2>&1 || exit 1
  ;;
esac
echo "part $PART done"
