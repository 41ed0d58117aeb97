#!/bin/bash
# copy what tools/final_round.sh + tools/extra_profiles.sh left under gpurun_out/<tag>_* into profiles/r02_*: tools/collect_round.sh r02d
T=${1:?tag}; O=gpurun_out; P=profiles
cp $O/${T}_bench.json $P/r02_bench.json; cp $O/${T}_bench_kernel_stats.csv $P/r02_bench_kernel_stats.csv; cp $O/${T}_bench_under_rocprof.json $P/r02_bench_under_rocprof.json
cp $O/${T}_bench_1024.json $P/r02_bench_1024.json; cp $O/${T}_pmc_traffic.json $P/r02_pmc_traffic.json; cp $O/${T}_pmc_traffic.json $P/pmc_traffic.json
cp $O/${T}_catalog_sweep.json $P/r02_catalog_sweep.json; cp $O/${T}_fft_bench.log $P/r02_fft_bench.txt; cp $O/${T}_fft_kernel_stats.csv $P/r02_fftbench_x2048_4096pt_kernel_stats.csv
cp $O/${T}_bench_stft.json $P/r02_bench_stft_x1024.json; cp $O/${T}_bench_cbg.json $P/r02_bench_clickbegone_x1024.json
cp $O/${T}_sq_ddt/summary.txt $P/r02_sq/ddt_fast_nw2_x4096.txt; cp $O/${T}_sq_sp_tpar/summary.txt $P/r02_sq/sp_tpar.txt; cp $O/${T}_sq_sp_generic/summary.txt $P/r02_sq/sp_generic.txt
