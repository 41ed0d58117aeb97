#!/bin/bash
# copy what tools/final_round.sh left under gpurun_out/<tag>_* into profiles/<round>_*: tools/collect_round.sh r3f r03
T=${1:?tag}; R=${2:?round, e.g. r03}; O=gpurun_out; P=profiles
mkdir -p $P/${R}_sq
cp $O/${T}_bench.json $P/${R}_bench.json; cp $O/${T}_bench_kernel_stats.csv $P/${R}_bench_kernel_stats.csv; cp $O/${T}_bench_under_rocprof.json $P/${R}_bench_under_rocprof.json
cp $O/${T}_bench_group.json $P/${R}_bench_group.json
cp $O/${T}_bench_1024.json $P/${R}_bench_1024.json; cp $O/${T}_pmc_traffic.json $P/${R}_pmc_traffic.json; cp $O/${T}_pmc_traffic.json $P/pmc_traffic.json
cp $O/${T}_catalog_sweep.json $P/${R}_catalog_sweep.json; cp $O/${T}_catalog_mixed.json $P/${R}_catalog_mixed.json
cp $O/${T}_fft_bench.log $P/${R}_fft_bench.txt; cp $O/${T}_fft_kernel_stats.csv $P/${R}_fftbench_x2048_4096pt_kernel_stats.csv
cp $O/${T}_bench_stft.json $P/${R}_bench_stft_x1024.json; cp $O/${T}_bench_cbg.json $P/${R}_bench_clickbegone_x1024.json
cp $O/${T}_sq_ddt/summary.txt $P/${R}_sq/ddt_fast_nw2_x4096.txt; cp $O/${T}_sq_erb_tpar/summary.txt $P/${R}_sq/erbtilt_tpar.txt; cp $O/${T}_sq_erb_generic/summary.txt $P/${R}_sq/erbtilt_generic.txt
