#!/bin/bash
# HBM traffic (FETCH_SIZE / WRITE_SIZE passes) of the dominant kernel of every configuration bench.py is run on this round; the
# pmc_<kernel>_<instances>x<frames>.json files go into profiles/, where bench.py finds them by kernel name and batch shape.
O=gpurun_out; mkdir -p $O
run() { t=$1; shift; bash tools/pmc_pass.sh $t "$@" > $O/${t}.log 2>&1; echo "$t rc=$?"; tail -1 $O/${t}.log | cut -c1-300; }
run p4_ddt
run p4_ddt1024 --instances-total 1024
run p4_cbg --leaf ClickBeGoneSG --instances-total 1024 --frames 48000
run p4_cbg8192 --leaf ClickBeGoneSG --instances-total 8192 --frames 48000
run p4_stft --leaf fx_stft --instances-total 1024 --frames 16384
run p4_alias --leaf Alias --instances-total 1024 --frames 48000 --no-null-test --mem-cap 524288
run p4_sample --leaf Sample --instances-total 256 --frames 12000 --no-null-test --mem-cap 524288
ls $O/pmc_*.json
