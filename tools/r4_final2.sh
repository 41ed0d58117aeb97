#!/bin/bash
# Round 4, second half: end-of-round evidence on the final binaries (tools/r4_final2.sh 1|2).
#   1: whole GPU suite, smoke(), then part A of tools/final_round_r04.sh (bench + rocprofv3 stats + PMC passes, the other configs' lines)
#   2: parts B and C (sweep with CPU columns, mixed run, FFT harness and its PMC traffic, SQ passes, per-GPU batches)
O=gpurun_out; mkdir -p $O
case ${1:?1|2} in
1)
  timeout -k 10 900 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/r4h_gpu_suite.log 2>&1; echo "suite rc=$?"; tail -2 $O/r4h_gpu_suite.log
  timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
  bash tools/final_round_r04.sh A r4h || echo "part A failed"
  ;;
2)
  bash tools/final_round_r04.sh B r4h || echo "part B failed"
  timeout -k 10 400 bash tools/fft_traffic.sh r4h > $O/r4h_fft_traffic.log 2>&1; echo "fft traffic rc=$?"
  bash tools/r4_extra_stats.sh > $O/r4h_extra_stats.log 2>&1; echo "extra stats rc=$?"
  bash tools/final_round_r04.sh C r4h || echo "part C failed"
  ;;
esac
