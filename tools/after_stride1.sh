#!/bin/bash
# replica-lane leaves compiled for contiguous arenas (ZA_MEM_STRIDE1): FFT harness, catalog rows, parity
set -e
python tools/fft_bench.py > gpurun_out/s1_fft_bench.log 2>&1
python tools/catalog_sweep.py --only TSEQ,SpectralStabilizer,Texture,Contour,DOT,PsychoConvolver,Sample > gpurun_out/s1_sweep.log 2>&1
python -m pytest tests -m gpu -q -x > gpurun_out/s1_tests.log 2>&1
