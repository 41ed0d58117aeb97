#!/bin/bash
# after the ZaEnv / ZaState split: per-sample arena access cost, the FFT leaves, and the generic part of the catalog
set -e
python tools/ring_io.py 1024 > gpurun_out/env_ring_io.log 2>&1
python tools/fft_bench.py > gpurun_out/env_fft_bench.log 2>&1
python -m pytest tests/test_catalog_gpu.py tests/test_fft_builtins.py tests/test_msg_bus.py tests/test_gmem.py tests/test_file_slots.py tests/test_pool.py -m gpu -q -x > gpurun_out/env_tests.log 2>&1
