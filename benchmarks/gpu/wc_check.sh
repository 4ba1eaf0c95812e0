#!/bin/bash
# write-combining ring for the positions of the DNA walk kernel (shipped) against direct stores
# (FMX_VARIANT=26, measurement build): parity first (shipped and bounds-checked library), then A/B
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_large_batches.py tests/test_gpu_text_order.py tests/test_gpu_fuzz.py tests/test_naive_fixtures.py tests/test_gpu_fullsize.py -x -q -m gpu 2>&1 | tail -3
FMX_LIB=$PWD/fm_index_amd/libfmx_debug.so timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_large_batches.py -x -q -m gpu 2>&1 | tail -2
bash benchmarks/gpu/variant_ab.sh "x 26" --no-rlfm
