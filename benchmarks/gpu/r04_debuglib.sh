#!/bin/bash
# the range-checked library (every index into the HBM arrays trapped: FMX_CHECK) on the tests of round 4's new kernels
O=gpurun_out/r04_debug; mkdir -p $O
export FMX_LIB=$PWD/fm_index_amd/libfmx_debug.so
python -m pytest tests/test_gpu_walk_records.py tests/test_gpu_wide.py tests/test_gpu_text_order.py tests/test_gpu_large_batches.py tests/test_gpu_parity.py tests/test_gpu_save_load.py tests/test_gpu_concurrency.py tests/test_gpu_leaks.py -q -m gpu > $O/pytest_debuglib.txt 2>&1; tail -5 $O/pytest_debuglib.txt
