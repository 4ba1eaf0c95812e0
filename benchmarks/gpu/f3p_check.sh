#!/bin/bash
# hand-over at the top of the round (fmx_locate_f3p_kernel, shipped) against hand-over at the end
# (fmx_locate_f3q_kernel<Q,false>, FMX_VARIANT=23): parity tests, then A/B on one box
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_text_order.py tests/test_gpu_fuzz.py tests/test_naive_fixtures.py tests/test_gpu_save_load.py tests/test_gpu_fullsize.py -x -q -m gpu 2>&1 | tail -3
FMX_LIB=$PWD/fm_index_amd/libfmx_debug.so timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_naive_fixtures.py -x -q -m gpu 2>&1 | tail -2
bash benchmarks/gpu/variant_ab.sh "x 23" --no-rlfm
