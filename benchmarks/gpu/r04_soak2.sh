#!/bin/bash
# after the wide RLFM / multi-pieces work: the range-checked library on the wide / multi / RLFM tests, and the
# differential soak (forced-wide indexes of every kind in the mix) on the shipped and the range-checked library
O=gpurun_out/r04_soak2; mkdir -p $O
FMX_LIB=$PWD/fm_index_amd/libfmx_debug.so python -m pytest tests/test_gpu_wide.py tests/test_multi_pieces.py tests/test_gpu_rlfm.py tests/test_gpu_wide_symbols.py tests/test_gpu_forward.py -q -m gpu > $O/pytest_debuglib.txt 2>&1; tail -3 $O/pytest_debuglib.txt
for seed in 71 72 73 74; do python tests/fuzz_gpu_vs_oracle.py 90 $seed 2>&1 | tail -1; done > $O/soak.txt
for seed in 75 76; do FMX_LIB=$PWD/fm_index_amd/libfmx_debug.so python tests/fuzz_gpu_vs_oracle.py 90 $seed 2>&1 | tail -1; done >> $O/soak.txt
cat $O/soak.txt
