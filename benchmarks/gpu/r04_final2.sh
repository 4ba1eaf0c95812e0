#!/bin/bash
# after the small-build trimming: the whole suite, a soak on both libraries
O=gpurun_out/r04_final2; mkdir -p $O
( time timeout 1500 python -m pytest tests -q -m gpu ) > $O/pytest_gpu.txt 2>&1; tail -6 $O/pytest_gpu.txt
for seed in 111 112; do timeout 200 python tests/fuzz_gpu_vs_oracle.py 80 $seed 2>&1 | tail -1; done > $O/soak.txt
for seed in 113 114; do FMX_LIB=$PWD/fm_index_amd/libfmx_debug.so timeout 200 python tests/fuzz_gpu_vs_oracle.py 80 $seed 2>&1 | tail -1; done >> $O/soak.txt
FMX_LIB=$PWD/fm_index_amd/libfmx_debug.so timeout 600 python -m pytest tests/test_gpu_rlfm.py tests/test_gpu_parity.py tests/test_gpu_forward.py -q -m gpu 2>&1 | tail -2 >> $O/soak.txt
cat $O/soak.txt
