#!/bin/bash
# differential soak of the shipped and the bounds-checked library against the CPU oracle
mkdir -p gpurun_out/soak
for seed in ${SOAK_SEEDS:-31 32}; do timeout 400 python tests/fuzz_gpu_vs_oracle.py 150 $seed 2>&1 | tail -2; done
FMX_LIB=$PWD/fm_index_amd/libfmx_debug.so timeout 400 python tests/fuzz_gpu_vs_oracle.py 150 33 2>&1 | tail -2
