#!/bin/bash
# last soak of round 4 on the final sources: 10 seeds on the shipped library, 4 on the range-checked one
O=gpurun_out/r04_soak5; mkdir -p $O
for seed in 201 202 203 204 205 206 207 208 209 210; do timeout 200 python tests/fuzz_gpu_vs_oracle.py 90 $seed 2>&1 | tail -1; done > $O/soak.txt
for seed in 211 212 213 214; do FMX_LIB=$PWD/fm_index_amd/libfmx_debug.so timeout 200 python tests/fuzz_gpu_vs_oracle.py 90 $seed 2>&1 | tail -1; done >> $O/soak.txt
cat $O/soak.txt
