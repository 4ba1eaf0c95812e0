#!/bin/bash
# wide RLFM: fuzz (forced-wide RLFM in the mix), the wide test files, the beyond-4G file
O=gpurun_out/r04_wrl; mkdir -p $O
for seed in 51 52 53; do python tests/fuzz_gpu_vs_oracle.py 60 $seed 2>&1 | tail -3; done > $O/fuzz.txt
cat $O/fuzz.txt
python -m pytest tests/test_gpu_wide.py tests/test_gpu_beyond_4g.py tests/test_gpu_rlfm.py tests/test_gpu_save_load.py -x -q 2>&1 | tail -5 > $O/pytest_wide.txt
cat $O/pytest_wide.txt
