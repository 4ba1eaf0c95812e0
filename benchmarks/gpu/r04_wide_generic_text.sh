#!/bin/bash
# generic wide indexes with text-order samples + four walks per group: parity, fuzz, the beyond-4G byte text
O=gpurun_out/r04_wgt; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_wide.py tests/test_multi_pieces.py tests/test_gpu_save_load.py -x -q 2>&1 | tail -5 > $O/pytest.txt; cat $O/pytest.txt
for seed in 91 92; do timeout 200 python tests/fuzz_gpu_vs_oracle.py 60 $seed 2>&1 | tail -2; done > $O/fuzz.txt; cat $O/fuzz.txt
timeout 600 python tests/test_gpu_beyond_4g.py bytes > $O/beyond_4g_bytes.json 2> $O/beyond_4g_bytes.err; tail -3 $O/beyond_4g_bytes.err; cat $O/beyond_4g_bytes.json
