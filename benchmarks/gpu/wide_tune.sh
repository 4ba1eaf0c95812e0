#!/bin/bash
# the wide engine next to the 32-bit one on the config-2 / config-3 shapes (benchmarks/gpu/wide_tune.py):
# n = 2^30 narrow, n = 2^32 + 2^20 wide.  Output: gpurun_out/wide_tune/now.jsonl
mkdir -p gpurun_out/wide_tune
O=gpurun_out/wide_tune/now.jsonl
: > $O
NARROW=1 python benchmarks/gpu/wide_tune.py 30 >> $O
python benchmarks/gpu/wide_tune.py 32 1048576 >> $O
cat $O
