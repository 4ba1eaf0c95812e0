#!/bin/bash
# distributed-state DNA walk kernel: parity tests, then A/B against the replicated-state kernel (FMX_VARIANT=22)
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_text_order.py tests/test_gpu_fuzz.py tests/test_naive_fixtures.py tests/test_gpu_save_load.py -x -q -m gpu 2>&1 | tail -5
bash benchmarks/gpu/variant_ab.sh "x 22" --no-rlfm
for nb in 384 512; do
  FMX_LOC_BLOCKS=$nb bash benchmarks/gpu/variant_ab.sh "x" --no-rlfm | sed "s/^/blocks $nb: /"
done
