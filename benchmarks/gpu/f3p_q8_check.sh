#!/bin/bash
# DNA walk kernel: 8 walks per group (one per lane, FMX_VARIANT=15) against 4 (shipped); parity first
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_text_order.py tests/test_gpu_fuzz.py tests/test_naive_fixtures.py -x -q -m gpu 2>&1 | tail -3
bash benchmarks/gpu/variant_ab.sh "x 15" --no-rlfm
FMX_LOC_BLOCKS=256 bash benchmarks/gpu/variant_ab.sh "15" --no-rlfm | sed "s/^/blocks 256: /"
FMX_LOC_BLOCKS=512 FMX_LOC_THREADS=512 bash benchmarks/gpu/variant_ab.sh "15" --no-rlfm | sed "s/^/512x512: /"
