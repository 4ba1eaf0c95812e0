#!/bin/bash
# small builds after the round-trip trimming: the construction rows + the tests that build many small indexes
O=gpurun_out/r04_small2; mkdir -p $O
timeout 300 python benchmarks/gpu/construction_rows.py > $O/construction_rows.jsonl 2>/dev/null; cat $O/construction_rows.jsonl
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_rlfm.py tests/test_multi_pieces.py tests/test_gpu_fuzz.py tests/test_gpu_concurrency.py tests/test_gpu_leaks.py -x -q 2>&1 | tail -4 > $O/pytest.txt; cat $O/pytest.txt
