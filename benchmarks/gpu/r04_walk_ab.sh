#!/bin/bash
# round 4, VERDICT item 2: the walk-record kernel (text order, no phase probes) against the row-order walk and the
# round-3 text-order walk on config 3 / config 3b.  Output under gpurun_out/r04_walk/.
set -x
O=gpurun_out/r04_walk; mkdir -p $O
python -m pytest tests/test_gpu_walk_records.py tests/test_gpu_text_order.py tests/test_gpu_save_load.py -x -q -m gpu > $O/pytest.txt 2>&1; tail -5 $O/pytest.txt
python benchmarks/gpu/walk_ab.py > $O/walk_ab.jsonl 2> $O/walk_ab.err; cat $O/walk_ab.jsonl; tail -3 $O/walk_ab.err
