#!/bin/bash
O=gpurun_out/r04_rlfm; mkdir -p $O
python -m pytest tests/test_gpu_rlfm.py tests/test_gpu_text_order.py tests/test_naive_fixtures.py tests/test_gpu_parity.py tests/test_gpu_save_load.py tests/test_gpu_forward.py -x -q -m gpu > $O/pytest2.txt 2>&1; tail -4 $O/pytest2.txt
python benchmarks/criterion_shapes.py > $O/criterion_shapes.jsonl 2>/dev/null; grep -i "locate" $O/criterion_shapes.jsonl | cut -c1-220
SAMPLING= python benchmarks/gpu/small_shapes.py rlfm 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['index'], d['log2n'], d['log2npat'], d['hits'], 'count_us', d['count_us'], 'locate_us', d['locate_us'])"
