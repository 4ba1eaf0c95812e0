#!/bin/bash
# the lane-per-walk DNA locate for long intervals: parity tests, config 3 / 3b
O=gpurun_out/r04_walk_lane; mkdir -p $O
timeout 400 python -m pytest tests/test_gpu_walk_records.py tests/test_gpu_text_order.py tests/test_gpu_large_batches.py -x -q 2>&1 | tail -3 > $O/pytest.txt; cat $O/pytest.txt
timeout 200 python bench.py --no-pmc --no-accel --no-wide --no-rlfm --no-cpu-baseline --no-d2h --no-rccl-check --no-census --no-early-exit --no-pretouch > $O/bench.json 2> $O/bench.err
python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r04_walk_lane/bench.json") if l.startswith("{")][-1])
l=d["locate"]; b=d["locate_3b"]
print("locate", l["hits_per_s"], l["ms_per_batch"], l.get("matches_golden"))
print("3b", b["hits"], b["hits_per_s"], b["ms_per_batch"], b.get("walk_kernel_ms"), b.get("lf_steps"))
PY
