#!/bin/bash
O=gpurun_out/r04_rlfm; mkdir -p $O
python -m pytest tests/test_gpu_rlfm.py tests/test_gpu_text_order.py tests/test_gpu_save_load.py tests/test_naive_fixtures.py tests/test_gpu_fuzz.py -x -q -m gpu > $O/pytest.txt 2>&1; tail -5 $O/pytest.txt
python bench.py --steps 20 --warmup 5 --workload bytes-rlfm --no-pmc --no-accel --no-d2h --no-rccl-check --no-wide > $O/bench_rlfm.json 2> $O/bench_rlfm.err; tail -c 300 $O/bench_rlfm.err
python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r04_rlfm/bench_rlfm.json") if l.startswith("{")][-1])
l=d["locate"]
print("rlfm count", d["value"], d["ms_per_step"], d["config"]["index_bytes"], d["config"]["build_ms"])
print("rlfm locate", l["hits_per_s"], l["ms_per_batch"], l.get("walk_kernel_ms"), l["lf_steps"], l["roofline"].get("requested_lines"), l["roofline"].get("requested_records"), l["roofline"].get("requested_probes"))
PY
