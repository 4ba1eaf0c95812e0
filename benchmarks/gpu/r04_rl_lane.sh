#!/bin/bash
# the lane-per-walk RLFM locate on the 32-bit engine: parity tests, config 4b
O=gpurun_out/r04_rl_lane; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_rlfm.py -x -q 2>&1 | tail -3 > $O/pytest.txt; cat $O/pytest.txt
timeout 400 python bench.py --workload rep-rlfm --steps 10 --warmup 2 --no-pmc --no-accel --no-d2h --no-rccl-check --no-wide --no-cpu-baseline --no-census > $O/bench_config4b.json 2> $O/bench_config4b.err
python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r04_rl_lane/bench_config4b.json") if l.startswith("{")][-1])
l=d["locate"]
print("4b count", d["value"], d["ms_per_step"])
print("4b locate", l["hits"], l["hits_per_s"], l["ms_per_batch"], l.get("walk_kernel_ms"), l.get("lf_steps"))
PY
timeout 300 python benchmarks/gpu/wide_rlfm.py --log2n 30 --narrow 2>/dev/null | tail -1
