#!/bin/bash
# differential soak on the final sources (shipped + range-checked library) and config 4b with the run table
O=gpurun_out/r04_soak; mkdir -p $O
for seed in 41 42 43; do timeout 200 python tests/fuzz_gpu_vs_oracle.py 90 $seed 2>&1 | tail -1; done > $O/soak.txt
FMX_LIB=$PWD/fm_index_amd/libfmx_debug.so timeout 200 python tests/fuzz_gpu_vs_oracle.py 90 44 2>&1 | tail -1 >> $O/soak.txt   # needs `make debug`
python bench.py --workload rep-rlfm --steps 10 --warmup 2 --no-pmc --no-accel --no-d2h --no-rccl-check --no-wide --no-cpu-baseline --no-census > $O/bench_config4b.json 2> $O/bench_config4b.err
python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r04_soak/bench_config4b.json") if l.startswith("{")][-1])
l=d["locate"]
print("4b count", d["value"], d["ms_per_step"], d["config"]["index_bytes"], d["config"]["build_ms"])
print("4b locate", l["hits"], l["hits_per_s"], l["ms_per_batch"], l.get("walk_kernel_ms"))
PY
