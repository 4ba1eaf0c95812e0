#!/bin/bash
# walk-record locate with 8 walks per group (FMX_VARIANT=15 in the measurement build) against the shipped 4
O=gpurun_out/r04_q8; mkdir -p $O
F="--no-pmc --no-accel --no-wide --no-rlfm --no-cpu-baseline --no-d2h --no-rccl-check --no-census --no-early-exit --no-pretouch --steps 50 --warmup 10"
for v in 0 15 0 15; do
  FMX_LIB=$PWD/fm_index_amd/libfmx_measure.so FMX_VARIANT=$v timeout 300 python bench.py $F > $O/bench_v$v.json 2> $O/err.txt
  python - <<PY
import json
d=json.loads([l for l in open("$O/bench_v$v.json") if l.startswith("{")][-1])
l=d["locate"]; b=d["locate_3b"]
print("variant $v: locate ms/batch %.4f walk kernel %.4f hits/s %.3e two %.3e | 3b ms %.3f" % (l["ms_per_batch"], l["walk_kernel_ms"], l["hits_per_s"], (l.get("two_streams") or {}).get("hits_per_s", 0), b["ms_per_batch"]))
PY
done
