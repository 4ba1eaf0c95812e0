#!/bin/bash
# the whole -m gpu suite + the default bench line; output under gpurun_out/r04_suite/
O=gpurun_out/r04_suite; mkdir -p $O
( time python -m pytest tests -x -q -m gpu ) > $O/pytest_gpu.txt 2>&1; tail -6 $O/pytest_gpu.txt
( time python bench.py --steps 20 --warmup 5 ) > $O/bench_default.json 2> $O/bench_default.err; tail -c 400 $O/bench_default.err
python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r04_suite/bench_default.json") if l.startswith("{")][-1])
def g(o,*k):
    for x in k:
        o=(o or {}).get(x)
    return o
print("value",d["value"],"frac",g(d,"roofline","frac"))
for leg in ("locate","locate_row_order","locate_3b"):
    print(leg, g(d,leg,"hits_per_s"), g(d,leg,"ms_per_batch"), g(d,leg,"sampling"), "frac", g(d,leg,"roofline","frac"), "two", g(d,leg,"two_streams","hits_per_s"), g(d,leg,"error"))
print("rlfm", g(d,"rlfm","value"), "loc", g(d,"rlfm","locate","ms_per_batch"), g(d,"rlfm","locate","roofline","frac"))
print("d2h", d.get("value_incl_d2h"), "c5", g(d,"config5_g1","value"), g(d,"config5_g1","matches_golden"), g(d,"config5_g1","error"))
print("wide", g(d,"wide","value"), g(d,"wide","locate","hits_per_s"), g(d,"wide","build_ms"), g(d,"wide","error"))
print("accel", g(d,"pair_index","value"), g(d,"kmer_table","value"), g(d,"kmer_table+pair_index","value"))
print("cpu", g(d,"cpu_baseline","value"), g(d,"cpu_baseline","cores"))
PY
