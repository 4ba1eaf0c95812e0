#!/bin/bash
# the default bench line (what the driver runs), summarised
O=gpurun_out/r04_bench; mkdir -p $O
( time python bench.py --steps 20 --warmup 5 ) > $O/bench_default.json 2> $O/bench_default.err; tail -c 300 $O/bench_default.err
python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r04_bench/bench_default.json") if l.startswith("{")][-1])
def g(o,*k):
    for x in k:
        o=(o or {}).get(x)
    return o
print("value",d["value"],"frac",g(d,"roofline","frac"), "value_auto", d.get("value_auto"), "build_ms", g(d,"config","build_ms"))
for leg in ("locate","locate_row_order","locate_3b"):
    print(leg, g(d,leg,"hits_per_s"), g(d,leg,"ms_per_batch"), g(d,leg,"walk_kernel_ms"), "frac", g(d,leg,"roofline","frac"), g(d,leg,"roofline","frac_of_gather_ceiling"), "two", g(d,leg,"two_streams","hits_per_s"), g(d,leg,"matches_golden"), g(d,leg,"build_ms"), g(d,leg,"error"))
print("rlfm", g(d,"rlfm","value"), g(d,"rlfm","config","build_ms"), "loc", g(d,"rlfm","locate","ms_per_batch"), g(d,"rlfm","locate","walk_kernel_ms"), g(d,"rlfm","locate","roofline","frac"), g(d,"rlfm","error"))
print("d2h", d.get("value_incl_d2h"), g(d,"incl_d2h","pinned_ms_per_call"), "c5", g(d,"config5_g1","value"), g(d,"config5_g1","matches_golden"), g(d,"config5_g1","error"))
print("wide", g(d,"wide","value"), g(d,"wide","locate","hits_per_s"), g(d,"wide","build_ms"), g(d,"wide","walk_records"), g(d,"wide","error"))
for leg in ("pair_index","kmer_table","kmer_table+pair_index"):
    print(leg, g(d,leg,"value"), "frac", g(d,leg,"roofline","frac"), g(d,leg,"build_ms"), g(d,leg,"error"), g(d,leg,"skipped"))
print("cpu", g(d,"cpu_baseline","value"), g(d,"cpu_baseline","cores"), "matches", d.get("matches_golden"))
PY
