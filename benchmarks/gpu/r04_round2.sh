#!/bin/bash
# the tests touched since the last full suite + the default bench line; output under gpurun_out/r04_b/
O=gpurun_out/r04_b; mkdir -p $O
( time python -m pytest tests/test_gpu_walk_records.py tests/test_gpu_pair_index.py tests/test_gpu_fullsize.py tests/test_gpu_fuzz.py tests/test_gpu_bench_multirank.py tests/test_gpu_leaks.py tests/test_gpu_concurrency.py -x -q -m gpu ) > $O/pytest.txt 2>&1; tail -6 $O/pytest.txt
( time python bench.py --steps 20 --warmup 5 ) > $O/bench_default.json 2> $O/bench_default.err; tail -c 300 $O/bench_default.err
sed -i 's#gpurun_out/r04_suite/bench_default.json#gpurun_out/r04_b/bench_default.json#' benchmarks/gpu/suite_and_bench_r04.sh
python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r04_b/bench_default.json") if l.startswith("{")][-1])
def g(o,*k):
    for x in k:
        o=(o or {}).get(x)
    return o
print("value",d["value"],"frac",g(d,"roofline","frac"), "value_auto", d.get("value_auto"))
for leg in ("locate","locate_row_order","locate_3b"):
    print(leg, g(d,leg,"hits_per_s"), g(d,leg,"ms_per_batch"), g(d,leg,"walk_kernel_ms"), g(d,leg,"sampling"), "frac", g(d,leg,"roofline","frac"), g(d,leg,"roofline","frac_of_gather_ceiling"), "two", g(d,leg,"two_streams","hits_per_s"), g(d,leg,"error"))
print("rlfm", g(d,"rlfm","value"), "loc", g(d,"rlfm","locate","ms_per_batch"), g(d,"rlfm","locate","walk_kernel_ms"), g(d,"rlfm","locate","roofline","frac"))
print("d2h", d.get("value_incl_d2h"), "c5", g(d,"config5_g1","value"), g(d,"config5_g1","matches_golden"), g(d,"config5_g1","error"))
print("wide", g(d,"wide","value"), g(d,"wide","locate","hits_per_s"), g(d,"wide","build_ms"), g(d,"wide","error"))
for leg in ("pair_index","kmer_table","kmer_table+pair_index"):
    print(leg, g(d,leg,"value"), "frac", g(d,leg,"roofline","frac"), g(d,leg,"roofline","traffic"), g(d,leg,"roofline","traffic_kernel"), g(d,leg,"error"), g(d,leg,"skipped"))
print("cpu", g(d,"cpu_baseline","value"), g(d,"cpu_baseline","cores"))
PY
