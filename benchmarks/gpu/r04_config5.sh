#!/bin/bash
# round 4, VERDICT item 1: config 5 on one GPU -- the oracle-made golden hash, the multirank tests, the strong-scaling
# G = 1 line and the default line (config5_g1 object).  Output under gpurun_out/r04_config5/.
set -x
O=gpurun_out/r04_config5; mkdir -p $O
[ -n "$SKIP_GOLDEN" ] || { python tests/golden/make_config5_golden.py > $O/golden.log 2>&1; python tests/golden/make_config5_golden.py --seed 3 --total 1048576 >> $O/golden.log 2>&1; python tests/golden/make_config5_golden.py --log2n 16 --total 65536 >> $O/golden.log 2>&1; python tests/golden/make_config5_golden.py --log2n 16 --total 8193 >> $O/golden.log 2>&1; python tests/golden/make_config5_golden.py --log2n 16 --total 8192 >> $O/golden.log 2>&1; cp tests/golden/config5_counts.json $O/; }
python -m pytest tests/test_gpu_bench_multirank.py -x -q -m gpu > $O/pytest_multirank.txt 2>&1; tail -5 $O/pytest_multirank.txt
python bench.py --gpus 1 --total-patterns 8388608 --steps 20 --warmup 5 > $O/bench_strong_g1.json 2> $O/bench_strong_g1.err; tail -c 600 $O/bench_strong_g1.err
python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; tail -c 600 $O/bench_default.err
python - <<'PY'
import json
for f in ("bench_strong_g1","bench_default"):
    try:
        d=json.loads(open("gpurun_out/r04_config5/%s.json"%f).read().strip().splitlines()[-1])
        print(f, d["value"], d["ms_per_step"], d["scaling"], d.get("counts_sha256"), d.get("ranges_sha256"), d.get("matches_golden"),
              (d.get("roofline") or {}).get("frac"), (d.get("roofline") or {}).get("traffic_source"), (d.get("cpu_baseline") or {}).get("value"))
        print("  config5_g1:", json.dumps(d.get("config5_g1"))[:900])
        print("  locate:", {k:v for k,v in (d.get("locate") or {}).items() if k in ("hits_per_s","ms_per_batch","hits")})
    except Exception as ex:
        print(f, "ERR", ex)
PY
