#!/bin/bash
# one-walk-per-lane locate with the lane-wise probes chained in front of the LF step: parity, then numbers
timeout 1800 python -m pytest tests/test_gpu_rlfm.py tests/test_gpu_text_order.py tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_fullsize.py -x -q -m gpu 2>&1 | tail -3
O=gpurun_out/epc; mkdir -p $O
for wlk in dna rep-rlfm bytes-fm; do
  timeout 900 python bench.py --workload $wlk --no-pmc --no-census --no-cpu-baseline --no-accel --no-early-exit --no-d2h --no-3b > $O/$wlk.json 2> $O/$wlk.err
  python - $O/$wlk.json $wlk <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    l, r = d.get('locate') or {}, d.get('rlfm') or {}
    print(sys.argv[2], 'count ms', round(d['ms_per_step'], 4), 'locate', {k: l.get(k) for k in ('ms_per_batch', 'hits', 'hits_per_s', 'lf_steps')}, (l.get('roofline') or {}).get('avg_kernel_ms'))
    if r: print('  rlfm count ms', r.get('ms_per_step'), 'locate', {k: (r.get('locate') or {}).get(k) for k in ('ms_per_batch', 'hits_per_s', 'lf_steps')})
except Exception as ex:
    print(sys.argv[2], 'ERR', ex)
PY
done
