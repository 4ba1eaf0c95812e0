#!/bin/bash
# bench.py on the measurement build under a list of FMX_VARIANT values (default dispatch = "x"):
#   bash benchmarks/gpu/variant_ab.sh "x 21" [extra bench flags]
O=gpurun_out/vab; mkdir -p $O
export FMX_LIB=$PWD/fm_index_amd/libfmx_measure.so
for v in $1; do
  if [ "$v" = x ]; then unset FMX_VARIANT; else export FMX_VARIANT=$v; fi
  timeout 900 python bench.py --no-pmc --no-census --no-cpu-baseline --no-accel --no-early-exit --no-d2h $2 > $O/v$v.json 2> $O/v$v.err
  python - $O/v$v.json $v <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    l, b, r = d['locate'], d.get('locate_3b') or {}, d.get('rlfm') or {}
    print('variant', sys.argv[2], 'count ms', round(d['ms_per_step'], 4), 'index_bytes', d['config'].get('index_bytes'))
    print('  locate', {k: l.get(k) for k in ('ms_per_batch', 'hits_per_s', 'lf_steps')}, (l.get('roofline') or {}).get('avg_kernel_ms'))
    print('  3b', {k: b.get(k) for k in ('ms_per_batch', 'hits_per_s', 'lf_steps')})
    if r: print('  rlfm count ms', r.get('ms_per_step'), 'locate', {k: (r.get('locate') or {}).get(k) for k in ('ms_per_batch', 'hits_per_s', 'lf_steps')})
except Exception as ex:
    print(sys.argv[2], 'ERR', ex)
PY
done
