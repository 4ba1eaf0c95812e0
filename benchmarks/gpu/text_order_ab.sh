#!/bin/bash
# text-order sampling against row-order sampling on one box, through the measurement build:
# FMX_VARIANT=18 row order everywhere, 19 text order everywhere (the shipped builder picks per index kind)
O=gpurun_out/r02ab; mkdir -p $O
export FMX_LIB=$PWD/fm_index_amd/libfmx_measure.so
for v in 18 19; do
  FMX_VARIANT=$v timeout 900 python bench.py --no-pmc --no-census --no-cpu-baseline --no-accel --no-early-exit --no-d2h $1 > $O/v$v.json 2> $O/v$v.err
done
python - <<'PY'
import json
for name in ('v18', 'v19'):
    try:
        d = json.loads(open('gpurun_out/r02ab/%s.json' % name).read().strip().splitlines()[-1])
        l, b, r = d['locate'], d.get('locate_3b') or {}, d.get('rlfm') or {}
        print(name, 'count ms', d['ms_per_step'], 'index_bytes', d['config'].get('index_bytes'))
        print('  locate', {k: l.get(k) for k in ('ms_per_batch', 'hits_per_s', 'lf_steps')})
        print('  3b', {k: b.get(k) for k in ('ms_per_batch', 'hits_per_s', 'lf_steps')})
        print('  rlfm count ms', r.get('ms_per_step'), 'locate', {k: (r.get('locate') or {}).get(k) for k in ('ms_per_batch', 'hits_per_s', 'lf_steps')})
    except Exception as ex:
        print(name, 'ERR', ex)
PY
tail -n 3 $O/v18.err $O/v19.err
