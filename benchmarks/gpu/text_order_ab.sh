#!/bin/bash
# text-order sampling (shipped) against row-order sampling (measurement build, FMX_VARIANT=18) on one box
O=gpurun_out/r02ab; mkdir -p $O
timeout 900 python bench.py --no-pmc --no-census > $O/text.json 2> $O/text.err
FMX_LIB=$PWD/fm_index_amd/libfmx_measure.so FMX_VARIANT=18 timeout 900 python bench.py --no-pmc --no-census > $O/row.json 2> $O/row.err
python - <<'PY'
import json
for name in ('text', 'row'):
    try:
        d = json.loads(open('gpurun_out/r02ab/%s.json' % name).read().strip().splitlines()[-1])
        l, b, r = d['locate'], d['locate_3b'], d['rlfm']
        print(name, 'count ms', d['ms_per_step'], 'index_bytes', d['config'].get('index_bytes'))
        print('  locate', {k: l.get(k) for k in ('ms_per_batch', 'kernel_ms', 'hits_per_s', 'lf_steps')})
        print('  3b', {k: b.get(k) for k in ('ms_per_batch', 'hits_per_s', 'lf_steps')})
        print('  rlfm count ms', r.get('ms_per_step'), 'locate', {k: (r.get('locate') or {}).get(k) for k in ('ms_per_batch', 'hits_per_s', 'lf_steps')})
    except Exception as ex:
        print(name, 'ERR', ex)
PY
tail -3 $O/text.err $O/row.err
