#!/bin/bash
# text-order against row-order sampling of the one-level DNA index on one box: since round 3 a build flag of the
# shipped library (FMX_FLAG_TEXT_ORDER), reported by bench.py itself as the `locate_text_order` leg next to `locate`
O=gpurun_out/text_order_ab; mkdir -p $O
timeout 900 python bench.py --no-pmc --no-census --no-cpu-baseline --no-early-exit --no-d2h --no-rlfm --no-3b $1 > $O/bench.json 2> $O/bench.err
python - <<'PY'
import json
d = json.loads(open('gpurun_out/text_order_ab/bench.json').read().strip().splitlines()[-1])
l, t = d['locate'], d['locate_text_order']
print('row order ', {k: l.get(k) for k in ('ms_per_batch', 'hits_per_s', 'lf_steps')}, 'index_bytes', d['config']['index_bytes'])
print('text order', {k: t.get(k) for k in ('ms_per_batch', 'hits_per_s', 'lf_steps', 'index_bytes')})
PY
