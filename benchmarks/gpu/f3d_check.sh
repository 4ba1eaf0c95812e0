#!/bin/bash
# distributed-state DNA count kernel (measurement build: FMX_VARIANT=24 four patterns per group, 25 two):
# parity on the count tests, then A/B against the shipped group-per-pattern kernel
export FMX_LIB=$PWD/fm_index_amd/libfmx_measure.so
for v in 24 25; do
  FMX_VARIANT=$v timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_naive_fixtures.py tests/test_gpu_advice_r1.py -x -q -m gpu 2>&1 | tail -2
done
O=gpurun_out/f3d; mkdir -p $O
for v in x 24 25; do
  if [ "$v" = x ]; then unset FMX_VARIANT; else export FMX_VARIANT=$v; fi
  timeout 600 python bench.py --no-pmc --no-census --no-cpu-baseline --no-accel --no-d2h --no-rlfm --no-3b --no-locate > $O/v$v.json 2> $O/v$v.err
  python - $O/v$v.json $v <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print('variant', sys.argv[2], 'count ms', round(d['ms_per_step'], 4), 'kernel', d['roofline'].get('avg_kernel_ms'), 'early exit', (d.get('early_exit') or {}).get('kernel_ms'))
except Exception as ex:
    print(sys.argv[2], 'ERR', ex, open(sys.argv[1].replace('.json', '.err')).read()[-600:])
PY
done
