#!/bin/bash
# full -m gpu suite, then the default bench line (incl. the live PMC passes)
mkdir -p gpurun_out/r02
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r02/pytest_gpu.txt 2>&1
tail -15 gpurun_out/r02/pytest_gpu.txt
( time timeout 900 python bench.py > gpurun_out/r02/bench_default.json 2> gpurun_out/r02/bench_default.err ) 2> gpurun_out/r02/bench_default.time
tail -3 gpurun_out/r02/bench_default.err; cat gpurun_out/r02/bench_default.time
python - <<'PY'
import json
try:
    d = json.loads(open('gpurun_out/r02/bench_default.json').read().strip().splitlines()[-1])
    print(json.dumps({k: d[k] for k in ('value', 'ms_per_step', 'roofline', 'pmc', 'value_incl_d2h') if k in d}, indent=1)[:3000])
    print('locate', json.dumps(d.get('locate'))[:1500])
    print('3b', json.dumps(d.get('locate_3b'))[:800])
    print('rlfm', json.dumps(d.get('rlfm'))[:3000])
except Exception as ex:
    print('ERR', ex)
PY
