#!/bin/bash
# the locate kernels on large batches: config 3b (2.9e8 hits) inside the default line, config 4b locate (7.9e8 hits)
mkdir -p gpurun_out/r02b
timeout 600 python bench.py --no-cpu-baseline --no-accel --no-early-exit --no-pmc --no-d2h --steps 10 > gpurun_out/r02b/default.json 2> gpurun_out/r02b/default.err
timeout 600 python bench.py --workload rep-rlfm --no-cpu-baseline --no-accel --no-pmc --no-census --steps 10 > gpurun_out/r02b/rep_rlfm.json 2> gpurun_out/r02b/rep_rlfm.err
timeout 600 python bench.py --workload rep-fm --no-cpu-baseline --no-accel --no-pmc --no-census --steps 10 > gpurun_out/r02b/rep_fm.json 2> gpurun_out/r02b/rep_fm.err
python - <<'PY'
import json
for f in ('default','rep_rlfm','rep_fm'):
    try:
        d=json.loads(open('gpurun_out/r02b/%s.json'%f).read().strip().splitlines()[-1])
        print(f,'count ms',round(d['ms_per_step'],4),'locate',{k:d['locate'][k] for k in ('hits','hits_per_s','ms_per_batch')}, 'kernel', d['locate']['roofline']['avg_kernel_ms'])
        if 'locate_3b' in d: print('  3b', d['locate_3b'])
        if 'rlfm' in d: print('  rlfm', d['rlfm']['ms_per_step'], d['rlfm']['locate']['ms_per_batch'], d['rlfm']['locate']['roofline'])
        if f=='default': print('  roofline', d['roofline']); print('  locate roofline', d['locate']['roofline'])
    except Exception as ex:
        print(f,'ERR',ex, open('gpurun_out/r02b/%s.err'%f).read()[-400:])
PY
