#!/bin/bash
mkdir -p gpurun_out/ep1
timeout 900 python -m pytest tests/test_gpu_rlfm.py -x -q > gpurun_out/ep1/pytest_rlfm.txt 2>&1
tail -3 gpurun_out/ep1/pytest_rlfm.txt
export FMX_LIB=$PWD/fm_index_amd/libfmx_measure.so
for wl in bytes-rlfm rep-rlfm; do
  FMX_VARIANT=0 timeout 300 python bench.py --workload $wl --no-cpu-baseline --no-accel --steps 10 > gpurun_out/ep1/${wl}_v0.json 2> gpurun_out/ep1/${wl}_v0.err
  for b in 256 512 1024 2048; do
    FMX_EP_BLOCKS=$b timeout 300 python bench.py --workload $wl --no-cpu-baseline --no-accel --steps 10 > gpurun_out/ep1/${wl}_ep$b.json 2> gpurun_out/ep1/${wl}_ep$b.err
  done
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/ep1/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, 'count ms', round(d['ms_per_step'],4), 'locate ms', round(d.get('locate',{}).get('ms_per_batch',0),4), 'kern', d.get('locate',{}).get('roofline',{}).get('avg_kernel_ms'), 'hits', d.get('locate',{}).get('hits'))
    except Exception as ex:
        print(f, 'ERR', ex)
PY
