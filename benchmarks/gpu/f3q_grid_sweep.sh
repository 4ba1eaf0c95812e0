#!/bin/bash
# DNA walk kernel with distributed walk state: threads per block x blocks (measurement build)
O=gpurun_out/f3q; mkdir -p $O
export FMX_LIB=$PWD/fm_index_amd/libfmx_measure.so
for cfg in "1024 256" "1024 384" "1024 512" "512 512" "512 768" "512 1024" "256 1024" "256 2048" "768 512" "640 768"; do
  set -- $cfg
  FMX_LOC_THREADS=$1 FMX_LOC_BLOCKS=$2 timeout 600 python bench.py --no-pmc --no-census --no-cpu-baseline --no-accel --no-early-exit --no-d2h --no-rlfm > $O/t$1_b$2.json 2> $O/t$1_b$2.err
  python - $O/t$1_b$2.json "$cfg" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    l, b = d['locate'], d.get('locate_3b') or {}
    print('threads blocks', sys.argv[2], '| config 3 ms/batch', round(l['ms_per_batch'], 4), 'kernel', (l.get('roofline') or {}).get('avg_kernel_ms'), '| 3b ms', round(b.get('ms_per_batch', 0), 2))
except Exception as ex:
    print(sys.argv[2], 'ERR', ex)
PY
done
