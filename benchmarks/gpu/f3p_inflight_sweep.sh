#!/bin/bash
# config 3 (2^20 hits) on the shipped DNA walk kernel with FEWER walks in flight: does a shorter queue
# (lower loaded latency per step of the longest walks) beat more parallelism?  measurement build
O=gpurun_out/f3pi; mkdir -p $O
export FMX_LIB=$PWD/fm_index_amd/libfmx_measure.so
for cfg in "1024 256" "1024 128" "1024 192" "512 256" "512 384" "256 512" "256 768" "128 1024" "128 2048"; do
  set -- $cfg
  FMX_LOC_THREADS=$1 FMX_LOC_BLOCKS=$2 timeout 600 python bench.py --no-pmc --no-census --no-cpu-baseline --no-accel --no-early-exit --no-d2h --no-rlfm --no-3b > $O/t$1_b$2.json 2> $O/t$1_b$2.err
  python - $O/t$1_b$2.json "$cfg" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    l = d['locate']
    print('threads blocks', sys.argv[2], '| config 3 ms/batch', round(l['ms_per_batch'], 4), 'kernel alone', (l.get('roofline') or {}).get('avg_kernel_ms'))
except Exception as ex:
    print(sys.argv[2], 'ERR', ex)
PY
done
