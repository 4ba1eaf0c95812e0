#!/bin/bash
# DNA (one-level) locate on the one-walk-per-lane kernel (measurement build, FMX_VARIANT=21) against the
# group-per-walk kernel fmx_locate_f3w_kernel, over the number of blocks (= walks in flight)
O=gpurun_out/epdna; mkdir -p $O
export FMX_LIB=$PWD/fm_index_amd/libfmx_measure.so
run() {  # tag, env...
  tag=$1; shift
  env "$@" timeout 600 python bench.py --no-pmc --no-census --no-cpu-baseline --no-accel --no-early-exit --no-d2h --no-rlfm > $O/$tag.json 2> $O/$tag.err
  python - $O/$tag.json $tag <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    l, b = d['locate'], d.get('locate_3b') or {}
    print(sys.argv[2], 'locate ms', round(l['ms_per_batch'], 4), 'kernel', (l.get('roofline') or {}).get('avg_kernel_ms'), '3b ms', round(b.get('ms_per_batch', 0), 2))
except Exception as ex:
    print(sys.argv[2], 'ERR', ex)
PY
}
run f3w FMX_NOOP=1
for nb in 64 96 128 160 192 256 384; do run ep_$nb FMX_VARIANT=21 FMX_EP_LOC_BLOCKS=$nb; done
