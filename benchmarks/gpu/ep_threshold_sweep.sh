#!/bin/bash
# from how many hits does the one-walk-per-lane locate kernel beat the group-per-walk kernel?
# (measurement build: FMX_RL_EP_MIN / FMX_FM_EP_MIN = smallest batch routed to it)
O=gpurun_out/epth; mkdir -p $O
export FMX_LIB=$PWD/fm_index_amd/libfmx_measure.so
FMX_RL_EP_MIN=0 FMX_FM_EP_MIN=0 timeout 900 python benchmarks/gpu/small_shapes.py rlfm fm2 > $O/always.jsonl 2> $O/always.err
FMX_RL_EP_MIN=4000000000 FMX_FM_EP_MIN=4000000000 timeout 900 python benchmarks/gpu/small_shapes.py rlfm fm2 > $O/never.jsonl 2> $O/never.err
python - <<'PY'
import json
def rd(p):
    out = {}
    for l in open(p):
        try: d = json.loads(l)
        except Exception: continue
        out[(d['index'], d['log2n'], d['log2npat'])] = d
    return out
a, b = rd('gpurun_out/epth/always.jsonl'), rd('gpurun_out/epth/never.jsonl')
for k in sorted(a):
    print(k, 'hits', a[k]['hits'], 'locate us: per-lane', a[k]['locate_us'], 'group-per-walk', b.get(k, {}).get('locate_us'))
PY
