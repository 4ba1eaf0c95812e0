#!/bin/bash
# small and mid-size DNA batches: distributed-state walk kernel (shipped dispatch) against the
# replicated-state kernel (FMX_VARIANT=22), measurement build
O=gpurun_out/f3q; mkdir -p $O
export FMX_LIB=$PWD/fm_index_amd/libfmx_measure.so
timeout 900 python benchmarks/gpu/small_shapes.py dna > $O/small_f3q.jsonl 2> $O/small_f3q.err
FMX_VARIANT=22 timeout 900 python benchmarks/gpu/small_shapes.py dna > $O/small_f3w.jsonl 2> $O/small_f3w.err
python - <<'PY'
import json
def rd(p):
    out = {}
    for l in open(p):
        try: d = json.loads(l)
        except Exception: continue
        out[(d.get('log2n'), d.get('log2npat'))] = d
    return out
a, b = rd('gpurun_out/f3q/small_f3q.jsonl'), rd('gpurun_out/f3q/small_f3w.jsonl')
for k in sorted(a):
    print(k, {x: (round(a[k][x], 1), round(b.get(k, {}).get(x, 0), 1)) for x in a[k] if x.endswith('_us')}, 'hits', a[k].get('hits'))
PY
tail -n 2 $O/small_f3q.err
