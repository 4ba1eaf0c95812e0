#!/bin/bash
# DNA locate on small / mid-size texts and batches: the default index (text order + walk records) against row order
O=gpurun_out/r04_small; mkdir -p $O
python benchmarks/gpu/small_shapes.py dna > $O/dna_default.jsonl 2>/dev/null
SAMPLING=row python benchmarks/gpu/small_shapes.py dna > $O/dna_row.jsonl 2>/dev/null
python - <<'PY'
import json
a=[json.loads(l) for l in open("gpurun_out/r04_small/dna_default.jsonl")]
b=[json.loads(l) for l in open("gpurun_out/r04_small/dna_row.jsonl")]
print("log2n log2npat hits  locate_us default / row")
for x,y in zip(a,b):
    print(x["log2n"], x["log2npat"], x["hits"], x["locate_us"], y["locate_us"], "  count", x["count_us"], y["count_us"])
PY
