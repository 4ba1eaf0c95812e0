#!/bin/bash
# grid / walks-per-group sweep of the walk-record kernel (measurement build: FMX_LOC_BLOCKS, FMX_VARIANT=11/12/14)
O=gpurun_out/r04_walk; mkdir -p $O
python -m pytest tests/test_gpu_walk_records.py tests/test_gpu_text_order.py -x -q -m gpu > $O/pytest2.txt 2>&1; tail -3 $O/pytest2.txt
export FMX_LIB=$PWD/fm_index_amd/libfmx_measure.so ONLY=text_walk
for b in 256 512; do
  echo "== FMX_LOC_BLOCKS=$b"; FMX_LOC_BLOCKS=$b python benchmarks/gpu/walk_ab.py 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['index'],d['shape'],d['ms_per_batch'],d['walk_kernel_ms'])"
done 2>&1 | tee $O/sweep_blocks.txt
for v in; do
  echo "== FMX_VARIANT=$v (walks per group)"; FMX_VARIANT=$v python benchmarks/gpu/walk_ab.py 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['index'],d['shape'],d['ms_per_batch'],d['walk_kernel_ms'])"
done 2>&1 | tee $O/sweep_walks.txt
