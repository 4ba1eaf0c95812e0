#!/bin/bash
# page-locked fmx_count_batch: results written by the search (FMX_PIPE_DIRECT_OUT 0 / 1 / 2) x chunks
cd "$(dirname "$0")/../.."
O=gpurun_out/r04_hostpipe; mkdir -p $O
LIB=$PWD/benchmarks/gpu/libfmx_tune.so
for d in 0 1 2; do for ch in 4 6 8; do for rep in 1 2; do
  echo -n "direct_out $d chunks $ch: "
  FMX_LIB=$LIB FMX_PIPE_DIRECT_OUT=$d FMX_PIPE_CHUNKS=$ch python benchmarks/host_pointer_rate.py 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print(d['page_locked_ms_per_call'], d['page_locked_ms_per_call_counts_only'], d['ms_per_call_reused_buffers'])"
done; done; done 2>&1 | tee $O/sweep_direct.txt
python -m pytest tests/test_gpu_concurrency.py -x -q -m gpu 2>&1 | tail -3
