#!/bin/bash
# one-walk-per-lane locate (RLFM config 4, 2^20 hits): blocks x threads per block (measurement build knobs
# FMX_EP_LOC_BLOCKS / FMX_EP_LOC_THREADS; the shipped library picks 256 x 1024 below 4M hits, 512 x 640 above)
cd "$(dirname "$0")/../.."
export FMX_LIB=$PWD/fm_index_amd/libfmx_measure.so
for thr in 1024 768 640 512; do for bl in 256 384 512 768; do
  echo -n "threads $thr blocks $bl: "
  FMX_EP_LOC_BLOCKS=$bl FMX_EP_LOC_THREADS=$thr python bench.py --workload bytes-rlfm --no-pmc --no-accel --no-3b --no-d2h --no-early-exit --no-cpu-baseline --no-census --no-rccl-check --steps 10 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][0]); l = d['locate']
print('ms_per_batch %.4f  kernel_ms %.4f  two_streams %.4f' % (l['ms_per_batch'], l['roofline']['avg_kernel_ms'], l['two_streams']['ms_per_batch']))"
done; done
