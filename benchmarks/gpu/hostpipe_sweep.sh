#!/bin/bash
# fmx_count_batch on page-locked arrays: offsets on their own copy stream or not x chunks x grid cap of the chunk's
# search (tuning build of fmx_api.hip, -DFMX_TUNE_HOSTPIPE -> benchmarks/gpu/libfmx_tune.so; the shipped library has
# the chosen values compiled in).  Other knobs: FMX_PIPE_H2D=0 (upload by copy kernels), FMX_PIPE_COPY_BLOCKS
cd "$(dirname "$0")/../.."
LIB=$PWD/benchmarks/gpu/libfmx_tune.so
for os in ${OFFS:-1 0}; do for ch in ${CHUNKS:-4 8}; do for bl in ${SEARCH:-1792}; do for rep in 1 2; do
  echo -n "off_stream $os chunks $ch search_blocks $bl: "
  FMX_LIB=$LIB FMX_PIPE_OFF_STREAM=$os FMX_PIPE_CHUNKS=$ch FMX_PIPE_BLOCKS=$bl python benchmarks/host_pointer_rate.py 2>/dev/null | tail -1 | cut -c1-200
done; done; done; done
