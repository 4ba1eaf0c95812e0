#!/bin/bash
# fmx_count_batch on page-locked arrays: upload by DMA or by copy kernels x chunks x grid cap of the chunk's search
# x blocks of the copy kernels (tuning build of fmx_api.hip, -DFMX_TUNE_HOSTPIPE; the shipped library has the
# chosen values compiled in)
cd "$(dirname "$0")/../.."
LIB=$PWD/benchmarks/gpu/libfmx_tune.so
for dma in ${DMA:-1 0}; do for ch in ${CHUNKS:-4 8}; do for bl in ${SEARCH:-1792 2048}; do for cb in ${COPY:-32 64}; do
  echo -n "h2d_dma $dma chunks $ch search_blocks $bl copy_blocks $cb: "
  FMX_LIB=$LIB FMX_PIPE_H2D=$dma FMX_PIPE_CHUNKS=$ch FMX_PIPE_BLOCKS=$bl FMX_PIPE_COPY_BLOCKS=$cb python benchmarks/host_pointer_rate.py 2>/dev/null | tail -1 | cut -c1-200
done; done; done; done
