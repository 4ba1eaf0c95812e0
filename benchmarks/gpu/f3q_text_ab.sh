#!/bin/bash
# one-level DNA index, distributed-state walk kernel: row-order (18) against text-order (19) sampling
for nb in 256 512; do
  FMX_LOC_BLOCKS=$nb bash benchmarks/gpu/variant_ab.sh "18 19" --no-rlfm | sed "s/^/blocks $nb: /"
done
