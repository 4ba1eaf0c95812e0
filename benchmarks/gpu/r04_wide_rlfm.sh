#!/bin/bash
# wide RLFM at n = 2^32 + 2^20: the two shapes of the run-table walk
O=gpurun_out/r04_wrl; mkdir -p $O
for v in 0 1; do
FMXW_RL_LOCKSTEP=$v python benchmarks/gpu/wide_rlfm.py > $O/wide_rlfm_4g_lockstep$v.json 2> $O/wide_rlfm_4g.err; tail -2 $O/wide_rlfm_4g.err; cat $O/wide_rlfm_4g_lockstep$v.json
done
