#!/bin/bash
# wide RLFM with text-order samples: parity tests, fuzz, the beyond-4G protocol, the benchmark
O=gpurun_out/r04_wrt; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_wide.py -x -q -k "rlfm or refusals" 2>&1 | tail -5 > $O/pytest.txt; cat $O/pytest.txt
for seed in 81 82; do timeout 200 python tests/fuzz_gpu_vs_oracle.py 45 $seed 2>&1 | tail -2; done > $O/fuzz.txt; cat $O/fuzz.txt
timeout 600 python tests/test_gpu_beyond_4g.py rlfm > $O/beyond_4g_rlfm.json 2> $O/beyond_4g_rlfm.err; tail -3 $O/beyond_4g_rlfm.err; cat $O/beyond_4g_rlfm.json
timeout 600 python benchmarks/gpu/wide_rlfm.py > $O/wide_rlfm_4g_text.json 2> $O/wide_rlfm_4g.err; tail -2 $O/wide_rlfm_4g.err; cat $O/wide_rlfm_4g_text.json
