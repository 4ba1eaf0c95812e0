#!/bin/bash
# final differential soak of round 4 (shipped + range-checked library), smoke(), and the range-checked library on the
# wide / multi / RLFM test files
O=gpurun_out/r04_soak3; mkdir -p $O
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
for seed in 101 102 103 104 105 106; do timeout 200 python tests/fuzz_gpu_vs_oracle.py 100 $seed 2>&1 | tail -1; done > $O/soak.txt
for seed in 107 108 109; do FMX_LIB=$PWD/fm_index_amd/libfmx_debug.so timeout 200 python tests/fuzz_gpu_vs_oracle.py 100 $seed 2>&1 | tail -1; done >> $O/soak.txt
cat $O/soak.txt
FMX_LIB=$PWD/fm_index_amd/libfmx_debug.so timeout 900 python -m pytest tests/test_gpu_wide.py tests/test_multi_pieces.py tests/test_gpu_wide_symbols.py -q -m gpu > $O/pytest_debuglib.txt 2>&1; tail -3 $O/pytest_debuglib.txt
