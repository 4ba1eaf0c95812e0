#!/bin/bash
OUT=gpurun_out/final_r03
mkdir -p $OUT
timeout 1800 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.txt 2>&1; tail -3 $OUT/pytest_gpu.txt
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$?"
python bench.py --force-dist > $OUT/bench_force_dist.json 2> $OUT/bench_force_dist.err; echo "bench force-dist rc=$?"
python tests/test_gpu_beyond_4g.py dna > $OUT/beyond_4g.json 2>/dev/null; echo "beyond_4g dna rc=$?"
python tests/test_gpu_beyond_4g.py bytes > $OUT/beyond_4g_bytes.json 2>/dev/null; echo "beyond_4g bytes rc=$?"
bash profiles/run_rocprof.sh r03 > $OUT/rocprof.log 2>&1; echo "rocprof rc=$?"
T="tests/test_gpu_wide.py tests/test_gpu_parity.py tests/test_naive_fixtures.py"
FMX_LIB=$PWD/fm_index_amd/libfmx_debug.so timeout 1200 python -m pytest $T -m gpu -q > $OUT/pytest_debuglib_wide_parity.txt 2>&1; tail -2 $OUT/pytest_debuglib_wide_parity.txt
