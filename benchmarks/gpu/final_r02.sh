#!/bin/bash
# round-2 evidence: full -m gpu suite, the default bench line, rocprofv3 kernel stats + PMC summaries
mkdir -p gpurun_out/r02
timeout 2700 python -m pytest tests -m gpu -q > gpurun_out/r02/pytest_gpu.txt 2>&1
tail -6 gpurun_out/r02/pytest_gpu.txt
# the same kernels with every record / sample / piece index range-checked (libfmx_debug.so: -DFMX_DEBUG_BOUNDS)
FMX_LIB=$PWD/fm_index_amd/libfmx_debug.so timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_large_batches.py tests/test_naive_fixtures.py -m gpu -q > gpurun_out/r02/pytest_gpu_debuglib.txt 2>&1
tail -3 gpurun_out/r02/pytest_gpu_debuglib.txt
( time timeout 900 python bench.py > gpurun_out/r02/bench_default.json 2> gpurun_out/r02/bench_default.err ) 2> gpurun_out/r02/bench_default.time
tail -2 gpurun_out/r02/bench_default.err; grep real gpurun_out/r02/bench_default.time
bash profiles/run_rocprof.sh r02 > gpurun_out/r02/run_rocprof.log 2>&1
tail -3 gpurun_out/r02/run_rocprof.log
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r02/bench_default.json').read().strip().splitlines()[-1])
def rf(r): return {k: r.get(k) for k in ('kernel','avg_kernel_ms','traffic','achieved','frac','fabric_requests_per_s','frac_of_gather_ceiling','requested_lines','min_bytes','traffic_over_min_bytes')}
print('value', d['value'], 'ms', d['ms_per_step']); print(' roofline', rf(d['roofline']))
print('locate', d['locate']['hits_per_s'], d['locate']['ms_per_batch'], rf(d['locate']['roofline']))
print('3b', d.get('locate_3b'))
print('d2h', d.get('value_incl_d2h'), d.get('incl_d2h'))
r = d['rlfm']; print('rlfm', r['value'], r['ms_per_step'], rf(r['roofline'])); print(' rlfm locate', r['locate']['hits_per_s'], r['locate']['ms_per_batch'], rf(r['locate']['roofline'])); print(' rlfm cpu', r.get('cpu_baseline'))
print('cpu', d.get('cpu_baseline')); print('pmc', d.get('pmc'))
for k in ('early_exit','pair_index','kmer_table','kmer_table+pair_index'): print(k, d.get(k))
PY
