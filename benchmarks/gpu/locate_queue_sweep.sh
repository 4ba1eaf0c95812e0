#!/bin/bash
# locate kernels with the per-block LDS hit queue: parity tests that touch locate, then block-count sweeps
mkdir -p gpurun_out/r02q
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_rlfm.py tests/test_naive_fixtures.py tests/test_gpu_wide_symbols.py -m gpu -q > gpurun_out/r02q/pytest.txt 2>&1
tail -5 gpurun_out/r02q/pytest.txt
export FMX_LIB=$PWD/fm_index_amd/libfmx_measure.so
B="python bench.py --no-cpu-baseline --no-accel --no-early-exit --no-rlfm --no-3b --no-d2h --no-pmc --no-census --steps 10"
rm -f gpurun_out/r02q/*.json gpurun_out/r02q/*.err
for v in 12 14; do for w in 192 256; do
  FMX_VARIANT=$v FMX_LOC_BLOCKS=$w timeout 200 $B > gpurun_out/r02q/dna_v${v}_b${w}.json 2> gpurun_out/r02q/dna_v${v}_b${w}.err
done; done
for b in 128 192 256; do
  FMX_EP_LOC_BLOCKS=$b timeout 200 $B --workload bytes-rlfm > gpurun_out/r02q/rlfm_b${b}.json 2> gpurun_out/r02q/rlfm_b${b}.err
  FMX_EP_LOC_BLOCKS=$b timeout 200 $B --workload bytes-fm > gpurun_out/r02q/bfm_b${b}.json 2> gpurun_out/r02q/bfm_b${b}.err
done
FMX_VARIANT=0 timeout 200 $B --workload bytes-fm > gpurun_out/r02q/bfm_v0.json 2> gpurun_out/r02q/bfm_v0.err
unset FMX_LIB
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r02q/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, 'count ms', round(d['ms_per_step'],4), 'locate batch ms', round(d['locate']['ms_per_batch'],4), 'kernel ms', d['locate']['roofline']['avg_kernel_ms'])
    except Exception as ex:
        print(f,'ERR',ex, open(f.replace('.json','.err')).read()[-300:])
PY
