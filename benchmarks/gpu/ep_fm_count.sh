#!/bin/bash
# endpoint-per-lane count on FM indexes: wide alphabets (>= 3 levels, shipped path) and, as an experiment
# (FMX_VARIANT=20, measurement build), the single-level DNA headline
mkdir -p gpurun_out/r02e
timeout 900 python -m pytest tests/test_gpu_wide_symbols.py tests/test_gpu_fuzz.py -m gpu -q > gpurun_out/r02e/pytest_wide.txt 2>&1
tail -3 gpurun_out/r02e/pytest_wide.txt
export FMX_LIB=$PWD/fm_index_amd/libfmx_measure.so
FMX_VARIANT=20 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_kmer_table.py -m gpu -q > gpurun_out/r02e/pytest_v20.txt 2>&1
tail -3 gpurun_out/r02e/pytest_v20.txt
B="python bench.py --no-cpu-baseline --no-accel --no-rlfm --no-3b --no-d2h --no-pmc --no-census --no-locate --steps 20"
timeout 200 $B > gpurun_out/r02e/dna_f3.json 2> gpurun_out/r02e/dna_f3.err
for b in 512 1024 2048; do
  FMX_VARIANT=20 FMX_EP_BLOCKS=$b timeout 200 $B > gpurun_out/r02e/dna_ep$b.json 2> gpurun_out/r02e/dna_ep$b.err
  FMX_VARIANT=20 FMX_EP_BLOCKS=$b timeout 200 $B --workload bytes-fm > gpurun_out/r02e/bfm_ep$b.json 2> gpurun_out/r02e/bfm_ep$b.err
done
timeout 200 $B --workload bytes-fm > gpurun_out/r02e/bfm_old.json 2> gpurun_out/r02e/bfm_old.err
unset FMX_LIB
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r02e/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, 'count ms', round(d['ms_per_step'],4), 'early-exit ms', d.get('early_exit',{}).get('kernel_ms'))
    except Exception as ex:
        print(f,'ERR',ex, open(f.replace('.json','.err')).read()[-300:])
PY
