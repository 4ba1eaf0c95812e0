#!/bin/bash
# round-2 measurement batch: full -m gpu suite, DNA locate occupancy sweep, FETCH_SIZE calibration on
# small random requests
mkdir -p gpurun_out/r02
timeout 2700 python -m pytest tests -m gpu -q > gpurun_out/r02/pytest_gpu.txt 2>&1
tail -8 gpurun_out/r02/pytest_gpu.txt
export FMX_LIB=$PWD/fm_index_amd/libfmx_measure.so
B="python bench.py --no-cpu-baseline --no-accel --no-early-exit --no-rlfm --no-3b --no-d2h --no-pmc --no-census --steps 10"
for v in 12 14; do for w in 1024 2048 4096 8192; do
  FMX_VARIANT=$v FMX_LOC_WAVES=$w timeout 200 $B > gpurun_out/r02/loc_v${v}_w${w}.json 2> gpurun_out/r02/loc_v${v}_w${w}.err
done; done
unset FMX_LIB
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r02/loc_v*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, 'batch ms', round(d['locate']['ms_per_batch'],4), 'kernel ms', d['locate']['roofline']['avg_kernel_ms'])
    except Exception as ex:
        print(f,'ERR',ex)
PY
# FETCH_SIZE per request for 16 / 32 / 64 / 128-byte random requests (2 GiB table)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/gcal --output-format csv -- $GRAFT_REPO_ROOT/profiles/microbench/gather 2048 > $GRAFT_REPO_ROOT/gpurun_out/r02/gather_pmc_stdout.txt 2> $GRAFT_REPO_ROOT/gpurun_out/r02/gather_pmc.err
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv,glob,collections
agg=collections.OrderedDict()
for f in glob.glob('/tmp/gcal/**/*counter_collection*.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        k=(row['Kernel_Name'].split('(')[0], row.get('Grid_Size'))
        agg.setdefault(k,[]).append(float(row['Counter_Value']))
out=open('gpurun_out/r02/gather_fetch_calibration.txt','w')
for (k,g),v in agg.items():
    # every kernel is launched twice per configuration: warm-up (8 or 4 steps) and 256 steps; take the larger
    line="%s grid=%s launches=%d max_FETCH_KB=%.1f" % (k,g,len(v),max(v))
    print(line); out.write(line+"\n")
PY
