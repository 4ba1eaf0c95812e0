#!/bin/bash
# One parameterised entry for the GPU-box runs of a round (replaces the single-use r0x_*.sh scripts):
#   gpurun --timeout S -- 'bash benchmarks/gpu/run.sh <round> <task> [args...]'
# Outputs go to gpurun_out/<round>/ (scratch); what is kept is copied to profiles/<round>/ by hand.
# Every command is bounded by `timeout`.
#   driver          the driver's exact bench command; raw stdout/stderr + the detail file + line length
#   suite [expr]    pytest -m gpu (optionally -k expr), durations of the slowest tests
#   tests <files>   pytest -m gpu on the named files
#   rocprof         profiles/run_rocprof.sh <round> (kernel-trace --stats of the bench command)
#   soak <seconds> [seed] [lib|shipped] [long]   tests/fuzz_gpu_vs_oracle.py (optionally on another libfmx*.so; `long` =
#                        the long-interval batches: lane-per-walk kernels, per-ticket dispatch, every select branch)
#   variant "<envs>" [flags]   bench.py on the measurement build under each environment of the list (FMX_VARIANT=.., grid knobs)
#   ablib "<libs>" [flags]     bench.py once per library of the list (file names under fm_index_amd/, e.g. a copy of the last
#                        build next to the new one), same box, in the order given: A/B of two builds of a kernel
#   pmc             benchmarks/gpu/kernel_pmc.py: counters per query kernel (what bounds it)
#   mix [args]      benchmarks/gpu/locate_mix.py (DNA and RLFM): shipped library, then the measurement build on each path
#   py <script> [args]   any python script under benchmarks/
R=$1; T=$2; shift 2
O=gpurun_out/$R; mkdir -p $O
case $T in
driver)
  ( time timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 --detail-out $O/bench_detail.json ) > $O/driver_stdout.txt 2> $O/driver_stderr.txt
  echo "rc $? ; stdout lines $(wc -l < $O/driver_stdout.txt) ; last line bytes $(tail -n 1 $O/driver_stdout.txt | wc -c)"
  tail -n 1 $O/driver_stdout.txt; tail -n 4 $O/driver_stderr.txt ;;
suite)
  timeout 1500 python3 -m pytest tests -m gpu -x -q --durations=25 ${1:+-k "$1"} > $O/suite.txt 2>&1; echo "rc $?"; tail -n 40 $O/suite.txt ;;
tests)
  timeout 1500 python3 -m pytest "$@" -m gpu -x -q --durations=10 > $O/tests.txt 2>&1; echo "rc $?"; tail -n 25 $O/tests.txt ;;
rocprof)
  timeout 1200 bash profiles/run_rocprof.sh $R "$@" ;;
soak)
  SEC=$1; SEED=${2:-1}; LIB=$3; MODE=$4
  TAG=${LIB:-shipped}${MODE:+_long}_seed$SEED
  ( [ -n "$LIB" ] && [ "$LIB" != shipped ] && export FMX_LIB=$PWD/fm_index_amd/$LIB; timeout $((SEC + 600)) python3 tests/fuzz_gpu_vs_oracle.py $SEC $SEED ${MODE:+--long} ) > $O/soak_$TAG.txt 2>&1
  echo "rc $?"; tail -n 4 $O/soak_$TAG.txt ;;
mix)
  # locate on mixed batches: shipped library, then the measurement build forced onto each path
  M=$PWD/fm_index_amd/libfmx_measure.so
  for K in ${MIX_KINDS:-dna rlfm rlfm-random}; do
    timeout 600 python3 benchmarks/gpu/locate_mix.py --kind $K "$@" 2>&1 | tail -n 1
    FMX_LIB=$M FMX_VARIANT=28 timeout 600 python3 benchmarks/gpu/locate_mix.py --kind $K "$@" 2>&1 | tail -n 1
    FMX_LIB=$M FMX_ADJ_CLUSTERS=0 timeout 600 python3 benchmarks/gpu/locate_mix.py --kind $K "$@" 2>&1 | tail -n 1
    FMX_LIB=$M FMX_ADJ_CLUSTERS=65 timeout 600 python3 benchmarks/gpu/locate_mix.py --kind $K "$@" 2>&1 | tail -n 1
  done | tee $O/locate_mix.jsonl ;;
rlusweep)
  M=$PWD/fm_index_amd/libfmx_measure.so
  for SL in 2048 4096 16384; do for TH in 256 512 1024; do
    echo "slice $SL threads $TH"; FMX_LIB=$M FMX_RLU_SLICE=$SL FMX_RLU_THREADS=$TH timeout 600 python3 benchmarks/gpu/locate_mix.py --kind rlfm "$@" 2>&1 | tail -n 1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print({k:(d[k]['ms']) for k in ('pure_singletons','pure_long','mixed','mixed_low_average')})"
  done; done | tee $O/rlu_sweep.txt ;;
variant)
  # bench.py on the measurement build under a list of environments ("x" = the default dispatch), one summary line each:
  #   run.sh r05 variant "x FMX_VARIANT=28 FMX_ADJ_CLUSTERS=0,FMX_LOC_BLOCKS=512" [bench flags]
  LIST=$1; shift
  export FMX_LIB=$PWD/fm_index_amd/libfmx_measure.so
  for v in $LIST; do
    T=$(echo $v | tr -c 'A-Za-z0-9=\n' _)
    ( [ "$v" != x ] && export $(echo $v | tr , ' '); timeout 900 python3 bench.py --no-pmc --no-census --no-cpu-baseline --no-accel --no-early-exit --no-d2h --no-wide --no-ic-ab --no-config5 --no-rccl-check "$@" --detail-out $O/variant_$T.json > /dev/null 2> $O/variant_$T.err )
    python3 - $O/variant_$T.json "$v" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    l, b, r = d.get('locate') or {}, d.get('locate_3b') or {}, d.get('rlfm') or {}
    print(sys.argv[2], 'count ms', round(d['ms_per_step'], 4), '| locate', {k: l.get(k) for k in ('ms_per_batch', 'walk_kernel_ms')},
          '| 3b', {k: b.get(k) for k in ('ms_per_batch',)}, '| rlfm', r.get('ms_per_step'), {k: (r.get('locate') or {}).get(k) for k in ('ms_per_batch',)})
except Exception as ex:
    print(sys.argv[2], 'ERR', ex)
PY
  done | tee $O/variants.txt ;;
ablib)
  LIST=$1; shift
  for v in $LIST; do
    ( export FMX_LIB=$PWD/fm_index_amd/$v; timeout 900 python3 bench.py --no-pmc --no-census --no-cpu-baseline --no-accel --no-early-exit --no-d2h --no-wide --no-ic-ab --no-config5 --no-rccl-check "$@" --detail-out $O/ablib_$v.json > /dev/null 2> $O/ablib_$v.err )
    python3 - $O/ablib_$v.json "$v" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    l, b, r = d.get('locate') or {}, d.get('locate_3b') or {}, d.get('rlfm') or {}
    print(sys.argv[2], 'count ms', round(d['ms_per_step'], 4), '| locate', {k: l.get(k) for k in ('ms_per_batch', 'walk_kernel_ms')},
          '| 3b', {k: b.get(k) for k in ('ms_per_batch', 'walk_kernel_ms')}, '| positions', (l.get('positions_sha256') or '')[:12], (b.get('positions_sha256') or '')[:12])
except Exception as ex:
    print(sys.argv[2], 'ERR', ex)
PY
  done | tee -a $O/ablib.txt ;;
pmc)
  # what bounds each kernel: benchmarks/gpu/kernel_pmc.py for the DNA, config-4b and config-4 workloads
  timeout 900 python3 benchmarks/gpu/kernel_pmc.py --tag dna --workload dna --child-args no-accel,no-rlfm --out $O 2>&1 | tail -n 12
  timeout 900 python3 benchmarks/gpu/kernel_pmc.py --tag rep_rlfm --workload rep-rlfm --out $O 2>&1 | tail -n 8
  timeout 900 python3 benchmarks/gpu/kernel_pmc.py --tag bytes_rlfm --workload bytes-rlfm --out $O 2>&1 | tail -n 8 ;;
py)
  S=$1; shift; timeout 1500 python3 $S "$@" 2>&1 | tee $O/$(basename $S .py).txt | tail -n 40 ;;
*) echo "unknown task $T"; exit 2 ;;
esac
