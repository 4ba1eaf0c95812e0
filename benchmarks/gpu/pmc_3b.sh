#!/bin/bash
# HBM-side traffic of the config-3b walk launch (2.9e8 hits): FETCH_SIZE / WRITE_SIZE in their own passes,
# kernel-trace only; the 3b launch is told from the config-3 launches of the same kernel by its grid size
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmc3b; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 2 --warmup 1 --no-cpu-baseline --no-pmc --no-census --no-early-exit --no-d2h --no-accel --no-rlfm"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace -d $OUT/$c --output-format csv -- python3 $REPO/bench.py $ARGS > $OUT/$c.out 2> $OUT/$c.err
done
cd $OUT
python3 - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(list)
dur = collections.defaultdict(list)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(c + "/**/*counter_collection*.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "fmx_locate_f3p" in r.get("Kernel_Name", ""):
                agg[(r["Grid_Size"], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for f in glob.glob(c + "/**/*kernel_trace*.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "fmx_locate_f3p" in r.get("Kernel_Name", ""):
                dur[r.get("Grid_Size", r.get("Grid_Size_X", "?"))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
for k in sorted(agg):
    v = agg[k]
    print(k, "dispatches", len(v), "mean KB", sum(v) / len(v))
for k in sorted(dur):
    v = dur[k]
    print("grid", k, "launches", len(v), "mean ms", sum(v) / len(v))
PY
rm -rf $OUT/FETCH_SIZE $OUT/WRITE_SIZE
