#!/bin/bash
# vector-ALU busy fraction of every fmx_* query kernel of a bench run:
#   busy = SQ_INSTS_VALU x 4 cycles / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8)
#   bash benchmarks/gpu/issue_pmc_all.sh <tag> "<bench flags>"
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-dna}
OUT=$REPO/gpurun_out/issue_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 2 --warmup 1 --no-cpu-baseline --no-pmc --no-census --no-early-exit --no-d2h --no-accel $2"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE --kernel-trace -d $OUT/p --output-format csv -- python3 $REPO/bench.py $ARGS > $OUT/p.out 2> $OUT/p.err
cd $OUT
python3 - <<'PY'
import csv, glob, collections
rows = collections.defaultdict(dict)
for f in glob.glob("p/**/*counter_collection*.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r.get("Kernel_Name", "")
        if "fmx_count" not in name and "fmx_locate" not in name:
            continue
        rows[(name.split("(")[0][:60], r.get("Grid_Size"), r.get("Dispatch_Id"))][r["Counter_Name"]] = float(r["Counter_Value"])
seen = {}
for k in sorted(rows, key=lambda x: int(x[2])):
    v = rows[k]
    cyc = v.get("GRBM_GUI_ACTIVE", 0) / 8
    if cyc <= 0: continue
    busy = v.get("SQ_INSTS_VALU", 0) * 4 / 1024 / cyc
    sig = (k[0], k[1], round(v.get("SQ_INSTS_VALU", 0) / 1e6))
    if sig in seen: continue
    seen[sig] = 1
    print(k[0], "grid", k[1], "VALU %.3g SALU %.3g VMEM_RD %.3g cycles %.3g  valu_busy %.2f  valu/load %.1f" % (
        v.get("SQ_INSTS_VALU", 0), v.get("SQ_INSTS_SALU", 0), v.get("SQ_INSTS_VMEM_RD", 0), cyc, busy,
        v.get("SQ_INSTS_VALU", 0) / max(v.get("SQ_INSTS_VMEM_RD", 1), 1)))
PY
rm -rf $OUT/p
