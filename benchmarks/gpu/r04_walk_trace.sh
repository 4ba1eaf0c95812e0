#!/bin/bash
# kernel trace of config 3 on the walk-record index: expand + walk kernel durations
O=$PWD/gpurun_out/r04_walk; mkdir -p $O
R=$PWD
cd /tmp && export TMPDIR=/tmp
export ONLY=${ONLY:-text_walk} SHAPES=config3
rocprofv3 --kernel-trace --stats -d $O/trace --output-format csv -- python3 $R/benchmarks/gpu/walk_ab.py > $O/trace_run.txt 2>&1
f=$(ls $O/trace/*/*kernel_stats.csv | head -1); head -12 $f | cut -c1-200
python3 - <<PY
import csv,glob
f=glob.glob("$O/trace/*/*kernel_trace.csv")[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# last 12 kernels: timeline
sel=[r for r in rows if "locate_f3" in r["Kernel_Name"] or "expand" in r["Kernel_Name"]][-12:]
t0=int(sel[0]["Start_Timestamp"])
for r in sel:
    print(r["Kernel_Name"][:40], (int(r["Start_Timestamp"])-t0)/1e3, (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3, r.get("Grid_Size"), r.get("Workgroup_Size"))
PY
