#!/bin/bash
# HIP API + kernel trace of 200 small builds (benches/construction.rs shape): which runtime calls a small build is made of
#   gpurun -- 'bash benchmarks/gpu/small_build_api_trace.sh 1000 fm'
N=${1:-1000}; K=${2:-fm}
O=/tmp/small_build_${N}_$K; rm -rf $O; mkdir -p $O $PWD/gpurun_out/r06; S=$PWD/gpurun_out/r06/small_build_${N}_$K.txt
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --hip-trace --kernel-trace --stats -d $O -o trace --output-format csv -- python3 $GRAFT_REPO_ROOT/benchmarks/gpu/small_build_trace.py $N $K 2>&1 | tail -n 3
python3 - $O <<'PY' | tee $S
import csv, glob, sys, collections
d = sys.argv[1]
for pat, title in (("*hip_api_stats*.csv", "HIP API"), ("*kernel_stats*.csv", "kernels")):
    for f in glob.glob(d + "/**/" + pat, recursive=True):
        rows = list(csv.DictReader(open(f)))
        rows.sort(key=lambda r: -float(r.get("TotalDurationNs", 0) or 0))
        print("==", title, f.split("/")[-1])
        for r in rows[:22]:
            print("  %-60s calls %7s total %10.1f us avg %8.2f us" % (r["Name"][:60], r["Calls"], float(r["TotalDurationNs"]) / 1e3, float(r["AverageNs"]) / 1e3))
PY
