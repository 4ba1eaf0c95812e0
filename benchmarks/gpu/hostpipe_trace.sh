#!/bin/bash
# kernel timeline of one fmx_count_batch call on page-locked arrays (rocprofv3 --kernel-trace): which kernels of the
# chunk pipeline really overlap?   usage: hostpipe_trace.sh <outdir>
cd "$(dirname "$0")/../.."
OUT=${1:-gpurun_out/hostpipe_trace}
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $OUT/trace --output-format csv -- python3 benchmarks/host_pointer_rate.py > $OUT/run.txt 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys
out = sys.argv[1]
rows = []
for f in glob.glob(out + "/trace/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows = [r for r in rows if "fmx_copy" in r["Kernel_Name"] or "fmx_count" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last call: the last 8 count kernels and the copies around them
idx = [i for i, r in enumerate(rows) if "fmx_count" in r["Kernel_Name"]]
lo = idx[-8] - 4 if len(idx) >= 8 else 0
sel = rows[max(lo, 0):]
t0 = int(sel[0]["Start_Timestamp"])
with open(out + "/timeline.txt", "w") as fh:
    for r in sel:
        name = r["Kernel_Name"].split("(")[0].replace("void ", "")[:40]
        fh.write("%-42s queue %-3s grid %-8s start %9.1f us  end %9.1f us  dur %7.1f us\n" % (
            name, r.get("Queue_Id", "?"), r.get("Grid_Size", "?"), (int(r["Start_Timestamp"]) - t0) / 1e3,
            (int(r["End_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
print(open(out + "/timeline.txt").read())
PY
