#!/bin/bash
# timeline of the last page-locked fmx_count_batch call: kernels AND memory copies (rocprofv3 kernel + memory-copy trace)
O=$PWD/gpurun_out/r04_hostpipe; mkdir -p $O
R=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace -d $O/trace --output-format csv -- python3 $R/benchmarks/host_pointer_rate.py > $O/run.txt 2>&1
tail -1 $O/run.txt | cut -c1-300
python3 - "$O" <<'PY'
import csv, glob, sys
out = sys.argv[1]
ev = []
for f in glob.glob(out + "/trace/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "fmx_" in r["Kernel_Name"]:
            ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K " + r["Kernel_Name"].split("(")[0].replace("void ", "")[:36], r.get("Queue_Id", "?")))
for f in glob.glob(out + "/trace/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C %s %s B" % (r.get("Direction", "?"), r.get("Size", r.get("Bytes", "?"))), "-"))
ev.sort()
# the last full call before the counts-only calls: find the last 3 copy-kernel triples ... simply print the last 120 events
cnt = [i for i, e in enumerate(ev) if e[2].startswith("K fmx_count_f3")]
# calls have 4 count kernels each; take the window of the 12th..9th last count kernels (a full call in the middle of the run)
lo = cnt[-44] if len(cnt) >= 44 else 0
hi = cnt[-40] if len(cnt) >= 44 else len(ev) - 1
sel = [e for e in ev if ev[lo][0] - 400000 <= e[0] <= ev[hi][1] + 300000]
t0 = sel[0][0]
with open(out + "/timeline.txt", "w") as fh:
    for s, e, name, q in sel:
        fh.write("%-44s q %-3s start %9.1f us  end %9.1f us  dur %7.1f us\n" % (name, q, (s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3))
print(open(out + "/timeline.txt").read()[:6000])
PY
