#!/bin/bash
# Collects the rocprofv3 evidence for bench.py on the GPU box (run through gpurun from the
# repo root):   bash profiles/run_rocprof.sh <tag> ["extra bench flags"]
# Pass 1: --kernel-trace --stats of the bench command (per-kernel durations; bench.py's own HIP-event
#         figures of the same run land in bench_trace.json and must agree).
# Pass 2/3: --pmc TCC_EA0_RDREQ_sum + its _32B / _64B / _128B parts (FETCH_SIZE = RDREQ x 64 B on gfx950; the widths say
#         what the requests moved) and --pmc WRITE_SIZE in their own runs over `bench.py --pmc-child` (the same passes
#         bench.py runs live for roofline.traffic) -- TCC slots: MI355X_MICROARCH.md "rocprofv3 PMC slots"; counters are
#         never combined with trace domains.
TAG=${1:-r02}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# --no-accel --no-early-exit: only config-2/3/4 launches of the headline kernels in the trace
# --no-config5 / --no-rccl-check: the 8 M-pattern batch runs the same count kernel on another shape (5 ms per launch)
# 20 timed steps: the first launches of a kernel in a process run slower (cold TLB / caches: 0.70-0.77 ms against 0.63-0.65
# for the count kernel) and must not weigh on the per-kernel average the bench line is compared with
RD="TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum"
ARGS="--steps 20 --warmup 5 --no-cpu-baseline --no-pmc --no-census --no-early-exit --no-d2h --no-wide --no-config5 --no-rccl-check --no-ic-ab ${2:---no-accel}"
rocprofv3 --kernel-trace --stats -d $OUT/trace --output-format csv -- python3 $REPO/bench.py $ARGS --detail-out $OUT/bench_trace_detail.json > $OUT/bench_trace.json 2> $OUT/trace.err
rocprofv3 --pmc $RD --kernel-trace -d $OUT/pmc_fetch --output-format csv -- python3 $REPO/bench.py --pmc-child > $OUT/pmc_fetch.out 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc_write --output-format csv -- python3 $REPO/bench.py --pmc-child > $OUT/pmc_write.out 2> $OUT/pmc_write.err
# config 4b (repetitive text, RLFM): fmx_locate_rl_rounds_kernel (fmx_locate_rl_lane_kernel until round 5) -- its own trace and counter passes
rocprofv3 --kernel-trace --stats -d $OUT/trace4b --output-format csv -- python3 $REPO/bench.py --workload rep-rlfm --steps 10 --warmup 3 --no-cpu-baseline --no-pmc --no-census --no-d2h --no-accel --no-rccl-check --detail-out $OUT/bench_trace4b_detail.json > $OUT/bench_trace4b.json 2> $OUT/trace4b.err
rocprofv3 --pmc $RD --kernel-trace -d $OUT/pmc_fetch4b --output-format csv -- python3 $REPO/bench.py --pmc-child --workload rep-rlfm > $OUT/pmc_fetch4b.out 2> $OUT/pmc_fetch4b.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc_write4b --output-format csv -- python3 $REPO/bench.py --pmc-child --workload rep-rlfm > $OUT/pmc_write4b.out 2> $OUT/pmc_write4b.err
# keep only the small summaries (the full traces can be large)
cd $OUT
find trace -name "*kernel_stats*.csv" -exec cp {} $OUT/kernel_stats.csv \;
find trace4b -name "*kernel_stats*.csv" -exec cp {} $OUT/kernel_stats_config4b.csv \;
python3 - <<'PY'
import csv, glob, collections, json, os
out = {}
for tag in ("pmc_fetch", "pmc_write", "pmc_fetch4b", "pmc_write4b"):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for f in glob.glob(tag + "/**/*counter_collection*.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            kn = row.get("Kernel_Name", "?").replace("(anonymous namespace)::", "")
            name = kn.split("(")[0][:90]
            if "fmx_locate_f3u_kernel" in name:      # one kernel, two shapes (config 3 / config 3b): keyed by grid
                name += " @grid %s" % row.get("Grid_Size", "?")
            k = (name, row.get("Counter_Name", "?"))
            agg[k][0] += 1
            agg[k][1] += float(row.get("Counter_Value", 0) or 0)
    out[tag] = {"%s|%s" % k: {"dispatches": v[0], "sum": v[1], "per_dispatch": v[1] / max(v[0], 1)}
                for k, v in agg.items()}
json.dump(out, open("pmc_summary.json", "w"), indent=1)
PY
python3 - <<'PY'
import csv, glob, collections
for tag, name in (("trace", "bench"), ("trace4b", "config4b")):
    agg = collections.defaultdict(list)
    for f in glob.glob(tag + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            kn = r["Kernel_Name"]
            if "fmx_" not in kn:
                continue
            g = (kn.split("(")[0].replace("void ", "")[:70], r.get("Grid_Size") or r.get("Grid_Size_X"),
                 r.get("Workgroup_Size") or r.get("Workgroup_Size_X"))
            agg[g].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    with open("kernel_stats_by_grid_%s.csv" % name, "w") as fh:
        w = csv.writer(fh)
        w.writerow(["kernel", "grid_threads", "workgroup", "launches", "avg_ms", "min_ms"])
        for k, v in sorted(agg.items()):
            if sum(v) > 0.2:
                w.writerow([k[0], k[1], k[2], len(v), round(sum(v) / len(v), 4), round(min(v), 4)])
PY
rm -rf trace trace4b pmc_fetch pmc_write pmc_fetch4b pmc_write4b 2>/dev/null
ls -la $OUT
