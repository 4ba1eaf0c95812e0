#!/bin/bash
# PMC passes over one of the wide-engine scripts (n = 2^32 + 2^20): fabric request widths, L2 hits, wave-cycle shares of
# the fmxw_* kernels.  One counter set per pass, --kernel-trace only.
#   gpurun --timeout 900 -- 'bash benchmarks/gpu/wide_pmc.sh r05 rlfm'     wide_rlfm.py (RLFM, repetitive text)
#   gpurun --timeout 900 -- 'bash benchmarks/gpu/wide_pmc.sh r05 dna'      wide_tune.py 32 1048576 (DNA, walk records)
cd "$(dirname "$0")/../.." || exit 1
R=${1:-r05}; WHAT=${2:-rlfm}; OUT=$PWD/gpurun_out/$R/wide_${WHAT}_pmc; mkdir -p "$OUT"; export TMPDIR=/tmp
for set in "ea:TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" \
           "l2:TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_DRAM_sum" \
           "sq:SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES GRBM_GUI_ACTIVE"; do
  name=${set%%:*}; ctr=${set#*:}
  if [ "$WHAT" = dna ]; then
    timeout 300 rocprofv3 --kernel-trace --pmc $ctr -d "$OUT/$name" -o run --output-format csv -- \
      python3 benchmarks/gpu/wide_tune.py 32 1048576 > "$OUT/$name.out" 2> "$OUT/$name.err"
  else
    timeout 300 rocprofv3 --kernel-trace --pmc $ctr -d "$OUT/$name" -o run --output-format csv -- \
      python3 benchmarks/gpu/wide_rlfm.py --reps 2 --locate-patterns 16384 > "$OUT/$name.out" 2> "$OUT/$name.err"
  fi
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, json, collections
out = sys.argv[1]
res = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        kn = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if "fmxw_" in kn:
            res[kn][r["Counter_Name"]].append(float(r["Counter_Value"]))
summ = {k: {c: {"dispatches": len(v), "mean": sum(v) / len(v)} for c, v in d.items()} for k, d in res.items()}
json.dump(summ, open(out + "/summary.json", "w"), indent=1)
for k, d in summ.items():
    print(k, {c: round(x["mean"]) for c, x in d.items()})
PY
rm -rf "$OUT"/ea "$OUT"/l2 "$OUT"/sq
