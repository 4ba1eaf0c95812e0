#!/bin/bash
# instruction-issue counters of the DNA walk kernel (config 3: 2^20 hits; config 3b: 2.9e8 hits) -- is the
# kernel bound by vector-instruction issue or by memory?  Counters in their own passes, kernel-trace only.
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/issue; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export FMX_LIB=${FMX_LIB_OVERRIDE:-$REPO/fm_index_amd/libfmx.so}
ARGS="--steps 2 --warmup 1 --no-cpu-baseline --no-pmc --no-census --no-early-exit --no-d2h --no-accel --no-rlfm"
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "GRBM_GUI_ACTIVE SQ_WAVES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $set --kernel-trace -d $OUT/$tag --output-format csv -- python3 $REPO/bench.py $ARGS > $OUT/$tag.out 2> $OUT/$tag.err
done
cd $OUT
python3 - <<'PY'
import csv, glob, collections
rows = collections.defaultdict(dict)
for f in glob.glob("*/**/*counter_collection*.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r.get("Kernel_Name", "")
        if "fmx_locate" not in name and "fmx_count_f3" not in name:
            continue
        key = (name.split("(")[0][:40], r.get("Grid_Size"), r.get("Dispatch_Id"))
        rows[(f.split("/")[0], key)][r["Counter_Name"]] = float(r["Counter_Value"])
for k in sorted(rows):
    print(k, rows[k])
PY
rm -rf $OUT/*/ 2>/dev/null
