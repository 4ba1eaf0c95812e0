#!/bin/bash
# small builds: the construction rows, and the API-call profile of 200 builds at n = 1000 / 10000 (FM, RLFM)
O=gpurun_out/r04_small; mkdir -p $O
python benchmarks/gpu/construction_rows.py > $O/construction_rows.jsonl 2>/dev/null; head -4 $O/construction_rows.jsonl
cd /tmp; export TMPDIR=/tmp
for cfg in "1000 fm" "1000 rlfm" "10000 rlfm"; do
  tag=$(echo $cfg | tr ' ' '_')
  rocprofv3 --hip-trace --kernel-trace --stats -d /tmp/sb_$tag -- python3 $GRAFT_REPO_ROOT/benchmarks/gpu/small_build_trace.py $cfg > $GRAFT_REPO_ROOT/$O/trace_$tag.txt 2>&1
  f=$(find /tmp/sb_$tag -name '*hip_api_stats.csv' | head -1)
  k=$(find /tmp/sb_$tag -name '*kernel_stats.csv' | head -1)
  echo "== $cfg" >> $GRAFT_REPO_ROOT/$O/api_stats.txt; head -14 $f >> $GRAFT_REPO_ROOT/$O/api_stats.txt
  echo "== $cfg" >> $GRAFT_REPO_ROOT/$O/kernel_stats.txt; head -40 $k >> $GRAFT_REPO_ROOT/$O/kernel_stats.txt
  tail -1 $GRAFT_REPO_ROOT/$O/trace_$tag.txt
done
cat $GRAFT_REPO_ROOT/$O/api_stats.txt
