#!/bin/bash
# round 4 evidence from the final sources: the whole -m gpu suite, the default bench line, the strong-scaling G = 1
# line, the rocprofv3 summaries (kernel stats + PMC).  Output under gpurun_out/r04_final/ and gpurun_out/prof_r04/.
O=gpurun_out/r04_final; mkdir -p $O
( time python -m pytest tests -q -m gpu ) > $O/pytest_gpu.txt 2>&1; tail -6 $O/pytest_gpu.txt
( time python bench.py --steps 20 --warmup 5 ) > $O/bench_default.json 2> $O/bench_default.err; tail -c 200 $O/bench_default.err
python bench.py --gpus 1 --total-patterns 8388608 --steps 20 --warmup 5 > $O/bench_strong_g1.json 2> $O/bench_strong_g1.err
python benchmarks/gpu/walk_ab.py > $O/walk_ab.jsonl 2>/dev/null
bash profiles/run_rocprof.sh r04 > $O/run_rocprof.log 2>&1
sed -i 's#gpurun_out/r04_bench/bench_default.json#gpurun_out/r04_final/bench_default.json#' benchmarks/gpu/bench_default_r04.sh
python - <<'PY'
import re,subprocess
src=open("benchmarks/gpu/bench_default_r04.sh").read()
py=src[src.index("python - <<'PY'")+len("python - <<'PY'\n"):src.rindex("PY")]
exec(py)
PY
