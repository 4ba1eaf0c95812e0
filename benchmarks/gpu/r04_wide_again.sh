#!/bin/bash
O=gpurun_out/r04_wide; mkdir -p $O
python -m pytest tests/test_gpu_leaks.py tests/test_gpu_concurrency.py tests/test_gpu_wide.py tests/test_gpu_save_load.py -x -q -m gpu > $O/pytest_cache.txt 2>&1; tail -4 $O/pytest_cache.txt
FMX_BUILD_TRACE=1 python - <<'PY' > $O/wide_three_builds.txt 2>&1
import sys, os, time
sys.path.insert(0, os.getcwd())
import torch
import fm_index_amd as F
from fm_index_amd import workload as W
dev = torch.device("cuda", 0)
# a process that has already built a few 2^30 indexes (what bench.py is by the time of its `wide` leg)
t30 = W.dna_text_torch(1 << 30, 1, dev)
for rep in range(4):
    t0 = time.time(); ix = F.FMIndexWithLocate.from_device_text(t30.data_ptr(), 1 << 30, 4, level=2)
    sys.stderr.write("2^30 build %d: wall %.3f s build_ms %.1f\n" % (rep, time.time() - t0, ix._lib.fmx_build_ms(ix.handle()))); ix.close()
del t30; torch.cuda.empty_cache()
N = (1 << 32) + (1 << 20)
for rep in range(3):
    text = W.dna_text_torch(N, 17, dev)
    torch.cuda.synchronize()
    sys.stderr.write("== wide build %d\n" % rep); sys.stderr.flush()
    t0 = time.time()
    ix = F.FMIndexWithLocate.from_device_text(text.data_ptr(), N, 4, level=2)
    sys.stderr.write("wall %.2f s build_ms %.1f\n" % (time.time() - t0, ix._lib.fmx_build_ms(ix.handle())))
    ix.close(); del text; torch.cuda.empty_cache()
PY
grep -v "wide sort\|\[fmx build\] " $O/wide_three_builds.txt | tail -40
python benchmarks/host_pointer_rate.py 2>/dev/null | tail -3
