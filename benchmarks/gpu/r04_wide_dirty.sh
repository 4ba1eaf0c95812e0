#!/bin/bash
# does device memory that ANOTHER process used and freed make the next process's large hipMallocs slow?  (the wide
# builder's 137-172 GB of scratch: build_ms 0.6 s on a fresh box, 3-8 s after a test session on the same box)
O=gpurun_out/r04_wide; mkdir -p $O
python - <<'PY' > $O/dirty.txt 2>&1
import torch, time
t0=time.time()
xs=[torch.empty(32<<30, dtype=torch.uint8, device="cuda").fill_(1) for _ in range(7)]
torch.cuda.synchronize(); print("dirtied 224 GiB in %.1f s" % (time.time()-t0))
PY
cat $O/dirty.txt
FMX_BUILD_TRACE=1 python - <<'PY' > $O/wide_after_dirty.txt 2>&1
import sys, os, time
sys.path.insert(0, os.getcwd())
import torch
import fm_index_amd as F
from fm_index_amd import workload as W
dev = torch.device("cuda", 0)
N = (1 << 32) + (1 << 20)
for rep in range(2):
    text = W.dna_text_torch(N, 17, dev)
    torch.cuda.synchronize()
    sys.stderr.write("== build %d\n" % rep); sys.stderr.flush()
    t0 = time.time()
    ix = F.FMIndexWithLocate.from_device_text(text.data_ptr(), N, 4, level=2)
    sys.stderr.write("wall %.2f s build_ms %.1f\n" % (time.time() - t0, ix._lib.fmx_build_ms(ix.handle())))
    ix.close(); del text; torch.cuda.empty_cache()
PY
cat $O/wide_after_dirty.txt
