import sys, os, time
sys.path.insert(0, os.getcwd())
import torch
import fm_index_amd as F
from fm_index_amd import workload as W
dev = torch.device("cuda", 0)
N = (1 << 32) + (1 << 20)
for name, lvl in (("bytes", 3), ("dna", 2), ("bytes", 2)):
    text = W.dna_text_torch(N, 17, dev) if name == "dna" else W.byte_text_torch(N, 17, dev)
    sys.stderr.write("== %s level %d\n" % (name, lvl)); sys.stderr.flush()
    t0 = time.time()
    ix = F.FMIndexWithLocate.from_device_text(text.data_ptr(), N, 4 if name == "dna" else 255, level=lvl)
    sys.stderr.write("wall %.2f s\n" % (time.time() - t0))
    ix.close(); del text; torch.cuda.empty_cache()
