import ctypes as C, os, sys, time
sys.path.insert(0, os.getcwd())
import torch, numpy as np
import fm_index_amd as F
from fm_index_amd import workload as W, _lib as L
import bench
dev = torch.device("cuda", 0)
n = 1 << 28
text = W.dna_text_torch(n, 1, dev)
npat, m = 1 << 18, 32
flat, off, pos = W.substring_patterns_torch(text, npat, m, 3)
cl = bench.census_lib()
s = torch.empty(npat, dtype=torch.int64, device=dev); e = torch.empty_like(s)
for kw in (dict(pair_index=True), dict(pair_index=True, kmer_table=True), dict(kmer_table=True), dict()):
    idx = F.FMIndex.from_device_text(text.data_ptr(), n, 4, device=0, **kw)
    log = torch.empty(npat * m * 4, dtype=torch.int64, device=dev); cnt = torch.zeros(1, dtype=torch.int64, device=dev)
    cl.fmx_census_begin(C.c_void_p(log.data_ptr()), log.numel(), C.c_void_p(cnt.data_ptr()))
    rc = cl.fmx_count_batch_dev(idx.handle(), C.c_void_p(flat.data_ptr()), C.c_void_p(off.data_ptr()), npat, None,
                                C.c_void_p(s.data_ptr()), C.c_void_p(e.data_ptr()), None, None)
    torch.cuda.synchronize(); cl.fmx_census_end()
    print(kw, "kmer_k", idx.kmer_k(), "requested lines per pattern %.2f" % (int(cnt.item()) / npat))
    idx.close()
