import ctypes as C, os, sys, time
sys.path.insert(0, os.getcwd())
import torch, numpy as np
import fm_index_amd as F
from fm_index_amd import workload as W, _lib as L
dev = torch.device("cuda", 0)
n = 1 << 30
text = W.dna_text_torch(n, 1, dev)
npat, m = 1 << 20, 32
flat, off, pos = W.substring_patterns_torch(text, npat, m, 3)
lib = L.lib()
s = torch.empty(npat, dtype=torch.int64, device=dev); e = torch.empty_like(s)
for kw in (dict(pair_index=True), dict(pair_index=True, kmer_table=True), dict(kmer_table=True), dict()):
    idx = F.FMIndex.from_device_text(text.data_ptr(), n, 4, device=0, **kw)
    def step():
        rc = lib.fmx_count_batch_dev(idx.handle(), C.c_void_p(flat.data_ptr()), C.c_void_p(off.data_ptr()), npat, None,
                                     C.c_void_p(s.data_ptr()), C.c_void_p(e.data_ptr()), None, None)
        assert rc == 0
    for _ in range(3): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20): step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 20
    lib.fmx_set_timing(idx.handle(), 1); step(); torch.cuda.synchronize()
    print(kw, "kmer_k", idx.kmer_k(), "pair", idx.has_pair_index(), "ms %.4f" % (dt * 1e3), "kernel ms %.4f" % lib.fmx_last_kernel_ms(idx.handle()), "steps", lib.fmx_last_steps(idx.handle()))
    idx.close()
