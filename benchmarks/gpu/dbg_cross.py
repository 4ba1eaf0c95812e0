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
new = L.lib()
old = C.CDLL(os.path.join(os.getcwd(), "fm_index_amd", "libfmx_r01.so"))
for name, res, argt in L.SYMBOLS:
    fn = getattr(old, name); fn.restype, fn.argtypes = res, argt
s = torch.empty(npat, dtype=torch.int64, device=dev); e = torch.empty_like(s)
def bench(lib, h, tag):
    def step():
        rc = lib.fmx_count_batch_dev(h, C.c_void_p(flat.data_ptr()), C.c_void_p(off.data_ptr()), npat, None,
                                     C.c_void_p(s.data_ptr()), C.c_void_p(e.data_ptr()), None, None)
        assert rc == 0
    for _ in range(3): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20): step()
    torch.cuda.synchronize()
    print(tag, "ms %.4f" % ((time.perf_counter() - t0) / 20 * 1e3))
for builder, bname in ((new, "new"), (old, "old")):
    h = C.c_void_p()
    rc = builder.fmx_build_dev(C.c_void_p(text.data_ptr()), n, 1, 4, 0, 0xFFFFFFFF, 2 | 4, 0, C.byref(h))
    assert rc == 0
    bench(new, h, "index built by %s, queried by new" % bname)
    bench(old, h, "index built by %s, queried by old" % bname)
    builder.fmx_free(h)
