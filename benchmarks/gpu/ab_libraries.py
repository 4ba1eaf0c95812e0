"""A/B of several builds of libfmx on the same index and box: python benchmarks/gpu/ab_libraries.py "" _measure ...
(tags name fm_index_amd/libfmx<tag>.so; the first one builds the indexes).  Used for the pair+table kernel
regression hunt of round 2 (profiles/r02/sweeps.md)."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from fm_index_amd import workload as W, _lib as L
dev = torch.device("cuda", 0)
n = 1 << 30
text = W.dna_text_torch(n, 1, dev)
npat, m = 1 << 20, 32
flat, off, pos = W.substring_patterns_torch(text, npat, m, 3)
L.lib()
libs = {}
for tag in sys.argv[1:]:
    l = C.CDLL(os.path.join(os.getcwd(), "fm_index_amd", "libfmx%s.so" % tag))
    for name, res, argt in L.SYMBOLS:
        fn = getattr(l, name); fn.restype, fn.argtypes = res, argt
    libs[tag] = l
s = torch.empty(npat, dtype=torch.int64, device=dev); e = torch.empty_like(s)
first = libs[sys.argv[1]]
for flags, fname in ((2 | 4, "pair+kmer"), (2, "pair"), (4, "kmer"), (0, "plain")):
    h = C.c_void_p()
    assert first.fmx_build_dev(C.c_void_p(text.data_ptr()), n, 1, 4, 0, 0xFFFFFFFF, flags, 0, C.byref(h)) == 0
    out = []
    for tag, lib in libs.items():
        def step():
            assert lib.fmx_count_batch_dev(h, C.c_void_p(flat.data_ptr()), C.c_void_p(off.data_ptr()), npat, None,
                                           C.c_void_p(s.data_ptr()), C.c_void_p(e.data_ptr()), None, None) == 0
        for _ in range(3): step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(30): step()
        torch.cuda.synchronize()
        out.append("%s %.4f" % (tag or "cur", (time.perf_counter() - t0) / 30 * 1e3))
    print(fname, " | ".join(out))
    first.fmx_free(h)
