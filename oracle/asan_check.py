"""AddressSanitizer / UBSan pass over the CPU oracle (sanitizers run on the CPU build only):
   make -C oracle asan && LD_PRELOAD=$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so) python oracle/asan_check.py"""
import sys, ctypes, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import fm_oracle as O
import ctypes as C
O._LIB = O._bind(C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'libfm_oracle_asan.so')))
import numpy as np
from fm_index_amd import workload as W
for kind in ("fm", "rlfm", "multi"):
    for n in (2, 3, 50, 700, 5000):
        t = W.dna_text_np(n, n) if kind != "multi" else None
        if kind == "multi":
            t = (W.splitmix64_np(n, 0, n) % np.uint64(5)).astype(np.uint8)
            for i in range(n - 1):
                if t[i] == 0 and (i == 0 or t[i - 1] == 0): t[i] = 2
            if n >= 2 and t[n - 2] == 0: t[n - 2] = 3
            t[n - 1] = 0
        idx = O.OracleIndex(t, 7, level=2, kind=kind)
        flat, off = W.ragged_patterns_np(200, 6, 4, n)
        s, e = idx.count_batch(flat, off, nthreads=4)
        o, p = idx.locate_batch(s, e, nthreads=4)
        rows = np.arange(n)
        idx.get_l(rows); idx.lf_map(rows); idx.get_f(rows); idx.fl_map(rows)
        cc, ii = np.meshgrid(np.arange(8), np.arange(n + 1)); idx.lf_map2(cc.ravel(), ii.ravel())
        if kind == "multi": idx.piece_id(rows)
        idx.close()
t16 = ((W.splitmix64_np(1, 0, 3000) % np.uint64(900)) + np.uint64(1)).astype(np.uint32); t16[-1] = 0
for kind in ("fm", "rlfm"):
    idx = O.OracleIndex(t16, 1000, level=1, kind=kind)
    idx.count_batch(t16[5:9].copy(), np.array([0, 4], dtype=np.uint64)); idx.get_sa(np.arange(3000)); idx.close()
# the imports from an exported L column (the full-size GPU tests' checker), 32- and 64-bit sample entry points
t = W.repetitive_text_np(3000, 3, base_len=64, mut_per_1024=20)
for kind in ("fm", "rlfm"):
    a = O.OracleIndex(t, 255, level=2, kind=kind)
    rows = np.arange(len(t), dtype=np.uint64)
    bwt = a.get_l(rows).astype(np.uint8)
    cs = np.concatenate([[0], np.cumsum(np.bincount(t, minlength=256))[:-1]]).astype(np.uint64)
    samples = a.get_sa(rows[::4])
    b = O.OracleIndex.from_bwt(bwt, cs, 255, samples=samples, level=2, kind=kind)
    assert (b.get_sa(rows) == a.get_sa(rows)).all()
    b.close()
    h = C.c_void_p()
    s64 = np.ascontiguousarray(samples, dtype=np.uint64)
    if kind == "rlfm":
        assert O._LIB.orc_rlfm_from_bwt64(C.byref(h), O._p(bwt), len(bwt), 255, O._p(s64), 2) == 0
        O._LIB.orc_rlfm_free(h)
    else:
        assert O._LIB.orc_fm_from_bwt64(C.byref(h), O._p(bwt), len(bwt), 255, O._p(cs), O._p(s64), 2) == 0
        O._LIB.orc_fm_free(h)
    a.close()
print("asan run complete")
