#!/usr/bin/env python3
"""One-off scale check: a text of 3 * 2^30 symbols (rows and positions beyond 2^31) -- build,
count, locate, and verify every located position against the text.  ~120 GB of builder scratch.
    python benchmarks/big_text_check.py [n] [fm|rlfm]
With `rlfm` the run-length index is built (n may be anything below 2^32 - 16 since round 2) and its
(s, e) are additionally compared with an FMIndex over the same text."""
import ctypes as C
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    import fm_index_amd as F
    from fm_index_amd import _lib as L
    from fm_index_amd import workload as W
    lib = L.lib()
    dev = torch.device("cuda", 0)
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 3 << 30
    kind = sys.argv[2] if len(sys.argv) > 2 else "fm"
    text = W.dna_text_torch(n, 1, dev)
    t0 = time.time()
    cls = F.RLFMIndexWithLocate if kind == "rlfm" else F.FMIndexWithLocate
    index = cls.from_device_text(text.data_ptr(), n, 4, level=3, keep_sa=False)
    t_build = time.time() - t0
    npat, m = 1 << 18, 40
    pat, off, pos = W.substring_patterns_torch(text, npat, m, 3)
    s = torch.empty(npat, dtype=torch.int64, device=dev)
    e = torch.empty(npat, dtype=torch.int64, device=dev)
    c = torch.empty(npat, dtype=torch.int64, device=dev)
    h = index.handle()
    assert lib.fmx_count_batch_dev(h, C.c_void_p(pat.data_ptr()), C.c_void_p(off.data_ptr()), npat, None,
                                   C.c_void_p(s.data_ptr()), C.c_void_p(e.data_ptr()),
                                   C.c_void_p(c.data_ptr()), None) == 0
    torch.cuda.synchronize()
    assert bool((c >= 1).all())
    o = torch.empty(npat + 1, dtype=torch.int64, device=dev)
    lib.fmx_offsets_dev(h, C.c_void_p(s.data_ptr()), C.c_void_p(e.data_ptr()), npat, C.c_void_p(o.data_ptr()), None)
    total = int(o[-1].item())
    p = torch.empty(total, dtype=torch.int64, device=dev)
    assert lib.fmx_locate_batch_dev(h, C.c_void_p(s.data_ptr()), C.c_void_p(e.data_ptr()), npat,
                                    C.c_void_p(o.data_ptr()), total, C.c_void_p(p.data_ptr()), None) == 0
    torch.cuda.synchronize()
    hit = torch.repeat_interleave(torch.arange(npat, device=dev), c)
    ok = torch.ones(total, dtype=torch.bool, device=dev)
    for j in range(m):
        ok &= text[p + j] == pat.view(npat, m)[hit, j]
    found = torch.zeros(npat, dtype=torch.bool, device=dev)
    found[hit[p == pos[hit]]] = True
    same_as_fm = None
    if kind == "rlfm":
        index.close()
        fm = F.FMIndex.from_device_text(text.data_ptr(), n, 4)
        s2 = torch.empty_like(s)
        e2 = torch.empty_like(e)
        assert lib.fmx_count_batch_dev(fm.handle(), C.c_void_p(pat.data_ptr()), C.c_void_p(off.data_ptr()), npat,
                                       None, C.c_void_p(s2.data_ptr()), C.c_void_p(e2.data_ptr()), None, None) == 0
        torch.cuda.synchronize()
        same_as_fm = bool((s == s2).all()) and bool((e == e2).all())
        index = fm
    print(json.dumps({"n": n, "kind": kind, "rlfm_se_equal_fm_se": same_as_fm,
                      "build_s": round(t_build, 2), "index_bytes": index.heap_size(),
                      "hits": total, "max_row": int(e.max().item()), "max_pos": int(p.max().item()),
                      "all_positions_hold_pattern": bool(ok.all()),
                      "all_sources_found": bool(found.all()),
                      "rows_beyond_2^31": int((e > (1 << 31)).sum().item())}))


if __name__ == "__main__":
    main()
