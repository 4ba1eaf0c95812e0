#!/usr/bin/env python3
"""Rate of fmx_extract_batch_dev (Match::iter_chars_backward / forward for many rows): 2^20 random rows
x 32 characters on the config-2 index.  One JSON line."""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    import fm_index_amd as F
    from fm_index_amd import workload as W, _lib as L
    lib = L.lib()
    dev = torch.device("cuda", 0)
    n, nrows, k = 1 << 30, 1 << 20, 32
    text = W.dna_text_torch(n, 1, dev)
    idx = F.FMIndexWithLocate.from_device_text(text.data_ptr(), n, 4, level=2)
    rows = W.umod_torch(W.splitmix64_torch(9, 0, nrows, dev), n).to(torch.int64)
    out = torch.zeros(nrows * k, dtype=torch.uint8, device=dev)
    res = {}
    for name, fwd in (("backward", 0), ("forward", 1)):
        def run():
            assert lib.fmx_extract_batch_dev(idx.handle(), C.c_void_p(rows.data_ptr()), nrows, k, fwd,
                                             C.c_void_p(out.data_ptr()), None, None, None) == 0
        run()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(5):
            run()
        b.record()
        torch.cuda.synchronize()
        ms = a.elapsed_time(b) / 5
        res[name] = {"ms": round(ms, 3), "chars_per_s": round(nrows * k / ms * 1e3)}
        if fwd == 0:   # the characters are the text read backward from locate(row) - 1
            pos = torch.empty(nrows, dtype=torch.int64, device=dev)
            assert lib.fmx_get_sa_batch_dev(idx.handle(), C.c_void_p(rows.data_ptr()), nrows,
                                            C.c_void_p(pos.data_ptr()), None) == 0
            torch.cuda.synchronize()
            j = torch.arange(k, device=dev)[None, :]
            want = text[(pos[:, None] - 1 - j) % n]
            assert bool((out.view(nrows, k) == want).all())
    print(json.dumps({"entry_point": "fmx_extract_batch_dev", "rows": nrows, "len": k, **res}))


if __name__ == "__main__":
    main()
