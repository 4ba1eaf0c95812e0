#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-pointer entry point fmx_count_batch (config 2 workload):
patterns + offsets copied in, (s, e, count) copied out, every call.  Never the headline value
(bench.py times the device-resident path); DESIGN.md section 6 quotes this number."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import numpy as np
    import torch
    import fm_index_amd as F
    from fm_index_amd import workload as W
    n, npat, m = 1 << 30, 1 << 20, 32
    dev = torch.device("cuda", 0)
    text = W.dna_text_torch(n, 1, dev)
    index = F.FMIndex.from_device_text(text.data_ptr(), n, 4)
    pat, off, _ = W.substring_patterns_torch(text, npat, m, 3)
    flat = pat.cpu().numpy()
    offs = off.cpu().numpy().astype(np.uint64)
    index.search_many(flat=flat, off=offs)
    reps = 10
    t0 = time.perf_counter()
    for _ in range(reps):
        b = index.search_many(flat=flat, off=offs)
    dt = (time.perf_counter() - t0) / reps
    assert (b.counts >= 1).all()
    # the same entry point with caller-owned, already-touched output buffers (what a host
    # program that reuses its vectors sees; the mirror above allocates fresh numpy outputs)
    import ctypes as C
    from fm_index_amd import _lib as L
    lib = L.lib()
    o_s, o_e, o_c = (np.zeros(npat, dtype=np.uint64) for _ in range(3))
    def call():
        rc = lib.fmx_count_batch(index.handle(), flat.ctypes.data_as(C.c_void_p),
                                 offs.ctypes.data_as(C.POINTER(C.c_uint64)), npat, None,
                                 o_s.ctypes.data_as(C.POINTER(C.c_uint64)),
                                 o_e.ctypes.data_as(C.POINTER(C.c_uint64)),
                                 o_c.ctypes.data_as(C.POINTER(C.c_uint64)))
        assert rc == 0
    call()
    t0 = time.perf_counter()
    for _ in range(reps):
        call()
    dt2 = (time.perf_counter() - t0) / reps
    assert (o_c == b.counts).all() and (o_s == b.s).all()
    def call_counts_only():
        rc = lib.fmx_count_batch(index.handle(), flat.ctypes.data_as(C.c_void_p),
                                 offs.ctypes.data_as(C.POINTER(C.c_uint64)), npat, None, None, None,
                                 o_c.ctypes.data_as(C.POINTER(C.c_uint64)))
        assert rc == 0
    call_counts_only()
    t0 = time.perf_counter()
    for _ in range(reps):
        call_counts_only()
    dt3 = (time.perf_counter() - t0) / reps
    # page-locked caller arrays (torch pin_memory = hipHostMalloc): the copy-kernel pipeline
    hp = torch.from_numpy(flat).pin_memory()
    ho = torch.from_numpy(offs.astype(np.int64)).pin_memory()
    hs, he, hc = (torch.zeros(npat, dtype=torch.int64).pin_memory() for _ in range(3))

    def call_pinned(counts_only=False):
        rc = lib.fmx_count_batch(index.handle(), C.c_void_p(hp.data_ptr()), C.c_void_p(ho.data_ptr()), npat, None,
                                 None if counts_only else C.c_void_p(hs.data_ptr()),
                                 None if counts_only else C.c_void_p(he.data_ptr()), C.c_void_p(hc.data_ptr()))
        assert rc == 0
    for _ in range(3):
        call_pinned()
    t0 = time.perf_counter()
    for _ in range(2 * reps):
        call_pinned()
    dt4 = (time.perf_counter() - t0) / (2 * reps)
    assert (hc.numpy().view(np.uint64) == b.counts).all() and (hs.numpy().view(np.uint64) == b.s).all()
    call_pinned(True)
    t0 = time.perf_counter()
    for _ in range(2 * reps):
        call_pinned(True)
    dt5 = (time.perf_counter() - t0) / (2 * reps)
    print(json.dumps({"entry_point": "fmx_count_batch (host pointers)",
                      "page_locked_ms_per_call": round(dt4 * 1e3, 3),
                      "page_locked_pattern_chars_per_s": round(npat * m / dt4),
                      "page_locked_ms_per_call_counts_only": round(dt5 * 1e3, 3),
                      "ms_per_call": round(dt * 1e3, 3), "pattern_chars_per_s": round(npat * m / dt),
                      "ms_per_call_reused_buffers": round(dt2 * 1e3, 3),
                      "pattern_chars_per_s_reused_buffers": round(npat * m / dt2),
                      "ms_per_call_counts_only": round(dt3 * 1e3, 3),
                      "bytes_in": int(flat.nbytes + offs.nbytes), "bytes_out": 3 * 8 * npat}))


if __name__ == "__main__":
    main()
