"""Thread-safety promise of the ABI (include/fmx.h): a handle is immutable after build, so any
number of host threads may query it concurrently; *_dev calls honour the caller's stream."""
import ctypes as C
import os
import threading

import numpy as np
import pytest

import fm_index_amd as F
from fm_index_amd import _lib as L
from fm_index_amd import workload as W
from oracle import fm_oracle as O

pytestmark = pytest.mark.gpu


def test_concurrent_queries_on_one_handle():
    t = W.dna_text_np(1 << 18, 1)
    gi = F.FMIndexWithLocate(F.Text.with_max_character(t, 4), 2)
    oi = O.OracleIndex(t, 4, level=2)
    jobs = []
    for k in range(8):
        flat, off, _ = W.substring_patterns_np(t, 3000, 10 + k, 100 + k)
        s, e = oi.count_batch(flat, off, nthreads=4)
        ooff, opos = oi.locate_batch(s, e, nthreads=4)
        jobs.append((flat, off, s, e, opos))
    errors = []

    def worker(job):
        flat, off, s, e, opos = job
        try:
            for _ in range(5):
                b = gi.search_many(flat=flat, off=off)
                assert (b.s == s).all() and (b.e == e).all()
                _, pos = b.locate()
                assert (pos == opos).all()
        except Exception as ex:  # noqa: BLE001
            errors.append(repr(ex))

    threads = [threading.Thread(target=worker, args=(j,)) for j in jobs]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors


def test_dev_entry_points_on_a_side_stream():
    import torch
    lib = L.lib()
    dev = torch.device("cuda", 0)
    t = W.dna_text_np(1 << 18, 2)
    gi = F.FMIndexWithLocate(F.Text.with_max_character(t, 4), 1)
    oi = O.OracleIndex(t, 4, level=1)
    flat, off, _ = W.substring_patterns_np(t, 5000, 9, 3)
    s0, e0 = oi.count_batch(flat, off, nthreads=4)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        d_pat = torch.from_numpy(flat).to(dev, non_blocking=False)
        d_off = torch.from_numpy(off.astype(np.int64)).to(dev)
        npat = len(off) - 1
        d_s = torch.empty(npat, dtype=torch.int64, device=dev)
        d_e = torch.empty(npat, dtype=torch.int64, device=dev)
        d_c = torch.empty(npat, dtype=torch.int64, device=dev)
        sp = C.c_void_p(side.cuda_stream)
        h = gi.handle()
        assert lib.fmx_count_batch_dev(h, C.c_void_p(d_pat.data_ptr()), C.c_void_p(d_off.data_ptr()),
                                       npat, None, C.c_void_p(d_s.data_ptr()),
                                       C.c_void_p(d_e.data_ptr()), C.c_void_p(d_c.data_ptr()), sp) == 0
        d_o = torch.empty(npat + 1, dtype=torch.int64, device=dev)
        assert lib.fmx_offsets_dev(h, C.c_void_p(d_s.data_ptr()), C.c_void_p(d_e.data_ptr()), npat,
                                   C.c_void_p(d_o.data_ptr()), sp) == 0
        side.synchronize()
        total = int(d_o[-1].item())
        d_p = torch.empty(total, dtype=torch.int64, device=dev)
        assert lib.fmx_locate_batch_dev(h, C.c_void_p(d_s.data_ptr()), C.c_void_p(d_e.data_ptr()), npat,
                                        C.c_void_p(d_o.data_ptr()), total,
                                        C.c_void_p(d_p.data_ptr()), sp) == 0
        side.synchronize()
    assert lib.fmx_stream_status(h) == 0
    assert (d_s.cpu().numpy().view(np.uint64) == s0).all()
    assert (d_e.cpu().numpy().view(np.uint64) == e0).all()
    _, opos = oi.locate_batch(s0, e0, nthreads=4)
    assert (d_p.cpu().numpy().view(np.uint64) == opos).all()
    # ragged + refinement through the device entry points
    flat2, off2 = W.ragged_patterns_np(4000, 7, 4, 8)
    se = np.stack([s0[:4000], e0[:4000]], axis=1).reshape(-1)
    s2, e2 = oi.count_batch(flat2, off2, se)
    b = gi.search_many(flat=flat2, off=off2, s0e0=se)
    assert (b.s == s2).all() and (b.e == e2).all()


def test_count_dev_is_graph_capturable():
    """fmx_count_batch_dev launches kernels only (no allocation, no synchronisation), so a caller
    can capture it into a hipGraph and replay it (launch-bound small batches)."""
    import torch
    lib = L.lib()
    dev = torch.device("cuda", 0)
    t = W.dna_text_np(1 << 16, 5)
    gi = F.FMIndex(F.Text.with_max_character(t, 4))
    oi = O.OracleIndex(t, 4)
    flat, off, _ = W.substring_patterns_np(t, 256, 8, 3)
    s0, e0 = oi.count_batch(flat, off)
    d_pat = torch.from_numpy(flat).to(dev)
    d_off = torch.from_numpy(off.astype(np.int64)).to(dev)
    d_s = torch.zeros(256, dtype=torch.int64, device=dev)
    d_e = torch.zeros(256, dtype=torch.int64, device=dev)
    h = gi.handle()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        sp = C.c_void_p(side.cuda_stream)
        assert lib.fmx_count_batch_dev(h, C.c_void_p(d_pat.data_ptr()), C.c_void_p(d_off.data_ptr()), 256,
                                       None, C.c_void_p(d_s.data_ptr()), C.c_void_p(d_e.data_ptr()),
                                       None, sp) == 0      # warm-up outside the capture
        side.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            cap = C.c_void_p(torch.cuda.current_stream().cuda_stream)
            rc = lib.fmx_count_batch_dev(h, C.c_void_p(d_pat.data_ptr()), C.c_void_p(d_off.data_ptr()),
                                         256, None, C.c_void_p(d_s.data_ptr()),
                                         C.c_void_p(d_e.data_ptr()), None, cap)
        assert rc == 0
        d_s.zero_()
        d_e.zero_()
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
    assert (d_s.cpu().numpy().view(np.uint64) == s0).all()
    assert (d_e.cpu().numpy().view(np.uint64) == e0).all()


def _locate_ws_setup(kind="fm", n=1 << 17, npat=4096, plen=6):
    import torch
    lib = L.lib()
    dev = torch.device("cuda", 0)
    if kind == "fm":
        t = W.dna_text_np(n, 5)
        gi = F.FMIndexWithLocate(F.Text.with_max_character(t, 4), 2)
        oi = O.OracleIndex(t, 4, level=2)
    else:
        t = W.byte_text_np(n, 4)
        gi = F.RLFMIndexWithLocate(F.Text(t), 2)
        oi = O.OracleIndex(t, 255, level=2, kind="rlfm")
    flat, off, _ = W.substring_patterns_np(t, npat, plen, 3)
    s0, e0 = oi.count_batch(flat, off)
    ooff, opos = oi.locate_batch(s0, e0, nthreads=4)
    d_s = torch.from_numpy(s0.view(np.int64)).to(dev)
    d_e = torch.from_numpy(e0.view(np.int64)).to(dev)
    return lib, dev, gi, d_s, d_e, ooff, opos


@pytest.mark.parametrize("kind", ["fm", "rlfm"])
def test_locate_workspace_form_is_graph_capturable_and_stream_independent(kind):
    """fmx_offsets_ws_dev + fmx_locate_batch_ws_dev launch kernels only (scratch = the caller's workspace,
    no stream-ordered allocation): the pair can be captured into a hipGraph and replayed, two batches on two
    streams with their own workspaces give the reference's exact position sequence, and a missing / short
    workspace is FMX_ERR_ARG."""
    import torch
    lib, dev, gi, d_s, d_e, ooff, opos = _locate_ws_setup(kind)
    h, npat, total = gi.handle(), d_s.numel(), int(ooff[-1])
    wsb = int(lib.fmx_locate_workspace_bytes(h, total))
    osb = int(lib.fmx_offsets_workspace_bytes(npat))
    # (round 5: the default DNA index locates in ONE kernel that keeps its rows in LDS -- its workspace is 256 bytes;
    # every other index expands 4 bytes per hit into the workspace)
    assert (wsb == 256 if gi.walk_records() and kind == "fm" else wsb >= 4 * total) and wsb % 256 == 0 and osb % 256 == 0
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    ws = [torch.empty(wsb, dtype=torch.uint8, device=dev) for _ in range(2)]
    ows = [torch.empty(osb, dtype=torch.uint8, device=dev) for _ in range(2)]
    d_off = [torch.zeros(npat + 1, dtype=torch.int64, device=dev) for _ in range(2)]
    d_pos = [torch.zeros(total, dtype=torch.int64, device=dev) for _ in range(2)]
    torch.cuda.synchronize()

    def enqueue(i, sp):
        assert lib.fmx_offsets_ws_dev(h, C.c_void_p(d_s.data_ptr()), C.c_void_p(d_e.data_ptr()), npat,
                                      C.c_void_p(d_off[i].data_ptr()), C.c_void_p(ows[i].data_ptr()), osb, sp) == 0
        assert lib.fmx_locate_batch_ws_dev(h, C.c_void_p(d_s.data_ptr()), C.c_void_p(d_e.data_ptr()), npat,
                                           C.c_void_p(d_off[i].data_ptr()), total, C.c_void_p(d_pos[i].data_ptr()),
                                           C.c_void_p(ws[i].data_ptr()), wsb, sp) == 0
    # two streams, interleaved batches
    for r in range(6):
        i = r & 1
        enqueue(i, C.c_void_p(streams[i].cuda_stream))
    torch.cuda.synchronize()
    for i in range(2):
        assert (d_off[i].cpu().numpy().view(np.uint64) == ooff).all()
        assert (d_pos[i].cpu().numpy().view(np.uint64) == opos).all()
    # hipGraph capture + replay of the same pair of calls
    side = streams[0]
    with torch.cuda.stream(side):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            enqueue(0, C.c_void_p(torch.cuda.current_stream().cuda_stream))
        d_pos[0].zero_()
        d_off[0].zero_()
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
    assert (d_off[0].cpu().numpy().view(np.uint64) == ooff).all()
    assert (d_pos[0].cpu().numpy().view(np.uint64) == opos).all()
    assert lib.fmx_stream_status(h) == 0
    # a workspace that is missing or too small is refused before anything is launched
    sp = C.c_void_p(streams[0].cuda_stream)
    assert lib.fmx_locate_batch_ws_dev(h, C.c_void_p(d_s.data_ptr()), C.c_void_p(d_e.data_ptr()), npat,
                                       C.c_void_p(d_off[0].data_ptr()), total, C.c_void_p(d_pos[0].data_ptr()),
                                       None, wsb, sp) == L.ERR_ARG
    assert lib.fmx_locate_batch_ws_dev(h, C.c_void_p(d_s.data_ptr()), C.c_void_p(d_e.data_ptr()), npat,
                                       C.c_void_p(d_off[0].data_ptr()), total, C.c_void_p(d_pos[0].data_ptr()),
                                       C.c_void_p(ws[0].data_ptr()), wsb - 4, sp) == L.ERR_ARG
    assert lib.fmx_offsets_ws_dev(h, C.c_void_p(d_s.data_ptr()), C.c_void_p(d_e.data_ptr()), npat,
                                  C.c_void_p(d_off[0].data_ptr()), None, osb, sp) == L.ERR_ARG
    gi.close()


def test_large_host_pointer_batch_is_chunked_consistently():
    """Host-pointer batches >= 2^17 patterns are uploaded / searched / downloaded in two
    overlapping halves; the answers must equal the same patterns asked in small batches,
    with ragged (incl. empty) patterns and with refinement from given (s, e)."""
    t = W.dna_text_np(200000, 21)
    idx = F.FMIndexWithLocate(F.Text.with_max_character(t, 4), 2)
    npat = (1 << 17) + 12345
    flat, off = W.ragged_patterns_np(npat, 9, 4, 31)
    big = idx.search_many(flat=flat, off=off)
    step = 40000
    for a in range(0, npat, step):
        b = min(npat, a + step)
        sub_off = off[a:b + 1] - off[a]
        sub = idx.search_many(flat=flat[int(off[a]):int(off[b])], off=sub_off)
        assert (sub.s == big.s[a:b]).all() and (sub.e == big.e[a:b]).all()
        assert (sub.counts == big.counts[a:b]).all()
    # refinement: prepend one more symbol to every (s, e)
    se = np.stack([big.s, big.e], axis=1).reshape(-1).copy()
    one = np.full(npat, 2, dtype=np.uint8)
    off1 = np.arange(npat + 1, dtype=np.uint64)
    ref = idx.search_many(flat=one, off=off1, s0e0=se)
    j = np.arange(0, npat, 997)
    for k in j:
        pat = bytes([2]) + bytes(flat[int(off[k]):int(off[k + 1])])
        assert ref.counts[k] == idx.search(pat).count()
    # offsets that go backwards are refused, not dereferenced
    bad = off.copy()
    bad[npat // 2] = off[-1] + np.uint64(5)
    with pytest.raises(F.Error):
        idx.search_many(flat=flat, off=bad)
    # and the handle still works afterwards
    again = idx.search_many(flat=flat, off=off)
    assert (again.counts == big.counts).all()


def test_page_locked_host_arrays_take_the_copy_kernel_pipeline():
    """fmx_count_batch on page-locked caller arrays (torch pin_memory = hipHostMalloc): chunks are moved by copy
    kernels over three streams.  Same answers as the pageable call for ragged patterns, with the caller's buffers
    starting at odd bytes / odd words INSIDE their allocations (interior pointers, every alignment case of the
    copy kernels), with refinement ranges, and with outputs the caller does not want (NULL)."""
    import torch
    lib = L.lib()
    t = W.dna_text_np(300000, 23)
    idx = F.FMIndexWithLocate(F.Text.with_max_character(t, 4), 2)
    npat = (1 << 19) + 777                       # >= 2^19: eight chunks
    flat, off = W.ragged_patterns_np(npat, 11, 4, 29)
    want = idx.search_many(flat=flat, off=off)   # pageable numpy arrays
    total = int(off[-1])
    for pad_b, pad_w in ((0, 0), (3, 1), (13, 1)):
        hp = torch.zeros(total + 64, dtype=torch.uint8, pin_memory=True)
        ho = torch.zeros(npat + 1 + 4, dtype=torch.int64, pin_memory=True)
        hs = torch.zeros(npat + 4, dtype=torch.int64, pin_memory=True)
        he = torch.zeros(npat + 4, dtype=torch.int64, pin_memory=True)
        hc = torch.zeros(npat + 4, dtype=torch.int64, pin_memory=True)
        hp[pad_b:pad_b + total] = torch.from_numpy(flat)
        ho[pad_w:pad_w + npat + 1] = torch.from_numpy(off.astype(np.int64))
        rc = lib.fmx_count_batch(idx.handle(), C.c_void_p(hp.data_ptr() + pad_b), C.c_void_p(ho.data_ptr() + 8 * pad_w),
                                 npat, None, C.c_void_p(hs.data_ptr() + 8 * pad_w), C.c_void_p(he.data_ptr()),
                                 C.c_void_p(hc.data_ptr() + 8 * pad_w))
        assert rc == 0, lib.fmx_last_error()
        assert (hs[pad_w:pad_w + npat].numpy().view(np.uint64) == want.s).all()
        assert (he[:npat].numpy().view(np.uint64) == want.e).all()
        assert (hc[pad_w:pad_w + npat].numpy().view(np.uint64) == want.counts).all()
        assert int(hs[pad_w + npat]) == 0 and int(he[npat]) == 0      # nothing written behind the arrays
    # refinement from given ranges, only the counts wanted
    se = torch.zeros(2 * npat, dtype=torch.int64, pin_memory=True)
    se[:] = torch.from_numpy(np.stack([want.s, want.e], axis=1).reshape(-1).astype(np.int64))
    one = torch.full((npat,), 2, dtype=torch.uint8).pin_memory()
    off1 = torch.arange(npat + 1, dtype=torch.int64).pin_memory()
    hc = torch.zeros(npat, dtype=torch.int64, pin_memory=True)
    rc = lib.fmx_count_batch(idx.handle(), C.c_void_p(one.data_ptr()), C.c_void_p(off1.data_ptr()), npat,
                             C.c_void_p(se.data_ptr()), None, None, C.c_void_p(hc.data_ptr()))
    assert rc == 0
    ref = idx.search_many(flat=one.numpy(), off=off1.numpy().astype(np.uint64),
                          s0e0=se.numpy().view(np.uint64).copy())
    assert (hc.numpy().view(np.uint64) == ref.counts).all()
    # offsets that go backwards are refused on this path too, and an out-of-range symbol is reported
    bad = ho.clone().pin_memory()
    bad[pad_w + npat // 2] = int(off[-1]) + 5
    rc = lib.fmx_count_batch(idx.handle(), C.c_void_p(hp.data_ptr() + pad_b), C.c_void_p(bad.data_ptr() + 8 * pad_w),
                             npat, None, C.c_void_p(hs.data_ptr()), C.c_void_p(he.data_ptr()), None)
    assert rc == L.ERR_ARG
    k0 = int(np.nonzero(off[1:] > off[:-1])[0][0])           # last symbol of the first non-empty pattern: always consumed
    hp[pad_b + int(off[k0 + 1]) - 1] = 9
    rc = lib.fmx_count_batch(idx.handle(), C.c_void_p(hp.data_ptr() + pad_b), C.c_void_p(ho.data_ptr() + 8 * pad_w),
                             npat, None, C.c_void_p(hs.data_ptr()), C.c_void_p(he.data_ptr()), None)
    assert rc == L.ERR_SYMBOL_RANGE
    idx.close()


def test_concurrent_builds_and_queries_in_threads():
    """Builds are not required to run in parallel, but several host threads building and querying
    their own indexes at the same time must all get the right answers."""
    texts = [W.dna_text_np(30000 + 1111 * k, 50 + k) for k in range(4)]
    oracles = [O.OracleIndex(t, 4, level=2) for t in texts]
    errors = []

    def worker(k):
        try:
            for rep in range(3):
                t = texts[k]
                idx = F.FMIndexWithLocate(F.Text.with_max_character(t, 4), 2, kmer_table=bool(rep & 1))
                flat, off, _ = W.substring_patterns_np(t, 500, 9, 7 + rep)
                b = idx.search_many(flat=flat, off=off)
                os_, oe = oracles[k].count_batch(flat, off)
                assert (b.s == os_).all() and (b.e == oe).all()
                goff, gpos = b.locate()
                ooff, opos = oracles[k].locate_batch(b.s, b.e)
                assert (goff == ooff).all() and (gpos == opos).all()
                idx.close()
        except Exception as ex:   # noqa: BLE001
            errors.append((k, repr(ex)))

    threads = [threading.Thread(target=worker, args=(k,)) for k in range(4)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors


def test_pinned_host_batch_agrees_with_pageable():
    """Page-locked caller arrays go through the same two-chunk pipeline as pageable ones (their copies
    are real DMA): ragged patterns incl. empty ones and refinement must give what the pageable path
    gives, and a bad interior offset is still refused."""
    import ctypes as C
    import torch
    t = W.dna_text_np(300000, 23)
    idx = F.FMIndexWithLocate(F.Text.with_max_character(t, 4), 2)
    npat = (1 << 17) + 4321
    flat, off = W.ragged_patterns_np(npat, 9, 4, 37)
    want = idx.search_many(flat=flat, off=off)                       # pageable numpy arrays
    hp = torch.from_numpy(flat.copy()).pin_memory()
    ho = torch.from_numpy(off.astype(np.int64)).pin_memory()
    hs, he, hc = (torch.zeros(npat, dtype=torch.int64).pin_memory() for _ in range(3))
    lib = idx._lib
    rc = lib.fmx_count_batch(idx.handle(), C.c_void_p(hp.data_ptr()), C.c_void_p(ho.data_ptr()), npat, None,
                             C.c_void_p(hs.data_ptr()), C.c_void_p(he.data_ptr()), C.c_void_p(hc.data_ptr()))
    assert rc == 0
    assert (hs.numpy().view(np.uint64) == want.s).all() and (he.numpy().view(np.uint64) == want.e).all()
    assert (hc.numpy().view(np.uint64) == want.counts).all()
    # refinement from pinned (s, e) pairs, counts only
    se = torch.from_numpy(np.stack([want.s, want.e], axis=1).reshape(-1).astype(np.int64)).pin_memory()
    one = torch.full((npat,), 3, dtype=torch.uint8).pin_memory()
    off1 = torch.arange(npat + 1, dtype=torch.int64).pin_memory()
    rc = lib.fmx_count_batch(idx.handle(), C.c_void_p(one.data_ptr()), C.c_void_p(off1.data_ptr()), npat,
                             C.c_void_p(se.data_ptr()), None, None, C.c_void_p(hc.data_ptr()))
    assert rc == 0
    ref = idx.search_many(flat=one.numpy(), off=off1.numpy().astype(np.uint64),
                          s0e0=se.numpy().view(np.uint64))
    assert (hc.numpy().view(np.uint64) == ref.counts).all()
    # a bad interior offset in pinned memory is still refused
    ho[npat // 3] = int(off[-1]) + 99
    rc = lib.fmx_count_batch(idx.handle(), C.c_void_p(hp.data_ptr()), C.c_void_p(ho.data_ptr()), npat, None,
                             C.c_void_p(hs.data_ptr()), C.c_void_p(he.data_ptr()), C.c_void_p(hc.data_ptr()))
    assert rc == F._lib.ERR_ARG


def test_pinned_batch_with_equal_length_chunks_and_ragged_chunks():
    """page-locked batches (round 4): a chunk of equally long patterns gets its offsets from a kernel on the device
    instead of an upload; a ragged chunk still uploads them.  One batch whose first half is fixed-length and whose
    second half is ragged (so both kinds of chunk occur in one call), one that is fixed-length throughout, one with a
    constant stride but a non-zero first offset -- all against the pageable call."""
    import ctypes as C
    import torch
    t = W.dna_text_np(400000, 31)
    idx = F.FMIndex(F.Text.with_max_character(t, 4))
    lib = idx._lib
    npat = (1 << 18) + 40
    fixed_flat, fixed_off, _ = W.substring_patterns_np(t, npat // 2, 7, 3)
    rag_flat, rag_off = W.ragged_patterns_np(npat - npat // 2, 12, 4, 5)
    flat = np.concatenate([fixed_flat, rag_flat])
    off = np.concatenate([fixed_off, rag_off[1:] + fixed_off[-1]]).astype(np.uint64)
    cases = [(flat, off), (fixed_flat, fixed_off.astype(np.uint64))]
    lead = np.concatenate([np.full(5, 2, np.uint8), fixed_flat])           # offsets start at 5, stride 7
    cases.append((lead, (fixed_off + 5).astype(np.uint64)))
    for f, o in cases:
        k = len(o) - 1
        want = idx.search_many(flat=f, off=o) if int(o[0]) == 0 else None      # (the mirror wants offsets from 0)
        hp = torch.from_numpy(f.copy()).pin_memory()
        ho = torch.from_numpy(o.astype(np.int64)).pin_memory()
        hs, he, hc = (torch.zeros(k, dtype=torch.int64).pin_memory() for _ in range(3))
        rc = lib.fmx_count_batch(idx.handle(), C.c_void_p(hp.data_ptr()), C.c_void_p(ho.data_ptr()), k, None,
                                 C.c_void_p(hs.data_ptr()), C.c_void_p(he.data_ptr()), C.c_void_p(hc.data_ptr()))
        assert rc == 0, lib.fmx_last_error()
        if want is None:                                                    # same patterns as the fixed-length case
            want = idx.search_many(flat=fixed_flat, off=fixed_off.astype(np.uint64))
        assert (hs.numpy().view(np.uint64) == want.s).all() and (he.numpy().view(np.uint64) == want.e).all()
        assert (hc.numpy().view(np.uint64) == want.counts).all()
    idx.close()


def test_partly_page_locked_array_takes_the_pageable_path():
    """an array of which only a part is page-locked (hipHostRegister on a sub-range) must not send the copy
    kernels / the DMA engine into its unregistered tail: the call falls back to the pageable path and answers."""
    import ctypes as C
    import torch
    hip = C.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
    t = W.dna_text_np(200000, 9)
    idx = F.FMIndex(F.Text.with_max_character(t, 4))
    lib = idx._lib
    npat = (1 << 17)
    flat, off, _ = W.substring_patterns_np(t, npat, 8, 3)
    want = idx.search_many(flat=flat, off=off)
    page = 4096
    buf = np.zeros(len(flat) + 2 * page, dtype=np.uint8)
    start = (-buf.ctypes.data) % page                                      # a page-aligned window inside the buffer
    pat = buf[start:start + len(flat)]
    pat[:] = flat
    half = (len(flat) // 2) // page * page
    assert hip.hipHostRegister(C.c_void_p(pat.ctypes.data), C.c_size_t(half), C.c_uint(0)) == 0   # first half only
    try:
        offs = off.astype(np.uint64)
        o_s, o_e, o_c = (np.zeros(npat, dtype=np.uint64) for _ in range(3))
        rc = lib.fmx_count_batch(idx.handle(), C.c_void_p(pat.ctypes.data), offs.ctypes.data_as(C.c_void_p), npat, None,
                                 o_s.ctypes.data_as(C.c_void_p), o_e.ctypes.data_as(C.c_void_p),
                                 o_c.ctypes.data_as(C.c_void_p))
        assert rc == 0, lib.fmx_last_error()
        assert (o_s == want.s).all() and (o_e == want.e).all() and (o_c == want.counts).all()
    finally:
        hip.hipHostUnregister(C.c_void_p(pat.ctypes.data))
    idx.close()
