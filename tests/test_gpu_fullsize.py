"""BASELINE.json full sizes (n = 2^30) through size-independent properties:
suffix-array sortedness + permutation checked on the device AND completely on the host (the linear-time
checker over all 2^30 rows, from the text alone), count >= 1 for substrings, every
located position holds its pattern, the source position is among the hits, executed steps, and
bit-identity with the CPU oracle (fed the exported BWT) on a pattern sample -- the FM path for
config 2 / 3, the RLFM path (its own S / B / B' and the rlfmi.rs formulas) for config 4, which
additionally asserts RLFM (s,e) == FM (s,e)."""
import ctypes as C

import numpy as np
import pytest

import fm_index_amd as F
from fm_index_amd import _lib as L
from fm_index_amd import workload as W

pytestmark = pytest.mark.gpu

N = 1 << 30


def _count_dev(index, pat, off, npat):
    import torch
    lib = L.lib()
    dev = pat.device
    s = torch.empty(npat, dtype=torch.int64, device=dev)
    e = torch.empty(npat, dtype=torch.int64, device=dev)
    c = torch.empty(npat, dtype=torch.int64, device=dev)
    rc = lib.fmx_count_batch_dev(index.handle(), C.c_void_p(pat.data_ptr()), C.c_void_p(off.data_ptr()),
                                 npat, None, C.c_void_p(s.data_ptr()), C.c_void_p(e.data_ptr()),
                                 C.c_void_p(c.data_ptr()), None)
    assert rc == 0
    torch.cuda.synchronize()
    assert lib.fmx_stream_status(index.handle()) == 0
    return s, e, c


def _chunks(n, step):
    return [(a, min(a + step, n)) for a in range(0, n, step)]


def _full_sa_check(index, text, level, threads=16):
    """The suffix array the GPU builder made, verified COMPLETELY on the host, without any kernel of the builder or the
    index -- the linear-time checker (Burkhardt & Karkkainen): SA is a permutation of 0..n-1; for
    every adjacent pair text[SA[i]] <= text[SA[i+1]], and where the first symbols are equal the rest decides:
    ISA[SA[i] + 1] < ISA[SA[i+1] + 1].  Together these say the array is THE suffix array (sais.rs:546-557 is the naive
    definition it must equal).  Then the L column over ALL rows against text[SA[i] - 1] (fm_index.rs:44-58), and EVERY
    exported suffix-array sample against SA[k << level] (sample.rs:33-37: with text-order sampling each of them is a
    get_sa walk through the index).  numpy over all n rows, in chunks over a thread pool (fancy indexing releases the
    GIL); ~9 GB of host memory at n = 2^30."""
    from concurrent.futures import ThreadPoolExecutor
    n = index.len()
    sa = index.export_sa()
    th = text.cpu().numpy()
    assert sa.shape == (n,) and th.shape == (n,) and sa[0] == n - 1     # the terminator suffix sorts first
    step = 1 << 24
    isa = np.full(n, 0xFFFFFFFF, dtype=np.uint32)

    def scatter(ab):
        a, b = ab
        v = sa[a:b]
        assert int(v.max()) < n
        isa[v] = np.arange(a, b, dtype=np.uint32)

    def ordered(ab):
        a, b = ab
        b = min(b, n - 1)                                                # pairs (i, i + 1), i < n - 1
        s0, s1 = sa[a:b].astype(np.int64), sa[a + 1:b + 1].astype(np.int64)
        c0, c1 = th[s0], th[s1]
        if not (c0 <= c1).all():
            return "first symbols out of order in rows %d..%d" % (a, b)
        eq = np.nonzero(c0 == c1)[0]
        # (equal first symbols: neither suffix is the terminator -- it is unique -- so SA + 1 < n)
        if not (isa[s0[eq] + 1] < isa[s1[eq] + 1]).all():
            return "suffixes with equal first symbols out of order in rows %d..%d" % (a, b)
        return None

    with ThreadPoolExecutor(threads) as ex:
        list(ex.map(scatter, _chunks(n, step)))
        assert not (isa == 0xFFFFFFFF).any(), "SA is not a permutation"    # n values < n, every slot hit: each exactly once
        bad = [m for m in ex.map(ordered, _chunks(n, step)) if m]
        assert not bad, bad[:3]
        del isa
        bwt = index.export_bwt()

        def lcol(ab):
            a, b = ab
            v = sa[a:b].astype(np.int64)
            want = np.where(v > 0, th[np.maximum(v - 1, 0)], 0)
            return bool((bwt[a:b] == want).all())
        assert all(ex.map(lcol, _chunks(n, step))), "L column differs from text[SA - 1]"
    smp = index.export_sa_samples()
    assert smp.shape == (((n - 1) >> level) + 1,) and (smp == sa[::1 << level]).all(), "samples differ from SA[k << level]"
    del sa, th, bwt, smp


def test_config2_config3_dna_1gb():
    import torch
    from oracle import fm_oracle as O
    dev = torch.device("cuda", 0)
    lib = L.lib()
    text = W.dna_text_torch(N, 1, dev)
    index = F.FMIndexWithLocate.from_device_text(text.data_ptr(), N, 4, level=2, keep_sa=True)
    assert index.len() == N and index.level() == 2
    assert index.verify_sa() == 0                      # the array IS the suffix array (device-side check)
    _full_sa_check(index, text, 2)                     # ... and says the host, from the text alone, for EVERY row
    assert index.text_order() and index.walk_records()  # the default DNA index of round 4
    npat, m = 1 << 20, 32
    pat, off, pos = W.substring_patterns_torch(text, npat, m, 3)
    s, e, c = _count_dev(index, pat, off, npat)
    assert bool((c >= 1).all())
    # config 2b: uniform random patterns -> early exit path, checked against the oracle below
    rflat, roff = W.random_patterns_np(1 << 16, 32, 4, 5)
    rb = index.search_many(flat=rflat, off=roff)
    # locate (config 3): exact positions verified against the text itself
    d_off = torch.empty(npat + 1, dtype=torch.int64, device=dev)
    assert lib.fmx_offsets_dev(index.handle(), C.c_void_p(s.data_ptr()), C.c_void_p(e.data_ptr()), npat,
                               C.c_void_p(d_off.data_ptr()), None) == 0
    total = int(d_off[-1].item())
    assert total == int(c.sum().item())
    d_pos = torch.empty(total, dtype=torch.int64, device=dev)
    assert lib.fmx_locate_batch_dev(index.handle(), C.c_void_p(s.data_ptr()), C.c_void_p(e.data_ptr()),
                                    npat, C.c_void_p(d_off.data_ptr()), total,
                                    C.c_void_p(d_pos.data_ptr()), None) == 0
    torch.cuda.synchronize()
    hit_pat = torch.repeat_interleave(torch.arange(npat, device=dev), c)
    ok = torch.ones(total, dtype=torch.bool, device=dev)
    for j in range(m):
        ok &= text[d_pos + j] == pat.view(npat, m)[hit_pat, j]
    assert bool(ok.all())
    found = torch.zeros(npat, dtype=torch.bool, device=dev)
    found[hit_pat[d_pos == pos[hit_pat]]] = True
    assert bool(found.all())
    # hits of one pattern are distinct
    key = hit_pat * N + d_pos
    assert int(torch.unique(key).numel()) == total
    # structural invariant of the whole rank structure: LF is a permutation of the rows
    marks = torch.zeros(N, dtype=torch.uint8, device=dev)
    chunk = 1 << 26
    for a in range(0, N, chunk):
        rows_ = torch.arange(a, a + chunk, dtype=torch.int64, device=dev)
        outp = torch.empty(chunk, dtype=torch.int64, device=dev)
        assert lib.fmx_lf_map_batch_dev(index.handle(), C.c_void_p(rows_.data_ptr()), chunk,
                                        C.c_void_p(outp.data_ptr()), None) == 0
        torch.cuda.synchronize()
        assert int(outp.max().item()) < N and int(outp.min().item()) >= 0
        marks.index_add_(0, outp, torch.ones(chunk, dtype=torch.uint8, device=dev))
    assert bool((marks == 1).all())
    del marks, rows_, outp
    # oracle (independent rank structure + driver) on a sample, from the exported BWT
    oi = O.OracleIndex.from_bwt(index.export_bwt(), index.export_cs(), 4,
                                samples=index.export_sa_samples(), level=2)
    k = 1 << 15
    so, eo = oi.count_batch(pat[:k * m].cpu().numpy(), np.arange(k + 1, dtype=np.uint64) * np.uint64(m),
                            nthreads=16)
    assert (so == s[:k].cpu().numpy().view(np.uint64)).all()
    assert (eo == e[:k].cpu().numpy().view(np.uint64)).all()
    ooff, opos = oi.locate_batch(so[:4096], eo[:4096], nthreads=16)
    assert (opos == d_pos[:int(ooff[-1])].cpu().numpy().view(np.uint64)).all()
    rs, re = oi.count_batch(rflat, roff, nthreads=16)
    assert (rs == rb.s).all() and (re == rb.e).all()


def test_config4_rlfm_byte_text_1gb():
    import torch
    dev = torch.device("cuda", 0)
    text = W.byte_text_torch(N, 4, dev)
    npat, m = 1 << 20, 16
    pat, off, pos = W.substring_patterns_torch(text, npat, m, 6)
    lib = L.lib()
    rl = F.RLFMIndexWithLocate.from_device_text(text.data_ptr(), N, 255, level=3, keep_sa=True)
    _full_sa_check(rl, text, 3)
    s, e, c = _count_dev(rl, pat, off, npat)
    assert bool((c >= 1).all())
    runs = int(lib.fmx_num_runs(rl.handle()))
    assert 0.99 * N < runs <= N
    # RLFM locate (rlfmi.rs:127-133, 176-189) at full size: positions verified against the text
    k = 1 << 16
    d_off = torch.empty(k + 1, dtype=torch.int64, device=dev)
    assert lib.fmx_offsets_dev(rl.handle(), C.c_void_p(s.data_ptr()), C.c_void_p(e.data_ptr()), k,
                               C.c_void_p(d_off.data_ptr()), None) == 0
    total = int(d_off[-1].item())
    d_pos = torch.empty(total, dtype=torch.int64, device=dev)
    assert lib.fmx_locate_batch_dev(rl.handle(), C.c_void_p(s.data_ptr()), C.c_void_p(e.data_ptr()), k,
                                    C.c_void_p(d_off.data_ptr()), total, C.c_void_p(d_pos.data_ptr()),
                                    None) == 0
    torch.cuda.synchronize()
    hit_pat = torch.repeat_interleave(torch.arange(k, device=dev), c[:k])
    ok = torch.ones(total, dtype=torch.bool, device=dev)
    for j in range(m):
        ok &= text[d_pos + j] == pat.view(npat, m)[hit_pat, j]
    assert bool(ok.all())
    found = torch.zeros(k, dtype=torch.bool, device=dev)
    found[hit_pat[d_pos == pos[hit_pat]]] = True
    assert bool(found.all())
    # the oracle's RLFM path (rlfmi.rs formulas over its own S / B / B' built from the exported L column)
    # at FULL size: (s, e) of a pattern sample, lf_map2 / lf_map / get_l at random rows, locate order
    from oracle import fm_oracle as O
    oi = O.OracleIndex.from_bwt(rl.export_bwt(), rl.export_cs(), 255, samples=rl.export_sa_samples(),
                                level=3, kind="rlfm")
    ks = 1 << 14
    so, eo = oi.count_batch(pat[:ks * m].cpu().numpy(), np.arange(ks + 1, dtype=np.uint64) * np.uint64(m),
                            nthreads=16)
    assert (so == s[:ks].cpu().numpy().view(np.uint64)).all()
    assert (eo == e[:ks].cpu().numpy().view(np.uint64)).all()
    rows = (W.splitmix64_np(77, 0, 4096) % np.uint64(N)).astype(np.uint64)
    syms = (W.splitmix64_np(78, 0, 4096) % np.uint64(256)).astype(np.uint64)
    assert (rl.lf_map2(syms, rows) == oi.lf_map2(syms, rows)).all()
    assert (rl.lf_map(rows) == oi.lf_map(rows)).all() and (rl.get_l(rows) == oi.get_l(rows)).all()
    ooff, opos = oi.locate_batch(so[:2048], eo[:2048], nthreads=16)
    assert (opos == d_pos[:int(ooff[-1])].cpu().numpy().view(np.uint64)).all()
    oi.close()
    rl.close()
    fm = F.FMIndex.from_device_text(text.data_ptr(), N, 255)
    s2, e2, c2 = _count_dev(fm, pat, off, npat)
    assert bool((s == s2).all()) and bool((e == e2).all())
