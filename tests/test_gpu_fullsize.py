"""BASELINE.json full sizes (n = 2^30) through size-independent properties:
suffix-array sortedness + permutation checked on the device, count >= 1 for substrings, every
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


def _suffix_less(text, a, b, step=64):
    """suffix text[a:] < suffix text[b:] for arrays of start positions, on the host: windows of `step` symbols until
    the first differing symbol.  The text ends with its unique smallest symbol, so two different suffixes differ at or
    before the end of the shorter one (reads past the end are clamped onto the terminator, which decides)."""
    n = len(text)
    less = np.zeros(len(a), dtype=bool)
    undecided = np.ones(len(a), dtype=bool)
    win = np.arange(step, dtype=np.int64)
    o = 0
    while undecided.any():
        ii = np.nonzero(undecided)[0]
        ta = text[np.minimum(a[ii, None] + o + win, n - 1)]
        tb = text[np.minimum(b[ii, None] + o + win, n - 1)]
        ne = ta != tb
        has = ne.any(axis=1)
        first = ne.argmax(axis=1)
        r = np.nonzero(has)[0]
        less[ii[r]] = ta[r, first[r]] < tb[r, first[r]]
        undecided[ii[r]] = False
        o += step
        assert o < n
    return less


def _independent_sa_spot_check(index, text, level, pairs=1 << 16, seed=91):
    """The suffix array the GPU builder made, checked WITHOUT any kernel of the builder or the index:
    random adjacent pairs (SA[i], SA[i+1]) are compared on the host against the TEXT itself (numpy, first differing
    symbol), the exported L column against text[SA[i] - 1] on those rows (fm_index.rs:44-58), and EVERY exported
    suffix-array sample against SA[k << level] (sample.rs:33-37: with text-order sampling each of them is a get_sa walk
    through the index)."""
    n = index.len()
    sa = index.export_sa()
    th = text.cpu().numpy()
    assert sa.shape == (n,) and sa[0] == n - 1                  # the terminator suffix sorts first
    i = (W.splitmix64_np(seed, 0, pairs) % np.uint64(n - 1)).astype(np.int64)
    a, b = sa[i].astype(np.int64), sa[i + 1].astype(np.int64)
    assert _suffix_less(th, a, b).all(), "adjacent suffixes out of order"
    bwt = index.export_bwt()
    want = np.where(a > 0, th[np.maximum(a - 1, 0)], 0)
    assert (bwt[i] == want).all(), "L column differs from text[SA - 1]"
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
    _independent_sa_spot_check(index, text, 2)         # ... and says the host, from the text alone
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
    _independent_sa_spot_check(rl, text, 3, seed=92)
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
