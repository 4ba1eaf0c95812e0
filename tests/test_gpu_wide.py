"""The WIDE engine (64-bit rows, positions and samples: the reference's `usize`, fm_index.rs:86-95, 127-140) on
texts small enough for the oracle to answer every question: FMX_FLAG_FORCE_WIDE builds the index the wide builder
makes for n >= 2^32 - 16 -- 64-bit suffix sort in two radix passes per round, record counters relative to their
superblock, 64-bit bases and samples -- with superblocks of 2^12 rows, so that a text of 10^5 symbols crosses
dozens of them.  tests/test_gpu_beyond_4g.py runs the same engine at n = 2^32 + 2^20."""
import ctypes as C

import numpy as np
import pytest

import fm_index_amd as F
from fm_index_amd import _lib as L
from fm_index_amd import workload as W
from oracle import fm_oracle as O

pytestmark = pytest.mark.gpu


def _dna(n, seed, sigma=4, dtype=np.uint8):
    t = ((W.splitmix64_np(seed, 0, n) % np.uint64(sigma)) + np.uint64(1)).astype(dtype)
    t[-1] = 0
    return t


def _ragged(npat, mmax, sigma, seed, dtype):
    flat, off = W.ragged_patterns_np(npat, mmax, min(sigma, 255), seed)
    if dtype == np.uint8:
        return flat, off
    wide = ((W.splitmix64_np(seed + 5, 0, max(len(flat), 1)) % np.uint64(sigma)) + np.uint64(1)).astype(dtype)[:len(flat)]
    return wide, off


# sigma <= 7: the one-level engine.  Larger byte alphabets: the generic wide engine -- one 4-bit level (12), levels of
# 3 + 2 bits (20), 4 + 3 (100), 4 + 4 (255); (1 << 17) + 5 rows are 33 superblocks: bases from global memory
# u16 / u32 symbols: the same generic engine with up to seven levels (sigma = 70000: 4 + 4 + 3 + 3 + 3 bits), K[] in
# global memory
@pytest.mark.parametrize("n,sigma,level,dtype", [
    (5000, 4, 2, np.uint8), (70001, 4, 2, np.uint8), ((1 << 17) + 5, 4, 3, np.uint8), (40000, 7, 1, np.uint8),
    (9000, 2, 0, np.uint8), (30000, 12, 2, np.uint8), (50001, 20, 2, np.uint8), (60001, 100, 1, np.uint8),
    (45000, 255, 2, np.uint8), ((1 << 17) + 5, 200, 3, np.uint8), (7000, 8, 0, np.uint8),
    (30000, 1000, 2, np.uint16), (9000, 4, 1, np.uint16), (40001, 70000, 1, np.uint32), (20000, 300, 2, np.uint32)])
def test_wide_engine_equals_the_oracle_on_small_texts(n, sigma, level, dtype, tmp_path):
    t = _dna(n, 100 + n % 97, sigma, dtype)
    gi = F.FMIndexWithLocate(F.Text.with_max_character(t, sigma), level, keep_sa=True, force_wide=True)
    assert gi.is_wide() and gi.len() == n and gi.level() == level
    # one-level byte indexes with max_character <= 5 at levels 1..3 sample in text order and carry walk records (round 4)
    assert gi.walk_records() == (dtype == np.uint8 and sigma <= 5 and 1 <= level <= 3)
    # generic (multi-level) indexes sample in text order too, levels 1..4: phase pieces with 64-bit superblock bases
    generic = dtype != np.uint8 or sigma > 7
    assert gi.text_order() == (gi.walk_records() or (generic and 1 <= level <= 4))
    oi = O.OracleIndex(t if dtype == np.uint8 else t.astype(np.uint32), sigma, level=level)
    assert gi.verify_sa() == 0                                    # the 64-bit suffix sort
    # backward search: ragged patterns (empty ones included), substrings, early exit
    flat, off = _ragged(3000, 14, sigma, 7, dtype)
    gb = gi.search_many(flat=flat, off=off)
    os_, oe = oi.count_batch(flat, off)
    assert (gb.s == os_).all() and (gb.e == oe).all() and (gb.counts == oe - os_).all()
    flat2, off2, _ = W.substring_patterns_np(t, 2000, 9, 3)
    gb2 = gi.search_many(flat=flat2, off=off2)
    os2, oe2 = oi.count_batch(flat2, off2)
    assert (gb2.s == os2).all() and (gb2.e == oe2).all()
    # refinement from given ranges (wrapper.rs:99-124)
    se = np.stack([os2, oe2], axis=1).reshape(-1).copy()
    one = np.full(len(os2), 2, dtype=dtype)
    off1 = np.arange(len(os2) + 1, dtype=np.uint64)
    ref = gi.search_many(flat=one, off=off1, s0e0=se)
    rs, re_ = oi.count_batch(one, off1, s0e0=se)
    assert (ref.s == rs).all() and (ref.e == re_).all()
    # locate: exact sequences, suffix-array order (wrapper.rs:203-217, 238-242)
    goff, gpos = gb2.locate()
    ooff, opos = oi.locate_batch(os2, oe2, nthreads=4)
    assert (goff == ooff).all() and (gpos == opos).all()
    goff, gpos = gi.locate_many(np.array([0], np.uint64), np.array([n], np.uint64))    # every row
    rows = np.arange(n, dtype=np.uint64)
    want = oi.get_sa(rows).astype(np.uint64)
    assert (gpos == want).all()
    # the trait methods: every (c, i) with i == n included; every row
    for c in (range(0, sigma + 1) if sigma <= 255 else [0, 1, 2, sigma // 2, sigma - 1, sigma]):
        i = np.arange(n + 1, dtype=np.uint64)
        cc = np.full(n + 1, c, dtype=np.uint64)
        assert (gi.lf_map2(cc, i) == oi.lf_map2(cc, i)).all(), c
    assert (gi.lf_map(rows) == oi.lf_map(rows)).all() and (gi.get_l(rows) == oi.get_l(rows)).all()
    assert (gi.get_sa(rows[:4000]) == want[:4000]).all()
    samp = gi.export_sa_samples()
    assert samp.dtype == np.uint64 and (samp == want[::1 << level]).all()
    assert (gi.export_bwt() == oi.get_l(rows).astype(dtype)).all()
    # the extract path: get_f / fl_map on every row, iter_chars_backward / _forward (wrapper.rs:154-183)
    assert (gi.get_f(rows) == oi.get_f(rows)).all() and (gi.fl_map(rows) == oi.fl_map(rows)).all()
    some = rows[::97][:300]
    for forward in (False, True):
        syms, lens, nxt = gi.extract_many(some, 9, forward=forward)
        i = some.copy()
        for t_ in range(9):
            want_sym = oi.get_f(i) if forward else oi.get_l(i)
            assert (syms[:, t_] == want_sym).all(), (forward, t_)
            i = (oi.fl_map(i) if forward else oi.lf_map(i)).astype(np.uint64)
        assert (lens == 9).all() and (nxt == i).all()
    # the index file: saved, loaded, same answers, same size; a damaged file is refused
    path = str(tmp_path / "wide.fmx")
    gi.save(path)
    li = type(gi).load(path)
    assert li.is_wide() and li.len() == n and li.level() == level and li.heap_size() == gi.heap_size()
    lb = li.search_many(flat=flat2, off=off2)
    assert (lb.s == os2).all() and (lb.e == oe2).all()
    _, lpos = lb.locate()
    assert (lpos == opos).all()
    assert (li.fl_map(rows[:2000]) == oi.fl_map(rows[:2000])).all()
    li.close()
    blob = bytearray(open(path, "rb").read())
    with open(path, "wb") as fh:
        fh.write(blob[:-16])
    with pytest.raises(F.Error):
        type(gi).load(path)
    gi.close()


@pytest.mark.parametrize("n,sigma,level", [(70001, 4, 2), (3000, 5, 1), ((1 << 18) + 9, 4, 3), (113, 4, 1), (5000, 1, 2)])
def test_wide_walk_records_against_row_order_and_the_oracle(n, sigma, level, tmp_path):
    """the wide engine's walk records (text-order samples; counters relative to walk superblocks of 32 records here)
    against the same engine in row order (FMX_FLAG_ROW_ORDER: the round-3 walk) and the oracle: every row through the
    batched walk (four walks per group + ring from 2^16 hits, one per group below), step counts, trait get_sa, the
    exported reference samples, save / load."""
    t = _dna(n, 7 + n % 13, sigma)
    tx = F.Text.with_max_character(t, sigma)
    gw = F.FMIndexWithLocate(tx, level, force_wide=True)
    gr = F.FMIndexWithLocate(tx, level, force_wide=True, sampling="row")
    gn = F.FMIndexWithLocate(tx, level, force_wide=True, walk_records=False)
    assert gw.is_wide() and gw.walk_records() and gw.text_order()
    assert gr.is_wide() and not gr.walk_records() and not gn.walk_records()
    oi = O.OracleIndex(t, sigma, level=level)
    rows = np.arange(n, dtype=np.uint64)
    want = oi.get_sa(rows).astype(np.uint64)
    lib = gw._lib
    for gi in (gw, gr, gn):
        lib.fmx_set_timing(gi.handle(), 1)
        _, pos = gi.locate_many(np.array([0], np.uint64), np.array([n], np.uint64))
        steps = int(lib.fmx_last_steps(gi.handle()))
        lib.fmx_set_timing(gi.handle(), 0)
        assert (pos == want).all()
        if gi is gw:
            assert steps == int((want & np.uint64((1 << level) - 1)).sum())      # a walk is SA[row] mod 2^level steps
        s = np.array([0, min(5, n), n // 2, max(n - 9, 0)], np.uint64)
        e = np.array([min(3, n), min(40, n), min(n // 2 + 70, n), n], np.uint64)
        _, pos = gi.locate_many(s, e)
        assert (pos == np.concatenate([want[int(a):int(b)] for a, b in zip(s, e)])).all()
        assert (gi.get_sa(rows[:3000]) == want[:3000]).all()
        assert (gi.export_sa_samples() == want[::1 << level]).all()
    path = str(tmp_path / "ww.fmx")
    gw.save(path)
    lw = F.FMIndexWithLocate.load(path)
    assert lw.is_wide() and lw.walk_records() and lw.heap_size() == gw.heap_size()
    _, pos = lw.locate_many(np.array([0], np.uint64), np.array([n], np.uint64))
    assert (pos == want).all()
    for gi in (gw, gr, gn, lw):
        gi.close()


def test_wide_engine_errors_and_refusals(tmp_path):
    t = _dna(30000, 5)
    gi = F.FMIndexWithLocate(F.Text.with_max_character(t, 4), 2, force_wide=True)
    lib = gi._lib
    # a pattern symbol beyond max_character: reported, not dereferenced (the reference panics on cs[c])
    with pytest.raises(F.Error):
        gi.search_many(flat=np.array([1, 2, 9], dtype=np.uint8), off=np.array([0, 3], dtype=np.uint64))
    # offsets that go backwards, ranges and rows that are not of this index
    with pytest.raises(F.Error):
        gi.search_many(flat=np.array([1, 2, 3], dtype=np.uint8), off=np.array([0, 3, 2], dtype=np.uint64))
    with pytest.raises(F.Error):
        gi.search_many(flat=np.array([1], dtype=np.uint8), off=np.array([0, 1], dtype=np.uint64),
                       s0e0=np.array([0, 30001], dtype=np.uint64))
    with pytest.raises(F.Error):
        gi.get_sa(np.array([30000], dtype=np.uint64))
    assert gi.search(bytes([1, 2])).count() >= 1                    # and the handle still works
    # what the wide engine does not have says so
    out = np.zeros(8, dtype=np.uint32)
    assert lib.fmx_export_sa_samples(gi.handle(), F._p(out)) == L.ERR_UNSUPPORTED
    gi.close()
    # eligibility: every kind (round 4: RLFM and multi-pieces too); a text of fewer than two symbols is not
    bt = W.byte_text_np(5000, 3)
    for kind in (L.KIND_RLFM, L.KIND_MULTI):
        h = C.c_void_p()
        assert lib.fmx_build(F._p(bt), len(bt), 1, 255, kind, 2, L.FLAG_FORCE_WIDE, 0, C.byref(h)) == 0
        assert lib.fmx_is_wide(h) == 1 and lib.fmx_kind(h) == kind
        lib.fmx_free(h)
    h = C.c_void_p()
    assert lib.fmx_build(F._p(bt[-1:]), 1, 1, 255, L.KIND_FM, 2, L.FLAG_FORCE_WIDE, 0, C.byref(h)) == L.ERR_UNSUPPORTED
    # a count-only wide index has no locate
    ci = F.FMIndex(F.Text.with_max_character(t, 4), force_wide=True)
    assert ci.is_wide() and ci.search(bytes([1, 2])).count() == gi_count(t, [1, 2])
    ci.close()


def gi_count(t, pat):
    m = len(pat)
    hits = np.ones(len(t) - m + 1, dtype=bool)
    for j, c in enumerate(pat):
        hits &= t[j:len(t) - m + 1 + j] == c
    return int(hits.sum())


def _runs_text(n, sigma, seed, mean_run, dtype=np.uint8):
    """symbols drawn like _dna, each repeated 1 .. 2 * mean_run - 1 times (mean_run == 1: plain random text)"""
    if mean_run <= 1:
        return _dna(n, seed, sigma, dtype)
    sym = ((W.splitmix64_np(seed, 0, n) % np.uint64(sigma)) + np.uint64(1)).astype(dtype)
    rep = (W.splitmix64_np(seed + 1, 0, n) % np.uint64(2 * mean_run - 1)).astype(np.int64) + 1
    t = np.repeat(sym, rep)[:n].copy()
    t[-1] = 0
    return t


# RLFMIndex on the wide engine (round 4): S on the generic levels (one 3-bit level for sigma <= 7), B / B' in
# superblocks of 2 records here, the run table; texts of runs (sparse vectors: stored positions) and random texts
# (dense vectors: hints + record search); the repetitive text RLFM exists for
@pytest.mark.parametrize("n,sigma,level,mean_run,dtype,run_table", [
    (5000, 4, 2, 1, np.uint8, True), (70001, 4, 2, 12, np.uint8, True), ((1 << 17) + 5, 4, 3, 3, np.uint8, True),
    (40000, 7, 1, 20, np.uint8, False), (9000, 2, 0, 1, np.uint8, True), (30000, 12, 2, 15, np.uint8, True),
    (50001, 20, 2, 1, np.uint8, False), (60001, 100, 1, 30, np.uint8, True), (45000, 255, 2, 2, np.uint8, True),
    (7000, 8, 0, 40, np.uint8, True), (30000, 1000, 2, 11, np.uint16, True), (40001, 70000, 1, 1, np.uint32, True),
    (20000, 300, 2, 25, np.uint32, False), (777, 3, 5, 200, np.uint8, True), (100000, 0, 3, 0, np.uint8, True)])
def test_wide_rlfm_equals_the_oracle_on_small_texts(n, sigma, level, mean_run, dtype, run_table, tmp_path):
    if sigma == 0:                                                  # 256-symbol block repeated with mutations
        t = W.repetitive_text_np(n, 5, base_len=256, mut_per_1024=1)
        sigma = 255
    else:
        t = _runs_text(n, sigma, 300 + n % 89, mean_run, dtype)
    sampling = "row" if n in (30000, 60001) else None               # FMX_FLAG_ROW_ORDER: the reference's rows
    gi = F.RLFMIndexWithLocate(F.Text.with_max_character(t, sigma), level, keep_sa=True, force_wide=True,
                               walk_records=run_table, sampling=sampling, run_table=run_table)
    assert gi.is_wide() and gi.len() == n and gi.level() == (level if n > (1 << level) else 0)
    # with the run table the index samples in text order (levels 1..4: phase pieces with 64-bit superblock bases)
    assert gi.walk_records() == run_table
    assert gi.text_order() == (run_table and sampling is None and 1 <= gi.level() <= 4)
    oi = O.OracleIndex(t if dtype == np.uint8 else t.astype(np.uint32), sigma, level=level, kind="rlfm")
    assert gi.verify_sa() == 0
    rows = np.arange(n, dtype=np.uint64)
    want_l = oi.get_l(rows)
    assert int(gi._lib.fmx_num_runs(gi.handle())) == 1 + int((want_l[1:] != want_l[:-1]).sum())     # rlfmi.rs:56-59
    # the trait methods: every (c, i) with i == n included; every row
    for c in (range(0, sigma + 1) if sigma <= 20 else [0, 1, 2, 3, sigma // 2, sigma - 1, sigma]):
        i = np.arange(n + 1, dtype=np.uint64)
        cc = np.full(n + 1, c, dtype=np.uint64)
        assert (gi.lf_map2(cc, i) == oi.lf_map2(cc, i)).all(), c
    assert (gi.get_l(rows) == want_l).all() and (gi.lf_map(rows) == oi.lf_map(rows)).all()
    assert (gi.export_bwt() == want_l.astype(dtype)).all()
    # backward search: ragged patterns (empty ones included), substrings, refinement
    flat, off = _ragged(3000, 14, sigma, 7, dtype)
    gb = gi.search_many(flat=flat, off=off)
    os_, oe = oi.count_batch(flat, off)
    assert (gb.s == os_).all() and (gb.e == oe).all() and (gb.counts == oe - os_).all()
    flat2, off2, _ = W.substring_patterns_np(t, 2000, 9, 3)
    gb2 = gi.search_many(flat=flat2, off=off2)
    os2, oe2 = oi.count_batch(flat2, off2)
    assert (gb2.s == os2).all() and (gb2.e == oe2).all()
    se = np.stack([os2, oe2], axis=1).reshape(-1).copy()
    one = np.full(len(os2), 2, dtype=dtype)
    off1 = np.arange(len(os2) + 1, dtype=np.uint64)
    ref = gi.search_many(flat=one, off=off1, s0e0=se)
    rs, re_ = oi.count_batch(one, off1, s0e0=se)
    assert (ref.s == rs).all() and (ref.e == re_).all()
    # locate: exact sequences in suffix-array order; every row; the trait get_sa; the exported samples
    keep = slice(0, 600)
    goff, gpos = gi.locate_many(os2[keep], oe2[keep])
    ooff, opos = oi.locate_batch(os2[keep], oe2[keep], nthreads=4)
    assert (goff == ooff).all() and (gpos == opos).all()
    want = oi.get_sa(rows).astype(np.uint64)
    _, gpos = gi.locate_many(np.array([0], np.uint64), np.array([n], np.uint64))
    assert (gpos == want).all()
    assert (gi.get_sa(rows[:4000]) == want[:4000]).all()
    samp = gi.export_sa_samples()
    assert samp.dtype == np.uint64 and (samp == want[::1 << gi.level()]).all()
    # the extract path (rlfmi.rs:145-169; wrapper.rs:154-183)
    assert (gi.get_f(rows) == oi.get_f(rows)).all() and (gi.fl_map(rows) == oi.fl_map(rows)).all()
    some = rows[::97][:300]
    for forward in (False, True):
        syms, lens, nxt = gi.extract_many(some, 9, forward=forward)
        i = some.copy()
        for t_ in range(9):
            want_sym = oi.get_f(i) if forward else oi.get_l(i)
            assert (syms[:, t_] == want_sym).all(), (forward, t_)
            i = (oi.fl_map(i) if forward else oi.lf_map(i)).astype(np.uint64)
        assert (lens == 9).all() and (nxt == i).all()
    # the 32-bit engine's RLFM index gives the same ranges and positions
    ni = F.RLFMIndexWithLocate(F.Text.with_max_character(t, sigma), level)
    nb = ni.search_many(flat=flat2, off=off2)
    assert (nb.s == os2).all() and (nb.e == oe2).all()
    ni.close()
    # the index file
    path = str(tmp_path / "wide_rlfm.fmx")
    gi.save(path)
    li = type(gi).load(path)
    assert li.is_wide() and li.len() == n and li.level() == gi.level() and li.heap_size() == gi.heap_size()
    assert li.walk_records() == run_table and int(li._lib.fmx_num_runs(li.handle())) == int(gi._lib.fmx_num_runs(gi.handle()))
    assert li.text_order() == gi.text_order()
    assert (li.export_sa_samples() == samp).all() and (li.get_sa(rows[:3000]) == want[:3000]).all()
    lb = li.search_many(flat=flat2, off=off2)
    assert (lb.s == os2).all() and (lb.e == oe2).all()
    _, lpos = li.locate_many(os2[keep], oe2[keep])
    assert (lpos == opos).all()
    assert (li.fl_map(rows[:2000]) == oi.fl_map(rows[:2000])).all()
    li.close()
    blob = bytearray(open(path, "rb").read())
    with open(path, "wb") as fh:
        fh.write(blob[:-16])
    with pytest.raises(F.Error):
        type(gi).load(path)
    gi.close()
    # count-only
    ci = F.RLFMIndex(F.Text.with_max_character(t, sigma), force_wide=True)
    cb = ci.search_many(flat=flat2, off=off2)
    assert (cb.s == os2).all() and (cb.e == oe2).all() and not ci.walk_records()
    ci.close()


# Round 5: RLFM count on the wide engine runs with an interval endpoint per lane (fmxw_g_count_ep_kernel<RLFM>).  Batches far
# larger than the grid's slots (every lane pair refills many times), batches smaller than one block, one pattern, dense
# B / B' (no stored positions: lane-wise record search) and sparse ones, one / two / three wavelet levels, a symbol outside
# the alphabet (status word, not a crash), and ranges that are not ranges of the index.
@pytest.mark.parametrize("n,sigma,mean_run,dtype", [
    (70001, 4, 12, np.uint8), (50001, 20, 1, np.uint8), (45000, 255, 2, np.uint8), (30000, 1000, 11, np.uint16),
    (40001, 70000, 1, np.uint32)])
def test_wide_rlfm_count_endpoint_per_lane(n, sigma, mean_run, dtype):
    t = _runs_text(n, sigma, 410 + n % 89, mean_run, dtype)
    gi = F.RLFMIndex(F.Text.with_max_character(t, sigma), force_wide=True)
    oi = O.OracleIndex(t if dtype == np.uint8 else t.astype(np.uint32), sigma, kind="rlfm")
    assert gi.is_wide()
    for npat, m, seed in ((300000, 6, 3), (1, 9, 4), (5, 3, 5), (33, 40, 6), (4097, 1, 7)):
        flat, off, _ = W.substring_patterns_np(t, npat, m, seed)
        gb = gi.search_many(flat=flat, off=off)
        os_, oe = oi.count_batch(flat, off)
        assert (gb.s == os_).all() and (gb.e == oe).all() and (gb.counts == oe - os_).all(), (npat, m)
    flat, off = _ragged(200000, 11, sigma, 9, dtype)                # random symbols: most patterns end early, empty ones
    gb = gi.search_many(flat=flat, off=off)
    os_, oe = oi.count_batch(flat, off)
    assert (gb.s == os_).all() and (gb.e == oe).all()
    # refinement from given ranges (wrapper.rs:105-106), the whole-index range and empty ranges among them
    k = 5000
    s0 = (W.splitmix64_np(21, 0, k) % np.uint64(n + 1)).astype(np.uint64)
    e0 = np.minimum(s0 + (W.splitmix64_np(22, 0, k) % np.uint64(2000)), np.uint64(n)).astype(np.uint64)
    s0[:3], e0[:3] = 0, n
    se = np.stack([s0, e0], axis=1).reshape(-1).copy()
    one = ((W.splitmix64_np(23, 0, k) % np.uint64(min(sigma, 200))) + np.uint64(1)).astype(dtype)
    off1 = np.arange(k + 1, dtype=np.uint64)
    ref = gi.search_many(flat=one, off=off1, s0e0=se)
    rs, re_ = oi.count_batch(one, off1, s0e0=se)
    assert (ref.s == rs).all() and (ref.e == re_).all()
    if sigma < 255 or dtype != np.uint8:
        bad = np.array([1, 1, sigma + 1], dtype=dtype)              # the search starts at the pattern's last symbol
        with pytest.raises(F.Error) as ei:
            gi.search_many(flat=bad, off=np.array([0, 3], np.uint64))
        assert ei.value.code == L.ERR_SYMBOL_RANGE
    se_bad = np.array([0, n + 5], dtype=np.uint64)
    with pytest.raises(F.Error) as ei:
        gi.search_many(flat=one[:1], off=off1[:2], s0e0=se_bad)
    assert ei.value.code == L.ERR_ARG
    flat, off, _ = W.substring_patterns_np(t, 100, 5, 8)            # the status word was cleared
    gb = gi.search_many(flat=flat, off=off)
    os_, oe = oi.count_batch(flat, off)
    assert (gb.s == os_).all() and (gb.e == oe).all()
    gi.close()


# Round 5: the generic wide kernels (byte alphabets, multi-pieces) count with an interval endpoint per lane
# (fmxw_g_count_ep_kernel) and walk a lane per hit through text-order samples (fmxw_g_walk_text_ep_kernel).  Batches far
# beyond the grid (refills / many 64-hit tickets per wave), tiny ones, every row, rows that are not rows of the index,
# one / two / three wavelet levels, u16 symbols, several pieces (end markers: multi_pieces.rs:131-153).
@pytest.mark.parametrize("n,sigma,level,dtype,pieces", [
    (60001, 255, 2, np.uint8, 0), (40000, 20, 3, np.uint8, 0), (30011, 12, 1, np.uint8, 0), (50000, 1000, 2, np.uint16, 0),
    (45000, 100, 2, np.uint8, 37), (20000, 12, 4, np.uint8, 5)])
def test_wide_generic_lane_kernels(n, sigma, level, dtype, pieces):
    t = ((W.splitmix64_np(77 + n % 31, 0, n) % np.uint64(sigma)) + np.uint64(1)).astype(dtype)
    if pieces:                                                      # tests/testutil/mod.rs:7-32: no leading zero, no double
        cut = 2 + 2 * (W.splitmix64_np(5, 0, pieces - 1) % np.uint64((n - 6) // 2)).astype(np.int64)   # zero, nonzero at n - 2
        t[cut] = 0
    t[-1] = 0
    tx = F.Text.with_max_character(t, sigma)
    if pieces:
        gi = F.FMIndexMultiPiecesWithLocate(tx, level, force_wide=True)
        oi = O.OracleIndex(t if dtype == np.uint8 else t.astype(np.uint32), sigma, level=level, kind="multi")
    else:
        gi = F.FMIndexWithLocate(tx, level, force_wide=True)
        oi = O.OracleIndex(t if dtype == np.uint8 else t.astype(np.uint32), sigma, level=level)
    assert gi.is_wide() and gi.text_order() == (1 <= gi.level() <= 4)   # generic levels (sigma > 7): phase pieces
    for npat, m, seed in ((300000, 4, 3), (1, 6, 4), (7, 2, 5), (4097, 1, 7)):
        flat, off, _ = W.substring_patterns_np(t, npat, m, seed)
        gb = gi.search_many(flat=flat, off=off)
        os_, oe = oi.count_batch(flat, off)
        assert (gb.s == os_).all() and (gb.e == oe).all() and (gb.counts == oe - os_).all(), (npat, m)
    flat, off = _ragged(150000, 9, min(sigma, 255), 9, np.uint8)
    gb = gi.search_many(flat=flat, off=off)
    os_, oe = oi.count_batch(flat, off)
    assert (gb.s == os_).all() and (gb.e == oe).all()
    rows = np.arange(n, dtype=np.uint64)
    want = oi.get_sa(rows).astype(np.uint64)
    lib = gi._lib
    lib.fmx_set_timing(gi.handle(), 1)
    _, pos = gi.locate_many(np.array([0], np.uint64), np.array([n], np.uint64))
    steps = int(lib.fmx_last_steps(gi.handle()))
    lib.fmx_set_timing(gi.handle(), 0)
    assert (pos == want).all()
    if gi.text_order():
        assert steps == int((want & np.uint64((1 << gi.level()) - 1)).sum())       # a walk is SA[row] mod 2^level steps
    # many short intervals (ragged tickets), in batch order
    k = 20000
    s = (W.splitmix64_np(31, 0, k) % np.uint64(n)).astype(np.uint64)
    e = np.minimum(s + (W.splitmix64_np(32, 0, k) % np.uint64(9)), np.uint64(n)).astype(np.uint64)
    goff, gpos = gi.locate_many(s, e)
    assert (gpos == np.concatenate([want[int(a):int(b)] for a, b in zip(s, e)])).all()
    assert (gi.get_sa(rows[::7][:5000]) == want[::7][:5000]).all()
    with pytest.raises(F.Error):
        gi.get_sa(np.array([3, n, 5], dtype=np.uint64))
    assert (gi.get_sa(rows[:100]) == want[:100]).all()
    gi.close()
