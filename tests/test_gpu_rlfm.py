"""GPU parity tests for RLFMIndex / RLFMIndexWithLocate (src/rlfmi.rs) through the C ABI."""
import numpy as np
import pytest

import fm_index_amd as F
from fm_index_amd import workload as W
from oracle import fm_oracle as O

pytestmark = pytest.mark.gpu


def b(s):
    return s.encode("latin-1")


def test_mississippi_known_answers_rlfm(golden):
    g = golden["mississippi"]
    idx = F.RLFMIndexWithLocate(F.Text(b(g["text"])), 2)
    assert idx.len() == 12
    i, chain = 0, []
    for _ in range(12):
        i = int(idx.lf_map([i])[0])
        chain.append(i)
    assert chain == g["lf_chain_from_0"]["expected"]                 # rlfmi.rs:271-282
    for ch, (s, e) in g["lf_map2_ranges"]["expected"].items():       # rlfmi.rs:285-309
        assert int(idx.lf_map2([ord(ch)], [0])[0]) == s
        assert int(idx.lf_map2([ord(ch)], [12])[0]) == e
    for pat, se in g["search_ranges"]["expected"].items():           # rlfmi.rs:312-328
        assert idx.search(b(pat)).get_range() == tuple(se)
    assert bytes(int(x) for x in idx.get_l(np.arange(12))) == b(g["bwt"]["expected"])  # :259-268
    assert int(idx._lib.fmx_num_runs(idx.handle())) == len(g["rlfm_S"]["expected"])    # :197-206


def test_readme_and_small_rlfm(golden):
    g = golden["readme"]
    index = F.RLFMIndexWithLocate(F.Text(b(g["text"])), g["level"])
    search = index.search(b(g["pattern"]))
    assert search.count() == g["count"]
    assert [m.locate() for m in search.iter_matches()] == g["positions_in_order"]
    assert search.locate_all() == g["positions_in_order"]
    g = golden["small"]
    idx = F.RLFMIndexWithLocate(F.Text(b(g["text"])), g["level"])
    s = idx.search(b(g["pattern"]))
    assert s.count() == g["count"] and s.locate_all() == g["positions"]


def test_invalid_texts_rlfm(golden):
    for case in golden["invalid_texts"]["cases"]:
        with pytest.raises(F.Error) as ei:
            F.RLFMIndex(F.Text(b(case["text"])))
        assert str(ei.value) == "invalid text: " + case["message"]


@pytest.mark.parametrize("maxc,alpha,n", [(4, 4, 777), (255, 3, 2000), (255, 255, 1555), (49, 2, 3000)])
def test_rlfm_trait_methods_every_c_and_i(maxc, alpha, n):
    t = (W.splitmix64_np(maxc * 7 + alpha, 0, n) % np.uint64(alpha)).astype(np.uint8) + \
        (48 if maxc == 49 else 1)
    t[-1] = 0
    gi = F.RLFMIndexWithLocate(F.Text.with_max_character(t, maxc), 2)
    oi = O.OracleIndex(t, maxc, level=2, kind="rlfm")
    cc, ii = np.meshgrid(np.arange(maxc + 1), np.arange(n + 1))
    assert (gi.lf_map2(cc.ravel(), ii.ravel()) == oi.lf_map2(cc.ravel(), ii.ravel())).all()
    rows = np.arange(n)
    assert (gi.get_l(rows) == oi.get_l(rows)).all()
    assert (gi.lf_map(rows) == oi.lf_map(rows)).all()
    assert (gi.get_sa(rows) == oi.get_sa(rows)).all()


def test_rlfm_property_vs_bruteforce_and_oracle():
    """tests/test_rlfmindex.rs:26-89 shape."""
    for ti in range(20):
        size = 2 + int(W.splitmix64_np(1500 + ti, 0, 1)[0] % np.uint64(1023))
        text = (W.splitmix64_np(2500 + ti, 0, size) % np.uint64(8)).astype(np.uint8) + 1
        text[-1] = 0
        level = int(W.splitmix64_np(3500 + ti, 0, 1)[0] % np.uint64(4))
        gi = F.RLFMIndexWithLocate(F.Text(text), level)
        oi = O.OracleIndex(text, 255, level=level, kind="rlfm")
        flat, off = W.ragged_patterns_np(100, min(9, size), 7, 4500 + ti)
        gb = gi.search_many(flat=flat, off=off)
        os_, oe = oi.count_batch(flat, off)
        assert (gb.s == os_).all() and (gb.e == oe).all()
        goff, gpos = gb.locate()
        ooff, opos = oi.locate_batch(os_, oe)
        assert (goff == ooff).all() and (gpos == opos).all()
        for k in range(0, 100, 9):
            p = flat[int(off[k]):int(off[k + 1])]
            if len(p):
                assert int(gb.counts[k]) == len(O.naive_search(text, p))


def test_rlfm_repetitive_text_long_runs():
    """the case RLFM exists for (lib.rs:45-47): few, long runs -> sparse B / B', select hints
    and the binary search between hints are exercised."""
    n = 200000
    t = W.repetitive_text_np(n, 5, base_len=256, mut_per_1024=1)
    gi = F.RLFMIndexWithLocate(F.Text(t), 3)
    oi = O.OracleIndex(t, 255, level=3, kind="rlfm")
    fm = F.FMIndex(F.Text(t))
    runs = int(gi._lib.fmx_num_runs(gi.handle()))
    assert runs < n // 4
    flat, off, _ = W.substring_patterns_np(t, 4000, 12, 8)
    gb = gi.search_many(flat=flat, off=off)
    os_, oe = oi.count_batch(flat, off, nthreads=8)
    assert (gb.s == os_).all() and (gb.e == oe).all()
    fb = fm.search_many(flat=flat, off=off)   # identical to FMIndex counts
    assert (fb.s == gb.s).all() and (fb.e == gb.e).all()
    rows = np.arange(0, n, 37)
    assert (gi.lf_map(rows) == oi.lf_map(rows)).all()
    sub = slice(0, 300)
    goff, gpos = gi.locate_many(gb.s[sub], gb.e[sub])
    ooff, opos = oi.locate_batch(os_[sub], oe[sub], nthreads=8)
    assert (goff == ooff).all() and (gpos == opos).all()
    # all-equal text: a single long run per symbol
    t2 = np.array([3] * 5000 + [0], dtype=np.uint8)
    g2 = F.RLFMIndexWithLocate(F.Text(t2), 2)
    o2 = O.OracleIndex(t2, 255, level=2, kind="rlfm")
    cc, ii = np.meshgrid(np.array([0, 3, 4]), np.arange(5002))
    assert (g2.lf_map2(cc.ravel(), ii.ravel()) == o2.lf_map2(cc.ravel(), ii.ravel())).all()
    assert g2.search(bytes([3] * 10)).locate_all() == o2.locate(bytes([3] * 10))


def test_config4_shape_byte_text_rlfm():
    """BASELINE config 4 shape at a size the oracle handles: sigma=255 text, len-16 substrings."""
    n = 1 << 19
    t = W.byte_text_np(n, 4)
    gi = F.RLFMIndex(F.Text(t))
    oi = O.OracleIndex(t, 255, kind="rlfm")
    flat, off, _ = W.substring_patterns_np(t, 20000, 16, 6)
    gb = gi.search_many(flat=flat, off=off)
    os_, oe = oi.count_batch(flat, off, nthreads=8)
    assert (gb.s == os_).all() and (gb.e == oe).all() and (gb.counts >= 1).all()
    flat_r, off_r = W.random_patterns_np(5000, 16, 255, 9)
    gb = gi.search_many(flat=flat_r, off=off_r)
    os_, oe = oi.count_batch(flat_r, off_r, nthreads=8)
    assert (gb.s == os_).all() and (gb.e == oe).all()


@pytest.mark.parametrize("runlen", [20, 300, 1500])      # (5000 spent 100 s in the oracle's suffix sort)
def test_rlfm_long_runs_use_stored_positions(tmp_path, runlen):
    """B / B' with fewer than one 1 per 32 bits keep the positions of their ones (select = one
    load); every trait method, count, locate and the saved file must still match the oracle."""
    n = 120000
    base = ((W.splitmix64_np(runlen, 0, n // runlen + 2) % np.uint64(5)) + np.uint64(1)).astype(np.uint8)
    t = np.repeat(base, runlen)[:n].copy()
    t[-1] = 0
    gi = F.RLFMIndexWithLocate(F.Text.with_max_character(t, 5), 2)
    oi = O.OracleIndex(t, 5, kind="rlfm", level=2)
    rows = np.arange(0, n, 7, dtype=np.uint64)
    assert (gi.get_l(rows) == oi.get_l(rows)).all()
    assert (gi.lf_map(rows) == oi.lf_map(rows)).all()
    assert (gi.get_f(rows) == oi.get_f(rows)).all()
    assert (gi.fl_map(rows) == oi.fl_map(rows)).all()
    assert (gi.get_sa(rows[:3000]) == oi.get_sa(rows[:3000])).all()
    cs = (W.splitmix64_np(3, 0, 4000) % np.uint64(6)).astype(np.uint64)
    ii = (W.splitmix64_np(4, 0, 4000) % np.uint64(n + 1)).astype(np.uint64)
    assert (gi.lf_map2(cs, ii) == oi.lf_map2(cs, ii)).all()
    flat, off = W.ragged_patterns_np(1500, 8, 5, 11)
    flat2, off2, _ = W.substring_patterns_np(t, 1500, runlen // 10 + 3, 12)
    for fl, of in ((flat, off), (flat2, off2)):
        b = gi.search_many(flat=fl, off=of)
        os_, oe = oi.count_batch(fl, of)
        assert (b.s == os_).all() and (b.e == oe).all()
    few = gi.search_many(flat=flat2[: 20 * (runlen // 10 + 3)], off=off2[:21])
    goff, gpos = few.locate()
    ooff, opos = oi.locate_batch(few.s, few.e)
    assert (goff == ooff).all() and (gpos == opos).all()
    gi.save(tmp_path / "r.fmx")
    g2 = F.RLFMIndexWithLocate.load(tmp_path / "r.fmx")
    assert g2.heap_size() == gi.heap_size()
    assert (g2.lf_map(rows) == oi.lf_map(rows)).all() and (g2.fl_map(rows) == oi.fl_map(rows)).all()


@pytest.mark.parametrize("gen", ["sigma2", "sigma4", "sigma255", "rep400", "rep150", "rep100", "rep60", "rep25", "rep5"])
def test_every_run_density_gets_a_one_load_select(gen):
    """B / B' with any density of ones: select blocks of 64 / 32 / 16 / 8 ones or stored positions
    (fmx_internal.h).  Count, locate and the trait calls must equal the oracle in every band, and
    every band runs the endpoint-per-lane kernels."""
    n = 1 << 17
    if gen.startswith("sigma"):
        sig = int(gen[5:])
        t = (W.splitmix64_np(91, 0, n) % np.uint64(sig)).astype(np.uint8) + 1
        t[-1] = 0
    else:
        t = W.repetitive_text_np(n, 5, base_len=1 << 9, mut_per_1024=int(gen[3:]))
    gi = F.RLFMIndexWithLocate(F.Text(t), 2)
    oi = O.OracleIndex(t, 255, level=2, kind="rlfm")
    runs = int(gi._lib.fmx_num_runs(gi.handle()))
    flat, off, _ = W.substring_patterns_np(t, 3000, 6, 13)
    rflat, roff = W.ragged_patterns_np(500, 7, int(t.max()), 17)
    for f, o in ((flat, off), (rflat, roff)):
        gb = gi.search_many(flat=f, off=o)
        os_, oe = oi.count_batch(f, o)
        assert (gb.s == os_).all() and (gb.e == oe).all(), (gen, runs / n)
        goff, gpos = gb.locate()
        ooff, opos = oi.locate_batch(os_, oe)
        assert (goff == ooff).all() and (gpos == opos).all(), (gen, runs / n)
    rows = (W.splitmix64_np(19, 0, 2000) % np.uint64(n + 1)).astype(np.uint64)
    syms = (W.splitmix64_np(21, 0, 2000) % np.uint64(int(t.max()) + 1)).astype(np.uint64)
    assert (gi.lf_map2(syms, rows) == oi.lf_map2(syms, rows)).all()
    r2 = rows[rows < n]
    assert (gi.lf_map(r2) == oi.lf_map(r2)).all() and (gi.get_sa(r2) == oi.get_sa(r2)).all()
    assert (gi.fl_map(r2) == oi.fl_map(r2)).all()


@pytest.mark.parametrize("gen,level,sampling", [("sigma255", 1, None), ("sigma255", 3, None), ("sigma4", 2, "row"), ("rep60", 3, None),
                                                ("rep5", 1, "row"), ("sigma255", 4, None), ("rep5", 0, None)])
def test_run_table_walk_equals_the_wavelet_walk_and_the_oracle(gen, level, sampling, tmp_path):
    """RLFM indexes that locate carry lf_map of every run's first row (round 4, FmxDev::lfrun): an LF step of the batched
    walk (>= 2^18 hits: fmx_locate_ep_kernel<..., LFR>) is the B piece of the row + one table entry.  Same ordered
    positions as the walk through S and B' (FMX_FLAG_NO_WALK_RECORDS) and as the oracle -- text and row order, levels
    0..4, select blocks and stored positions, runs that start before their B piece -- and through save / load."""
    n = 1 << 17
    if gen.startswith("sigma"):
        sig = int(gen[5:])
        t = (W.splitmix64_np(94, 0, n) % np.uint64(sig)).astype(np.uint8) + 1
        t[-1] = 0
    else:
        t = W.repetitive_text_np(n, 5, base_len=1 << 9, mut_per_1024=int(gen[3:]))
    # (FMX_FLAG_RUN_TABLE: the random texts have about one run per row, where the builder leaves the table out by itself)
    with_t = F.RLFMIndexWithLocate(F.Text(t), level, sampling=sampling, run_table=True)
    without = F.RLFMIndexWithLocate(F.Text(t), level, sampling=sampling, walk_records=False)
    assert with_t.walk_records() and not without.walk_records()
    runs = int(with_t._lib.fmx_num_runs(with_t.handle()))
    assert with_t.heap_size() - without.heap_size() == 4 * runs
    oi = O.OracleIndex(t, 255, level=level, kind="rlfm")
    want = oi.get_sa(np.arange(n)).astype(np.uint64)
    s = np.array([0, 0, n // 3], np.uint64)                  # 2 n + ... hits: the one-walk-per-lane kernel
    e = np.array([n, n, n], np.uint64)
    expect = np.concatenate([want, want, want[n // 3:]])
    path = str(tmp_path / "rl.fmx")
    with_t.save(path)
    loaded = F.RLFMIndexWithLocate.load(path)
    assert loaded.walk_records() and loaded.heap_size() == with_t.heap_size()
    for gi in (with_t, without, loaded):
        _, pos = gi.locate_many(s, e)
        assert (pos == expect).all(), gen
        gi.close()


@pytest.mark.parametrize("gen", ["sigma4", "sigma255", "rep60", "rep5"])
def test_large_hit_batches_take_the_walk_per_lane_kernel(gen):
    """locate batches of >= 2^18 hits run fmx_locate_ep_kernel (one walk per lane, LDS hit queue);
    smaller ones the group-per-walk kernel.  Both must give the oracle's ordered positions -- here a
    batch of 3-6 x 10^5 hits, for select blocks of several sizes and for stored positions."""
    n = 1 << 17
    if gen.startswith("sigma"):
        sig = int(gen[5:])
        t = (W.splitmix64_np(93, 0, n) % np.uint64(sig)).astype(np.uint8) + 1
        t[-1] = 0
        m = 2 if sig == 4 else 1
    else:
        t = W.repetitive_text_np(n, 5, base_len=1 << 9, mut_per_1024=int(gen[3:]))
        m = 6
    gi = F.RLFMIndexWithLocate(F.Text(t), 2)
    oi = O.OracleIndex(t, 255, level=2, kind="rlfm")
    flat, off, _ = W.substring_patterns_np(t, 4096 if gen != "sigma4" else 64, m, 29)
    gb = gi.search_many(flat=flat, off=off)
    os_, oe = oi.count_batch(flat, off)
    assert (gb.s == os_).all() and (gb.e == oe).all()
    total = int((oe - os_).sum())
    assert total >= (1 << 18), (gen, total)
    goff, gpos = gb.locate()
    ooff, opos = oi.locate_batch(os_, oe, nthreads=8)
    assert (goff == ooff).all() and (gpos == opos).all(), gen


@pytest.mark.parametrize("sampling,level", [(None, 3), ("row", 2), (None, 1), ("row", 0), (None, 4), ("row", 6)])
def test_long_intervals_take_the_lane_per_walk_kernel(sampling, level):
    """batches that average 64+ hits per pattern (and 2^18+ hits in all) on an index with the run table go through
    fmx_locate_rl_rounds_kernel -- a lane per walk on consecutive hits (round 4), since round 6 in rounds of one LF step
    over four tickets per wave -- in text order and in row order: the exact
    sequences of the oracle, and of the same index without the run table (the endpoint-per-lane / group kernels)"""
    n = 300000
    t = W.repetitive_text_np(n, 11, base_len=512, mut_per_1024=4)
    gi = F.RLFMIndexWithLocate(F.Text(t), level, sampling=sampling)   # repetitive: r <= n / 4, the builder adds the table
    assert gi.walk_records() and gi.text_order() == (sampling is None and level >= 1)
    oi = O.OracleIndex(t, 255, level=level, kind="rlfm")
    flat, off, _ = W.substring_patterns_np(t, 600, 3, 21)            # short patterns: hundreds of hits each
    gb = gi.search_many(flat=flat, off=off)
    os_, oe = oi.count_batch(flat, off, nthreads=8)
    assert (gb.s == os_).all() and (gb.e == oe).all()
    total = int((oe - os_).sum())
    assert total >= (1 << 18) and total // 600 >= 64
    goff, gpos = gb.locate()
    ooff, opos = oi.locate_batch(os_, oe, nthreads=8)
    assert (goff == ooff).all() and (gpos == opos).all()
    plain = F.RLFMIndexWithLocate(F.Text(t), level, sampling=sampling, walk_records=False)
    _, ppos = plain.search_many(flat=flat, off=off).locate()
    assert (ppos == opos).all()
    gi.close(); plain.close()


def test_run_table_space_policy():
    """round 5: an index type that exists to save space gets the 4-bytes-per-run table by itself only
    when the text is repetitive (r <= n / 4); a text with about one run per row gets it with FMX_FLAG_RUN_TABLE only,
    FMX_FLAG_NO_WALK_RECORDS wins over both, and a count-only index never carries it.  Positions are the same."""
    n = 1 << 16
    rnd = (W.splitmix64_np(95, 0, n) % np.uint64(255)).astype(np.uint8) + 1
    rnd[-1] = 0
    rep = W.repetitive_text_np(n, 7, base_len=1 << 9, mut_per_1024=5)
    a = F.RLFMIndexWithLocate(F.Text(rnd), 2)
    b = F.RLFMIndexWithLocate(F.Text(rnd), 2, run_table=True)
    c = F.RLFMIndexWithLocate(F.Text(rep), 2)
    d = F.RLFMIndexWithLocate(F.Text(rep), 2, run_table=True, walk_records=False)
    e = F.RLFMIndex(F.Text(rep))
    runs = lambda g: int(g._lib.fmx_num_runs(g.handle()))     # noqa: E731
    assert runs(a) * 4 > n and not a.walk_records() and b.walk_records()
    assert b.heap_size() - a.heap_size() == 4 * runs(a)
    assert runs(c) * 4 <= n and c.walk_records() and not d.walk_records() and not e.walk_records()
    assert c.heap_size() - d.heap_size() == 4 * runs(c)
    s, en = np.array([0], np.uint64), np.array([n], np.uint64)
    assert (a.locate_many(s, en)[1] == b.locate_many(s, en)[1]).all()
    assert (c.locate_many(s, en)[1] == d.locate_many(s, en)[1]).all()
    for g in (a, b, c, d, e):
        g.close()
