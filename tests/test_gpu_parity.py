"""GPU parity tests: the HIP path, called through the C ABI, against the CPU oracle and the
reference's known answers.  Bit-exact: (s, e), counts, LF values, and the ORDERED locate
sequences (suffix-array order, README.md:64)."""
import numpy as np
import pytest

import fm_index_amd as F
from fm_index_amd import workload as W
from oracle import fm_oracle as O

pytestmark = pytest.mark.gpu


def b(s):
    return s.encode("latin-1")


KINDS = [("fm", F.FMIndex, F.FMIndexWithLocate)]


# ------------------------------------------------------------- known answers ----
def test_mississippi_known_answers(golden):
    g = golden["mississippi"]
    idx = F.FMIndexWithLocate(F.Text(b(g["text"])), 2)
    assert idx.len() == 12
    i, chain = 0, []
    for _ in range(12):
        i = int(idx.lf_map([i])[0])
        chain.append(i)
    assert chain == g["lf_chain_from_0"]["expected"]                 # fm_index.rs:149-160
    for ch, (s, e) in g["lf_map2_ranges"]["expected"].items():       # rlfmi.rs:285-309
        assert int(idx.lf_map2([ord(ch)], [0])[0]) == s
        assert int(idx.lf_map2([ord(ch)], [12])[0]) == e
    for pat, se in g["search_ranges"]["expected"].items():           # rlfmi.rs:312-328
        assert idx.search(b(pat)).get_range() == tuple(se)
    assert bytes(int(x) for x in idx.get_l(np.arange(12))) == b(g["bwt"]["expected"])
    assert bytes(idx.export_bwt()) == b(g["bwt"]["expected"])


def test_readme_example(golden):
    g = golden["readme"]
    index = F.FMIndexWithLocate(F.Text(b(g["text"])), g["level"])
    search = index.search(b(g["pattern"]))
    assert search.count() == g["count"]
    positions = [m.locate() for m in search.iter_matches()]
    assert positions == g["positions_in_order"]
    assert search.locate_all() == g["positions_in_order"]
    it = next(iter(search.iter_matches())).iter_chars_backward()
    prefix = bytes(reversed([next(it) for _ in range(16)]))
    assert prefix == b(g["backward_16_from_first_match"])


def test_small(golden):
    g = golden["small"]
    idx = F.FMIndexWithLocate(F.Text(b(g["text"])), g["level"])
    assert idx.level() == 0  # n=2 <= 2^2 -> level forced to 0 (sample.rs:28-31)
    s = idx.search(b(g["pattern"]))
    assert s.count() == g["count"]
    assert [m.locate() for m in s.iter_matches()] == g["positions"]


def test_invalid_texts(golden):
    for case in golden["invalid_texts"]["cases"]:
        with pytest.raises(F.Error) as ei:
            F.FMIndex(F.Text(b(case["text"])))
        assert str(ei.value) == "invalid text: " + case["message"]


def test_len(golden):
    t = F.Text(b(golden["len"]["text"]))
    assert F.FMIndex(t).len() == golden["len"]["expected"]
    assert F.FMIndexWithLocate(t, 2).len() == golden["len"]["expected"]


def test_sampling_grid_levels(golden):
    """sample.rs:96-136 through the index: effective level and number of samples."""
    for level, n in golden["sampling_grid"]["cases"] + [[4, 10]]:
        t = W.dna_text_np(n, 100 + n)
        idx = F.FMIndexWithLocate(F.Text.with_max_character(t, 4), level)
        eff = 0 if n <= (1 << level) else level
        assert idx.level() == eff
        sa = O.suffix_array(t)
        assert (idx.export_sa_samples() == sa[::1 << eff]).all()
        assert (idx.get_sa(np.arange(n)) == sa).all()


# ------------------------------------------------------------- suffix array -----
@pytest.mark.parametrize("maker,maxc", [(W.dna_text_np, 4), (W.byte_text_np, 255)])
def test_suffix_array_matches_oracle(maker, maxc):
    for n in (2, 3, 17, 256, 257, 5000, 70000):
        t = maker(n, n)
        idx = F.FMIndex(F.Text.with_max_character(t, maxc), keep_sa=True)
        assert (idx.export_sa() == O.suffix_array(t)).all()
        assert idx.verify_sa() == 0
        assert (idx.export_bwt() == O.bwt(t, O.suffix_array(t))).all()
        assert (idx.export_cs() == O.bucket_start(t, maxc)).all()


def test_suffix_array_repetitive_and_interior_zero():
    t = W.repetitive_text_np(40000, 5, base_len=512)
    idx = F.FMIndex(F.Text(t), keep_sa=True)
    assert (idx.export_sa() == O.suffix_array(t)).all()
    assert idx.verify_sa() == 0
    t2 = np.array([1] * 300 + [0], dtype=np.uint8)
    idx2 = F.FMIndex(F.Text(t2), keep_sa=True)
    assert (idx2.export_sa() == O.suffix_array(t2)).all()
    # interior zeros are accepted by FMIndexBackend::new (sais.rs:128-139 only checks the ends)
    t3 = np.array([2, 0, 0, 3, 1, 0, 2, 2, 0], dtype=np.uint8)
    idx3 = F.FMIndex(F.Text(t3), keep_sa=True)
    assert (idx3.export_sa() == O.suffix_array(t3, naive=True)).all()


# ------------------------------------------------------------- backend trait ----
@pytest.mark.parametrize("maxc,alpha", [(4, 4), (7, 7), (15, 12), (49, 2), (255, 8), (255, 255)])
def test_trait_methods_every_c_and_i(maxc, alpha):
    """lf_map2 for EVERY symbol and EVERY i in [0, n]; get_l / lf_map / get_sa for every row."""
    n = 777
    t = (W.splitmix64_np(maxc * 31 + alpha, 0, n) % np.uint64(alpha)).astype(np.uint8) + \
        (48 if maxc == 49 else 1)
    t[-1] = 0
    gi = F.FMIndexWithLocate(F.Text.with_max_character(t, maxc), 2)
    oi = O.OracleIndex(t, maxc, level=2)
    cc, ii = np.meshgrid(np.arange(maxc + 1), np.arange(n + 1))
    assert (gi.lf_map2(cc.ravel(), ii.ravel()) == oi.lf_map2(cc.ravel(), ii.ravel())).all()
    rows = np.arange(n)
    assert (gi.get_l(rows) == oi.get_l(rows)).all()
    assert (gi.lf_map(rows) == oi.lf_map(rows)).all()
    assert (gi.get_sa(rows) == oi.get_sa(rows)).all()


# ------------------------------------------------------------- property test ----
@pytest.mark.parametrize("maxc", [255, 8])
def test_property_count_locate_vs_bruteforce_and_oracle(maxc):
    """tests/test_fmindex.rs:26-89 shape: random texts (alphabet 8), level 0..3, patterns < 10
    (plus empty patterns), checked against brute force AND the oracle's exact sequences."""
    for ti in range(25):
        size = 2 + int(W.splitmix64_np(1000 + ti, 0, 1)[0] % np.uint64(1023))
        text = (W.splitmix64_np(2000 + ti, 0, size) % np.uint64(8)).astype(np.uint8) + 1
        text[-1] = 0
        level = int(W.splitmix64_np(3000 + ti, 0, 1)[0] % np.uint64(4))
        gi = F.FMIndexWithLocate(F.Text.with_max_character(text, maxc), level)
        oi = O.OracleIndex(text, maxc, level=level)
        flat, off = W.ragged_patterns_np(100, min(9, size), 7, 4000 + ti)
        gb = gi.search_many(flat=flat, off=off)
        os_, oe = oi.count_batch(flat, off)
        assert (gb.s == os_).all() and (gb.e == oe).all()
        assert (gb.counts == oe - os_).all()
        goff, gpos = gb.locate()
        ooff, opos = oi.locate_batch(os_, oe)
        assert (goff == ooff).all() and (gpos == opos).all()  # exact order
        for k in range(0, 100, 7):
            p = flat[int(off[k]):int(off[k + 1])]
            if len(p):
                exp = O.naive_search(text, p)
                assert int(gb.counts[k]) == len(exp)
                assert (np.sort(gpos[int(goff[k]):int(goff[k + 1])]) == exp).all()


# ------------------------------------------------------------- config 1 ---------
def test_config1_dna_1mb_random_patterns():
    """BASELINE config 1: n = 2^20 sigma=4, 10k uniform random length-20 patterns."""
    n = 1 << 20
    t = W.dna_text_np(n, 1)
    gi = F.FMIndexWithLocate(F.Text.with_max_character(t, 4), 2, keep_sa=True)
    assert gi.verify_sa() == 0
    oi = O.OracleIndex(t, 4, level=2)
    flat, off = W.random_patterns_np(10000, 20, 4, 2)
    gb = gi.search_many(flat=flat, off=off)
    os_, oe = oi.count_batch(flat, off, nthreads=8)
    assert (gb.s == os_).all() and (gb.e == oe).all()
    # substrings: all 20 steps execute, every count >= 1
    flat2, off2, pos2 = W.substring_patterns_np(t, 20000, 24, 3)
    gb2 = gi.search_many(flat=flat2, off=off2)
    os2, oe2 = oi.count_batch(flat2, off2, nthreads=8)
    assert (gb2.s == os2).all() and (gb2.e == oe2).all() and (gb2.counts >= 1).all()
    goff, gpos = gb2.locate()
    ooff, opos = oi.locate_batch(os2, oe2, nthreads=8)
    assert (goff == ooff).all() and (gpos == opos).all()
    # short patterns: wide [s, e) ranges (config 3b shape)
    flat3, off3, _ = W.substring_patterns_np(t, 512, 5, 4)
    gb3 = gi.search_many(flat=flat3, off=off3)
    os3, oe3 = oi.count_batch(flat3, off3)
    assert (gb3.s == os3).all() and (gb3.e == oe3).all()
    goff3, gpos3 = gb3.locate()
    ooff3, opos3 = oi.locate_batch(os3, oe3, nthreads=8)
    assert int(goff3[-1]) > 100000
    assert (gpos3 == opos3).all()


def test_byte_text_two_level_path():
    """sigma=255 text, Text::new => L=8 => two 4-bit levels."""
    n = 300000
    t = W.byte_text_np(n, 4)
    gi = F.FMIndexWithLocate(F.Text(t), 3)
    oi = O.OracleIndex(t, 255, level=3)
    flat, off, _ = W.substring_patterns_np(t, 5000, 3, 6)
    flat_r, off_r = W.random_patterns_np(5000, 4, 255, 7)
    for fl, of in ((flat, off), (flat_r, off_r)):
        gb = gi.search_many(flat=fl, off=of)
        os_, oe = oi.count_batch(fl, of, nthreads=8)
        assert (gb.s == os_).all() and (gb.e == oe).all()
        goff, gpos = gb.locate()
        ooff, opos = oi.locate_batch(os_, oe, nthreads=8)
        assert (goff == ooff).all() and (gpos == opos).all()


def test_binary_bench_text_L6():
    """benches/common.rs:5-15 shape: '0'/'1' bytes with max_character b'1' => L = 6."""
    n = 50001
    r = W.splitmix64_np(11, 0, n)
    t = np.where((r & np.uint64(1)) == 0, ord("0"), ord("1")).astype(np.uint8)
    t[-1] = 0
    gi = F.FMIndexWithLocate(F.Text.with_max_character(t, ord("1")), 2)
    oi = O.OracleIndex(t, ord("1"), level=2)
    pats = [format(k, "08b").encode() for k in range(256)]  # benches/common.rs:18-27
    gb = gi.search_many(pats)
    flat, off = F.pack_patterns(pats)
    os_, oe = oi.count_batch(flat, off)
    assert (gb.s == os_).all() and (gb.e == oe).all()
    assert int(gb.counts.sum()) == 50000 - 7  # every position matches exactly one pattern
    goff, gpos = gb.locate()
    ooff, opos = oi.locate_batch(os_, oe, nthreads=8)
    assert (gpos == opos).all()


# ------------------------------------------------------------- API behaviour ----
def test_refinement_and_empty_pattern():
    t = W.dna_text_np(5000, 3)
    gi = F.FMIndex(F.Text.with_max_character(t, 4))
    a = gi.search(bytes([2, 3])).search(bytes([1]))          # wrapper.rs:99-124
    assert a.get_range() == gi.search(bytes([1, 2, 3])).get_range()
    assert gi.search(b"").count() == 5000                    # empty pattern: (0, n)
    # a pattern that cannot occur: count 0, s == e as in the reference
    oi = O.OracleIndex(t, 4)
    p = bytes([1, 0, 1])
    assert gi.search(p).get_range() == oi.search(p)


def test_symbol_out_of_range_is_an_error_not_a_crash():
    t = W.dna_text_np(1000, 3)
    gi = F.FMIndex(F.Text.with_max_character(t, 4))
    with pytest.raises(F.Error) as ei:
        gi.search(bytes([1, 9, 1]))
    assert ei.value.code == F._lib.ERR_SYMBOL_RANGE
    assert gi.search(bytes([1, 2])).count() > 0  # status was cleared
    with pytest.raises(F.Error) as ei:
        F.FMIndex(F.Text.with_max_character(np.array([1, 9, 0], dtype=np.uint8), 4))
    assert ei.value.code == F._lib.ERR_SYMBOL_RANGE


def test_count_only_index_has_no_locate():
    t = W.dna_text_np(1000, 3)
    gi = F.FMIndex(F.Text.with_max_character(t, 4))
    with pytest.raises(F.Error) as ei:
        gi.locate_many([0], [4])
    assert ei.value.code == F._lib.ERR_NO_LOCATE


def test_tiny_texts():
    for raw in (b"\x00", b"a", b"ab\x00", b"aa\x00"):
        t = np.frombuffer(raw, dtype=np.uint8)
        gi = F.FMIndexWithLocate(F.Text(t), 1)
        oi = O.OracleIndex(t, 255, level=1)
        for p in (b"a", b"b", b"ab", b"aa", b""):
            assert gi.search(p).get_range() == oi.search(p), (raw, p)
            assert gi.search(p).locate_all() == oi.locate(p)


def test_empty_text():
    """sais.rs:121-123: a 0-length text builds; every search is the empty interval."""
    for cls in (F.FMIndex, lambda t: F.FMIndexWithLocate(t, 2)):
        gi = cls(F.Text(b""))
        assert gi.len() == 0
        for p in (b"a", b"", b"ab"):
            s = gi.search(p)
            assert s.get_range() == (0, 0) and s.count() == 0


def test_ranges_that_are_not_of_this_index_are_refused():
    """(s, e) handed to the refinement / locate entry points must be rows of this index; anything
    else is reported (FMX_ERR_ARG) instead of being dereferenced."""
    t = W.dna_text_np(5000, 2)
    idx = F.FMIndexWithLocate(F.Text.with_max_character(t, 4), 2)
    n = idx.len()
    flat, off = F.pack_patterns([bytes([1, 2]), bytes([3])])
    with pytest.raises(F.Error) as ei:
        idx.search_many(flat=flat, off=off, s0e0=np.array([0, n, 0, n + 7], dtype=np.uint64))
    assert ei.value.code == F._lib.ERR_ARG
    with pytest.raises(F.Error) as ei:
        idx.locate_many(np.array([10, n - 3], dtype=np.uint64), np.array([12, n + 100], dtype=np.uint64))
    assert ei.value.code == F._lib.ERR_ARG
    # the handle keeps working
    b = idx.search_many(flat=flat, off=off)
    assert b.counts[0] >= 0 and (b.locate()[1] < n).all()


def test_locate_offsets_with_gaps_are_reported_not_walked():
    t = W.dna_text_np(5000, 2)
    idx = F.FMIndexWithLocate(F.Text.with_max_character(t, 4), 2)
    lib = idx._lib
    s = np.array([10, 100, 200], dtype=np.uint64)
    e = np.array([12, 103, 201], dtype=np.uint64)
    off = np.array([4, 9, 20, 30], dtype=np.uint64)          # gaps before, between and behind
    pos = np.full(30, 2**64 - 1, dtype=np.uint64)
    rc = lib.fmx_locate_batch(idx.handle(), F._p(s), F._p(e), 3, F._p(off), F._p(pos))
    assert rc == F._lib.ERR_ARG
    # the ranges themselves were still located where the offsets put them
    oi = O.OracleIndex(t, 4, level=2)
    assert (pos[4:6] == oi.get_sa(np.array([10, 11], dtype=np.uint64))).all()
    assert (pos[9:12] == oi.get_sa(np.array([100, 101, 102], dtype=np.uint64))).all()
    assert (pos < 5000).all()                                  # gap slots hold a valid position (row 0's)


def test_multi_level_fm_locate_large_batch():
    """FM over two wavelet levels: batches of >= 2^22 hits take the walk-per-lane kernel
    (fmx_locate_ep_kernel<FM>), smaller ones the group-per-walk kernel; same ordered positions as the
    oracle either way."""
    n = 1 << 16
    t = W.byte_text_np(n, 4)
    gi = F.FMIndexWithLocate(F.Text(t), 2)
    oi = O.OracleIndex(t, 255, level=2)
    for npat in (300, 20000):                      # ~7.7e4 and ~5.1e6 hits
        flat, off, _ = W.substring_patterns_np(t, npat, 1, 31)
        gb = gi.search_many(flat=flat, off=off)
        os_, oe = oi.count_batch(flat, off)
        assert (gb.s == os_).all() and (gb.e == oe).all()
        total = int((oe - os_).sum())
        assert (total >= (1 << 22)) == (npat == 20000), total
        goff, gpos = gb.locate()
        ooff, opos = oi.locate_batch(os_, oe, nthreads=8)
        assert (goff == ooff).all() and (gpos == opos).all(), npat
