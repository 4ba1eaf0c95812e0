"""Opt-in 2-step (pair) index (FMX_FLAG_PAIR_INDEX): bit-identical (s, e) to the 1-step path
and to the oracle, including the equal pair left by the reference's early exit."""
import numpy as np
import pytest

import fm_index_amd as F
from fm_index_amd import workload as W
from oracle import fm_oracle as O

pytestmark = pytest.mark.gpu


def _check(text, maxc, flat, off, s0e0=None):
    gp = F.FMIndex(F.Text.with_max_character(text, maxc), pair_index=True)
    g1 = F.FMIndex(F.Text.with_max_character(text, maxc))
    oi = O.OracleIndex(text, maxc)
    bp = gp.search_many(flat=flat, off=off, s0e0=s0e0)
    b1 = g1.search_many(flat=flat, off=off, s0e0=s0e0)
    os_, oe = oi.count_batch(flat, off, s0e0)
    assert (bp.s == os_).all() and (bp.e == oe).all()
    assert (b1.s == os_).all() and (b1.e == oe).all()
    return gp


@pytest.mark.parametrize("n", [4, 5, 9, 130, 1000, 70001])
def test_pair_index_random_and_substring_patterns(n):
    t = W.dna_text_np(n, 40 + n)
    flat, off = W.ragged_patterns_np(3000, 13, 4, 50 + n)      # lengths 0..13: odd and even
    gp = _check(t, 4, flat, off)
    assert gp.has_pair_index()
    if n > 40:
        flat2, off2, _ = W.substring_patterns_np(t, 2000, 17, 60 + n)
        _check(t, 4, flat2, off2)
        flat3, off3, _ = W.substring_patterns_np(t, 2000, 2, 61 + n)   # one pair step, wide ranges
        _check(t, 4, flat3, off3)


def test_pair_index_every_short_pattern_including_zero_symbol():
    """all patterns of length 1..3 over {0..4}: code-0 rows, symbol 0 inside a pattern."""
    t = W.dna_text_np(3001, 7)
    pats = []
    for m in (1, 2, 3):
        for v in range(5 ** m):
            pats.append(bytes((v // 5 ** q) % 5 for q in range(m)))
    flat, off = F.pack_patterns(pats)
    _check(t, 4, flat, off)


def test_pair_index_all_equal_text_and_small_alphabets():
    t = np.array([1] * 999 + [0], dtype=np.uint8)           # every 2-gram is code 0
    flat, off = F.pack_patterns([bytes([1] * m) for m in range(0, 40)] + [bytes([1, 2]), bytes([2, 1])])
    _check(t, 4, flat, off)
    t2 = (W.splitmix64_np(5, 0, 5000) % np.uint64(2)).astype(np.uint8) + 1
    t2[-1] = 0
    flat2, off2 = W.ragged_patterns_np(2000, 12, 2, 9)
    gp = _check(t2, 2, flat2, off2)                          # max_character 2 (L = 2)
    assert gp.has_pair_index()


def test_pair_index_refinement_and_out_of_range():
    t = W.dna_text_np(20000, 3)
    flat, off, _ = W.substring_patterns_np(t, 500, 6, 4)
    g1 = F.FMIndex(F.Text.with_max_character(t, 4))
    first = g1.search_many(flat=flat, off=off)
    se = np.stack([first.s, first.e], axis=1).reshape(-1)
    flat2, off2 = W.ragged_patterns_np(500, 7, 4, 5)
    _check(t, 4, flat2, off2, s0e0=se)                       # wrapper.rs:99-124 from (s, e)
    gp = F.FMIndex(F.Text.with_max_character(t, 4), pair_index=True)
    with pytest.raises(F.Error) as ei:
        gp.search(bytes([1, 2, 7, 1]))
    assert ei.value.code == F._lib.ERR_SYMBOL_RANGE


def test_pair_index_not_applicable_falls_back():
    t = W.byte_text_np(5000, 4)
    g = F.FMIndex(F.Text(t), pair_index=True)                # sigma = 255: ignored
    assert not g.has_pair_index()
    t2 = np.array([2, 1, 0, 3, 1, 2, 0], dtype=np.uint8)     # interior zero: ignored
    g2 = F.FMIndex(F.Text.with_max_character(t2, 4), pair_index=True)
    assert not g2.has_pair_index()
    oi = O.OracleIndex(t2, 4)
    for p in (bytes([1, 2]), bytes([3, 1, 2]), bytes([1])):
        assert g2.search(p).get_range() == oi.search(p)


def test_the_default_index_gets_both_accelerators_where_they_pay():
    """Round 6: a DNA-like FM index of >= 2^24 symbols gets the pair index and the k-mer start table by DEFAULT when the
    device has room (FMX_FLAG_AUTO asked for it in rounds 4-5 and is still accepted); FMX_FLAG_PLAIN vetoes; a shorter
    text, a larger alphabet or another kind stays plain.  Same (s, e) either way -- the oracle's."""
    n = 1 << 24
    t = W.dna_text_np(n, 3)
    tx = F.Text.with_max_character(t, 4)
    ga = F.FMIndexWithLocate(tx, 2)
    gb = F.FMIndex(tx, auto=True)
    g1 = F.FMIndex(tx, plain=True)
    assert ga.has_pair_index() and ga.kmer_k() >= 8 and gb.has_pair_index() and gb.kmer_k() == ga.kmer_k()
    assert not g1.has_pair_index() and g1.kmer_k() == 0
    flat, off, _ = W.substring_patterns_np(t, 20000, 21, 5)
    rflat, roff = W.ragged_patterns_np(20000, 40, 4, 6)
    for f, o in ((flat, off), (rflat, roff)):
        ba, bb, b1 = ga.search_many(flat=f, off=o), gb.search_many(flat=f, off=o), g1.search_many(flat=f, off=o)
        assert (ba.s == b1.s).all() and (ba.e == b1.e).all() and (bb.s == b1.s).all() and (bb.e == b1.e).all()
    oi = O.OracleIndex.from_bwt(g1.export_bwt(), g1.export_cs(), 4)
    so, eo = oi.count_batch(rflat[:int(roff[4000])], roff[:4001], nthreads=8)
    b = ga.search_many(flat=rflat, off=roff)
    assert (so == b.s[:4000]).all() and (eo == b.e[:4000]).all()
    oi.close()
    ga.close(); gb.close(); g1.close()
    small = F.FMIndex(F.Text.with_max_character(W.dna_text_np(1 << 20, 3), 4))
    assert not small.has_pair_index() and small.kmer_k() == 0
    wide_alpha = F.FMIndex(F.Text.with_max_character(W.byte_text_np(1 << 24, 4), 255))
    assert not wide_alpha.has_pair_index() and wide_alpha.kmer_k() == 0
    # the named flags are still honoured next to the veto
    only_pair = F.FMIndex(tx, plain=True, pair_index=True)
    assert only_pair.has_pair_index() and only_pair.kmer_k() == 0
    only_pair.close()


def test_the_default_rlfm_index_gets_the_kmer_table():
    """Round 6: a run-length index over u8 symbols of >= 2^24 symbols gets the k-mer start table by default (k = 3 at a
    byte alphabet); FMX_FLAG_PLAIN vetoes; same (s, e) -- the oracle's RLFM path."""
    n = 1 << 24
    t = W.byte_text_np(n, 7)
    tx = F.Text(t)
    gd = F.RLFMIndex(tx)
    gp = F.RLFMIndex(tx, plain=True)
    assert gd.kmer_k() >= 2 and gp.kmer_k() == 0 and gd.heap_size() > gp.heap_size()      # (k = 2 here: the table stays below n / 2 bytes; 3 at n = 2^30)
    flat, off, _ = W.substring_patterns_np(t, 20000, 16, 8)
    rflat, roff = W.ragged_patterns_np(20000, 7, 255, 9)
    for f, o in ((flat, off), (rflat, roff)):
        a, b = gd.search_many(flat=f, off=o), gp.search_many(flat=f, off=o)
        assert (a.s == b.s).all() and (a.e == b.e).all()
    oi = O.OracleIndex.from_bwt(gp.export_bwt(), gp.export_cs(), 255, kind="rlfm")
    so, eo = oi.count_batch(flat[:4000 * 16], off[:4001], nthreads=8)
    a = gd.search_many(flat=flat, off=off)
    assert (so == a.s[:4000]).all() and (eo == a.e[:4000]).all()
    oi.close()
    gd.close(); gp.close()
    assert F.RLFMIndex(F.Text(W.byte_text_np(1 << 20, 7))).kmer_k() == 0          # a shorter text stays plain
