"""Opt-in k-mer start table (FMX_FLAG_KMER_TABLE): a pattern's first k steps come from one table
lookup.  (s, e) must stay bit-identical to the stepwise path and to the oracle -- including the
equal pair the reference's early exit leaves (wrapper.rs:111-113) when the k-mer itself does
not occur -- alone and combined with the pair index."""
import numpy as np
import pytest

import fm_index_amd as F
from fm_index_amd import workload as W
from oracle import fm_oracle as O

pytestmark = pytest.mark.gpu


def _check(text, maxc, flat, off, s0e0=None, expect_k=None):
    txt = F.Text.with_max_character(text, maxc)
    gk = F.FMIndex(txt, kmer_table=True)
    gkp = F.FMIndex(txt, kmer_table=True, pair_index=True)
    oi = O.OracleIndex(text, maxc)
    os_, oe = oi.count_batch(flat, off, s0e0)
    for g in (gk, gkp):
        b = g.search_many(flat=flat, off=off, s0e0=s0e0)
        assert (b.s == os_).all() and (b.e == oe).all()
        assert (b.counts == oe - os_).all()
    if expect_k is not None:
        assert gk.kmer_k() == expect_k
    return gk


@pytest.mark.parametrize("n,k", [(70001, 6), (1 << 20, 8), (300, 2), (40, 0), (9, 0)])
def test_kmer_table_dna(n, k):
    t = W.dna_text_np(n, 11 + n)
    # k = min(12, floor(log2(n / 16) / 2)): the table never exceeds n/2 bytes
    flat, off = W.ragged_patterns_np(4000, 24, 4, 5 + n)          # random: mostly absent k-mers
    g = _check(t, 4, flat, off, expect_k=k)
    if n > 100:
        for m in (k, k + 1, 2 * k + 3, max(k - 1, 1)):
            flat2, off2, _ = W.substring_patterns_np(t, 3000, m, 60 + m)   # present patterns
            _check(t, 4, flat2, off2)
    assert g.kmer_k() == k


def test_kmer_table_zero_and_out_of_range_symbols_take_the_stepwise_path():
    t = W.dna_text_np(50000, 3)
    g = F.FMIndex(F.Text.with_max_character(t, 4), kmer_table=True)
    k = g.kmer_k()
    assert k == 5
    base, off, _ = W.substring_patterns_np(t, 64, 20, 8)
    base = base.reshape(64, 20).copy()
    for r in range(64):
        base[r, 19 - (r % 10)] = 0                 # a zero inside / outside the last k symbols
    flat = base.reshape(-1)
    _check(t, 4, flat, off)
    bad = base.copy()
    bad[bad == 0] = 6                               # > max_character inside the last k symbols
    with pytest.raises(F.Error) as ei:
        g.search_many(flat=bad.reshape(-1), off=off)
    assert ei.value.code == F._lib.ERR_SYMBOL_RANGE
    # the terminator row itself: pattern "\\0" alone and "x\\0"
    flat2, off2 = F.pack_patterns([bytes([0]), bytes([1, 0]), bytes([0, 1])])
    _check(t, 4, flat2, off2)


def test_kmer_table_refinement_ignores_the_table():
    t = W.dna_text_np(30000, 5)
    g1 = F.FMIndex(F.Text.with_max_character(t, 4))
    flat, off, _ = W.substring_patterns_np(t, 800, 5, 4)
    first = g1.search_many(flat=flat, off=off)
    se = np.stack([first.s, first.e], axis=1).reshape(-1)
    flat2, off2 = W.ragged_patterns_np(800, 15, 4, 6)
    _check(t, 4, flat2, off2, s0e0=se)


@pytest.mark.parametrize("maxc,k", [(2, 9), (3, 4), (7, 3), (1, 9)])
def test_kmer_table_other_small_alphabets(maxc, k):
    n = 10000
    t = ((W.splitmix64_np(maxc, 0, n) % np.uint64(maxc)) + np.uint64(1)).astype(np.uint8)
    t[-1] = 0
    flat, off = W.ragged_patterns_np(3000, 30, maxc, 17)
    _check(t, maxc, flat, off, expect_k=k)
    flat2, off2, _ = W.substring_patterns_np(t, 2000, k + 5, 3)
    _check(t, maxc, flat2, off2)


def test_kmer_table_not_applicable_is_ignored():
    t = W.byte_text_np(5000, 4)
    assert F.FMIndex(F.Text(t), kmer_table=True).kmer_k() == 0           # sigma 255: 5000 rows are too few for k = 2
    assert F.RLFMIndex(F.Text.with_max_character(W.dna_text_np(5000, 2), 4)).kmer_k() == 0


def test_kmer_table_survives_save_load(tmp_path):
    t = W.dna_text_np(40000, 8)
    g = F.FMIndexWithLocate(F.Text.with_max_character(t, 4), 2, kmer_table=True, pair_index=True)
    flat, off = W.ragged_patterns_np(3000, 20, 4, 2)
    a = g.search_many(flat=flat, off=off)
    g.save(tmp_path / "k.fmx")
    g2 = F.FMIndexWithLocate.load(tmp_path / "k.fmx")
    assert g2.kmer_k() == g.kmer_k() == 5 and g2.has_pair_index()
    assert g2.heap_size() == g.heap_size()
    b = g2.search_many(flat=flat, off=off)
    assert (a.s == b.s).all() and (a.e == b.e).all()


@pytest.mark.parametrize("kind", ["fm", "rlfm", "multi"])
@pytest.mark.parametrize("maxc,alphabet,n,k", [(255, 255, 1 << 20, 2), (63, 50, 70000, 2), (20, 20, 300000, 2),
                                              (15, 12, 70000, 3)])
def test_kmer_table_generic_kernels(kind, maxc, alphabet, n, k):
    """byte / protein-sized alphabets go through the generic count kernels (any kind, 2 levels)."""
    t = ((W.splitmix64_np(maxc + n, 0, n) % np.uint64(alphabet)) + np.uint64(1)).astype(np.uint8)
    if kind == "multi":
        t[np.arange(997, n - 5, 4001)] = 0
    if kind == "rlfm":                      # give the run-length index some runs
        t = np.repeat(t[: n // 3 + 1], 3)[:n].copy()
    t[-1] = 0
    txt = F.Text.with_max_character(t, maxc)
    cls = {"fm": F.FMIndex, "rlfm": F.RLFMIndex, "multi": F.FMIndexMultiPieces}[kind]
    plain = cls(txt)
    g = cls(txt, kmer_table=True)
    # FM / multi-pieces over two wavelet levels: the flag is ignored (the lookup would cost more
    # than the cache-resident steps it replaces); RLFM always takes it
    assert g.kmer_k() == (k if (kind == "rlfm" or maxc <= 15) else 0)
    oi = O.OracleIndex(t, maxc, kind=kind)
    flat, off = W.ragged_patterns_np(3000, 9, alphabet, 5 + maxc)
    flat2, off2, _ = W.substring_patterns_np(t, 3000, k + 3, 7)
    flat3, off3, _ = W.substring_patterns_np(t, 2000, k, 8)
    for fl, of in ((flat, off), (flat2, off2), (flat3, off3)):
        a = g.search_many(flat=fl, off=of)
        b = plain.search_many(flat=fl, off=of)
        os_, oe = oi.count_batch(fl, of)
        assert (a.s == os_).all() and (a.e == oe).all()
        assert (b.s == os_).all() and (b.e == oe).all()


@pytest.mark.parametrize("kind", ["rlfm", "multi"])
def test_kmer_table_other_kinds_save_load_and_zero_symbols(tmp_path, kind):
    n = 60000
    t = ((W.splitmix64_np(31, 0, n) % np.uint64(4)) + np.uint64(1)).astype(np.uint8)
    if kind == "multi":
        t[np.arange(501, n - 5, 3001)] = 0
    else:
        t = np.repeat(t[: n // 4 + 1], 4)[:n].copy()
    t[-1] = 0
    txt = F.Text.with_max_character(t, 4)
    cls = F.RLFMIndexWithLocate if kind == "rlfm" else F.FMIndexMultiPiecesWithLocate
    g = cls(txt, 2, kmer_table=True)
    assert g.kmer_k() == 5
    oi = O.OracleIndex(t, 4, kind=kind, level=2)
    pats = [bytes([1, 2, 3, 4, 1, 2, 3]), bytes([2, 2, 2, 2, 2, 2]), bytes([4, 3, 0, 1, 2, 3, 4]),
            bytes([0]), bytes([1, 0]), bytes([3, 3, 3, 3, 3]), bytes([1, 2, 3, 4])]
    flat, off = F.pack_patterns(pats)
    flat2, off2, _ = W.substring_patterns_np(t, 2000, 9, 4)
    flat3, off3 = W.ragged_patterns_np(2000, 12, 4, 6)
    g.save(tmp_path / "k.fmx")
    g2 = cls.load(tmp_path / "k.fmx")
    assert g2.kmer_k() == 5
    for fl, of in ((flat, off), (flat2, off2), (flat3, off3)):
        os_, oe = oi.count_batch(fl, of)
        for h in (g, g2):
            b = h.search_many(flat=fl, off=of)
            assert (b.s == os_).all() and (b.e == oe).all()
    if kind == "multi":   # the start-of-piece / end-of-piece searches go through s0e0 or the row filter
        plain = cls(txt, 2)
        for p in (bytes([1, 2, 3, 4, 1]), bytes([2, 2, 2, 2, 2, 2]), bytes(t[502:510])):
            assert g.search_prefix(p).count() == plain.search_prefix(p).count()
            assert g.search_suffix(p).count() == plain.search_suffix(p).count()
            assert g.search_exact(p).count() == plain.search_exact(p).count()
            assert sorted(g.search_prefix(p).piece_ids()) == sorted(plain.search_prefix(p).piece_ids())
