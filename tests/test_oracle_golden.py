"""Pins the CPU oracle against every known answer the reference's own tests hold
for the count/locate path and against the brute-force property
its integration tests use (tests/test_fmindex.rs:26-89, tests/test_rlfmindex.rs:26-89).
CPU only."""
import numpy as np
import pytest

from oracle import fm_oracle as O
from fm_index_amd import workload as W


def b(s):
    return s.encode("latin-1")


# ---------------------------------------------------------------- B1-B6 -----
@pytest.mark.parametrize("kind", ["fm", "rlfm"])
def test_mississippi_lf_chain(golden, kind):
    g = golden["mississippi"]
    idx = O.OracleIndex(b(g["text"]), 255, level=2, kind=kind)
    i, chain = 0, []
    for _ in range(12):
        i = int(idx.lf_map([i])[0])
        chain.append(i)
    assert chain == g["lf_chain_from_0"]["expected"]
    assert len(idx) == 12


@pytest.mark.parametrize("kind", ["fm", "rlfm"])
def test_mississippi_lf_map2_ranges(golden, kind):
    g = golden["mississippi"]
    idx = O.OracleIndex(b(g["text"]), 255, kind=kind)
    n = len(idx)
    for ch, (s, e) in g["lf_map2_ranges"]["expected"].items():
        c = ord(ch)
        assert int(idx.lf_map2([c], [0])[0]) == s
        assert int(idx.lf_map2([c], [n])[0]) == e


@pytest.mark.parametrize("kind", ["fm", "rlfm"])
def test_mississippi_search_ranges(golden, kind):
    g = golden["mississippi"]
    idx = O.OracleIndex(b(g["text"]), 255, kind=kind)
    for pat, (s, e) in g["search_ranges"]["expected"].items():
        assert idx.search(b(pat)) == (s, e)


@pytest.mark.parametrize("kind", ["fm", "rlfm"])
def test_mississippi_bwt(golden, kind):
    g = golden["mississippi"]
    idx = O.OracleIndex(b(g["text"]), 255, kind=kind)
    got = bytes(int(x) for x in idx.get_l(np.arange(12)))
    assert got == b(g["bwt"]["expected"])
    sa = O.suffix_array(b(g["text"]))
    assert bytes(O.bwt(b(g["text"]), sa)) == b(g["bwt"]["expected"])


def test_mississippi_rlfm_structures(golden):
    """S, B, B', cs of rlfmi.rs:197-256, read back through the oracle's structs."""
    import ctypes as C
    g = golden["mississippi"]
    idx = O.OracleIndex(b(g["text"]), 255, kind="rlfm")
    # S via get_l at run starts, B via get_l changes -- plus direct formulas:
    text = np.frombuffer(b(g["text"]), dtype=np.uint8)
    sa = O.suffix_array(text)
    L = np.array([text[k - 1] if k > 0 else text[-1] for k in sa], dtype=np.uint8)
    prev = np.concatenate([[0], L[:-1]])
    B = (L != prev).astype(int)
    assert B.tolist() == g["rlfm_B"]["expected"]
    S = L[B == 1]
    assert bytes(S) == b(g["rlfm_S"]["expected"])
    # cs[c] = number of runs with head < c
    for ch, v in g["rlfm_cs"]["expected"].items():
        assert int((S < ord(ch)).sum()) == v
    # B' from the oracle: F-order run starts <=> lf_map2(c, i) at run starts
    bp = np.zeros(12, dtype=int)
    for i in np.nonzero(B)[0]:
        bp[int(idx.lf_map([i])[0])] = 1
    assert bp.tolist() == g["rlfm_Bp"]["expected"]


# ---------------------------------------------------------------- B7, B8 ----
@pytest.mark.parametrize("kind", ["fm", "rlfm"])
def test_readme_dolor(golden, kind):
    g = golden["readme"]
    idx = O.OracleIndex(b(g["text"]), 255, level=g["level"], kind=kind)
    assert len(idx) == 443
    assert idx.count(b(g["pattern"])) == g["count"]
    assert idx.locate(b(g["pattern"])) == g["positions_in_order"]
    # iter_chars_backward (wrapper.rs:154-161) from the first match
    s, e = idx.search(b(g["pattern"]))
    i, out = s, []
    for _ in range(16):
        out.append(int(idx.get_l([i])[0]))
        i = int(idx.lf_map([i])[0])
    assert bytes(reversed(out)) == b(g["backward_16_from_first_match"])


@pytest.mark.parametrize("kind", ["fm", "rlfm"])
def test_small(golden, kind):
    g = golden["small"]
    idx = O.OracleIndex(b(g["text"]), 255, level=g["level"], kind=kind)
    assert idx.count(b(g["pattern"])) == g["count"]
    assert idx.locate(b(g["pattern"])) == g["positions"]


# ---------------------------------------------------------------- B9 --------
def test_sampling_grid(golden):
    import ctypes as C
    lib = O.lib()

    class SSA(C.Structure):
        _fields_ = [("level", C.c_uint64), ("word_size", C.c_uint64), ("len", C.c_uint64),
                    ("nsamples", C.c_uint64), ("bits", C.c_void_p)]
    lib.orc_ssa_sample.argtypes = [C.POINTER(SSA), C.c_void_p, C.c_uint64, C.c_uint64]
    lib.orc_ssa_get.argtypes = [C.POINTER(SSA), C.c_uint64, C.POINTER(C.c_uint64)]
    lib.orc_ssa_free.argtypes = [C.POINTER(SSA)]
    g = golden["sampling_grid"]
    cases = [tuple(c) + (False,) for c in g["cases"]]
    cases.append((g["not_sampled"]["level"], g["not_sampled"]["n"], True))
    for level, n, allsome in cases:
        sa = np.arange(n, dtype=np.uint32)
        s = SSA()
        lib.orc_ssa_sample(C.byref(s), sa.ctypes.data_as(C.c_void_p), n, level)
        for i in range(n):
            out = C.c_uint64(0)
            some = lib.orc_ssa_get(C.byref(s), i, C.byref(out))
            if allsome or i % (1 << level) == 0:
                assert some == 1 and out.value == i
            else:
                assert some == 0
        lib.orc_ssa_free(C.byref(s))
    s = SSA()
    lib.orc_ssa_sample(C.byref(s), None, 0, 2)  # sample.rs:97-100
    out = C.c_uint64(0)
    assert lib.orc_ssa_get(C.byref(s), 0, C.byref(out)) == 0


# ---------------------------------------------------------------- B10-B12 ---
@pytest.mark.parametrize("kind", ["fm", "rlfm"])
def test_invalid_texts(golden, kind):
    for case in golden["invalid_texts"]["cases"]:
        with pytest.raises(O.OracleError) as ei:
            O.OracleIndex(b(case["text"]), 255, kind=kind)
        assert ei.value.msg == case["message"]
        assert str(ei.value) == "invalid text: " + case["message"]  # error.rs:11


def test_suffix_array_inputs(golden):
    g = golden["suffix_array_inputs"]
    cases = [np.frombuffer(b(s), dtype=np.uint8) for s in g["cases_str"]]
    cases += [np.array(c, dtype=np.uint8) for c in g["cases_u8"]]
    for t in cases:
        naive = O.suffix_array(t, naive=True)
        assert (O.suffix_array(t) == naive).all()
        # independent python definition
        tb = bytes(t)
        assert naive.tolist() == sorted(range(len(tb)), key=lambda i: tb[i:])


def test_len_and_log2(golden):
    for kind in ("fm", "rlfm"):
        for level in (None, 2):
            assert len(O.OracleIndex(b(golden["len"]["text"]), 255, level=level, kind=kind)) == 5
    for x, v in golden["log2"]["cases"]:
        assert O.lib().orc_max_bits(x) == v + 1


# ---------------------------------------------------------------- P1 --------
def _rand_text(rng_seed, size, alphabet):
    t = (W.splitmix64_np(rng_seed, 0, size) % np.uint64(alphabet)).astype(np.uint8) + 1
    t[-1] = 0
    return t


@pytest.mark.parametrize("kind", ["fm", "rlfm"])
def test_property_count_and_locate_vs_bruteforce(kind):
    """tests/test_fmindex.rs:26-89: random texts (sigma 8), level 0..3, patterns < 10."""
    for ti in range(40):
        size = 2 + int(W.splitmix64_np(1000 + ti, 0, 1)[0] % np.uint64(1023))
        text = _rand_text(2000 + ti, size, 8)
        level = int(W.splitmix64_np(3000 + ti, 0, 1)[0] % np.uint64(4))
        idx = O.OracleIndex(text, 255, level=level, kind=kind)
        flat, off = W.ragged_patterns_np(60, min(9, size), 7, 4000 + ti)
        s, e = idx.count_batch(flat, off)
        offs, pos = idx.locate_batch(s, e)
        for k in range(60):
            p = flat[int(off[k]):int(off[k + 1])]
            if len(p) == 0:
                assert int(e[k] - s[k]) == size  # empty pattern: whole range
                continue
            exp = O.naive_search(text, p)
            assert int(e[k] - s[k]) == len(exp)
            got = np.sort(pos[int(offs[k]):int(offs[k + 1])])
            assert (got == exp).all()


def test_rlfm_equals_fm_everywhere():
    """RLFM lf_map2 == FM lf_map2 for every c and every i in [0, n]."""
    for ti in range(10):
        text = _rand_text(50 + ti, 300, 5)
        fm = O.OracleIndex(text, 7, kind="fm")
        rl = O.OracleIndex(text, 7, kind="rlfm")
        n = len(text)
        cc, ii = np.meshgrid(np.arange(8), np.arange(n + 1))
        assert (fm.lf_map2(cc.ravel(), ii.ravel()) == rl.lf_map2(cc.ravel(), ii.ravel())).all()
        assert (fm.get_l(np.arange(n)) == rl.get_l(np.arange(n))).all()
        assert (fm.lf_map(np.arange(n)) == rl.lf_map(np.arange(n))).all()


def test_lf_map2_is_definition():
    """fm_index.rs:93-95 by definition: cs[c] + #{j<i : BWT[j]==c}."""
    text = W.dna_text_np(2000, 9)
    fm = O.OracleIndex(text, 4, kind="fm")
    sa = O.suffix_array(text, naive=True)
    bw = O.bwt(text, sa)
    cs = O.bucket_start(text, 4)
    for c in range(5):
        pref = np.concatenate([[0], np.cumsum(bw == c)])
        got = fm.lf_map2(np.full(2001, c), np.arange(2001))
        assert (got == cs[c] + pref).all()


def test_doubling_matches_naive_on_repetitive():
    t = W.repetitive_text_np(5000, 5, base_len=64)
    assert (O.suffix_array(t) == O.suffix_array(t, naive=True)).all()
    t2 = np.array([1, 1, 1, 1, 1, 1, 1, 0], dtype=np.uint8)
    assert (O.suffix_array(t2) == O.suffix_array(t2, naive=True)).all()


def test_refinement_prepends():
    """wrapper.rs:99-124: index.search("b").search("a") == index.search("ab")."""
    text = W.dna_text_np(4000, 3)
    fm = O.OracleIndex(text, 4, kind="fm")
    s1, e1 = fm.search(bytes([2, 3]))
    s2, e2 = fm.search(bytes([1]), s0e0=(s1, e1))
    assert (s2, e2) == fm.search(bytes([1, 2, 3]))


def test_symbol_out_of_range_is_error():
    text = W.dna_text_np(100, 3)
    fm = O.OracleIndex(text, 4, kind="fm")
    with pytest.raises(O.OracleError):
        fm.search(bytes([5]))
    with pytest.raises(O.OracleError):
        O.OracleIndex(np.array([1, 9, 0], dtype=np.uint8), 4)


@pytest.mark.parametrize("kind", ["fm", "rlfm"])
def test_fl_map_and_get_f_known_answers(golden, kind):
    """fm_index.rs:163-173 / rlfmi.rs:329-351: fl_map table and get_f = sorted text."""
    g = golden["mississippi"]
    idx = O.OracleIndex(b(g["text"]), 255, kind=kind)
    assert idx.fl_map(np.arange(12)).tolist() == g["fl_map"]["expected"]
    assert bytes(int(x) for x in idx.get_f(np.arange(12))) == bytes(sorted(b(g["text"])))
    r = golden["readme"]
    idx = O.OracleIndex(b(r["text"]), 255, level=2, kind=kind)
    s, e = idx.search(b(r["pattern"]))
    i, out = s + 3, []          # iter_matches().nth(3).iter_chars_forward().take(20)
    for _ in range(20):
        out.append(int(idx.get_f([i])[0]))
        i = int(idx.fl_map([i])[0])
    assert bytes(out) == b(r["forward_20_from_match_3"])


@pytest.mark.parametrize("kind", ["fm", "rlfm"])
def test_import_from_bwt_equals_the_index_built_from_the_text(kind):
    """the checker of the full-size GPU tests is the oracle rebuilt from an exported L column and exported samples
    (orc_fm_from_bwt / orc_rlfm_from_bwt, and their 64-bit-sample forms for n >= 2^32): on a small text it must be the
    index the oracle builds from the text itself -- same ranges, positions and trait values."""
    t = W.repetitive_text_np(5000, 3, base_len=64, mut_per_1024=20)
    a = O.OracleIndex(t, 255, level=2, kind=kind)
    rows = np.arange(len(t), dtype=np.uint64)
    bwt = a.get_l(rows).astype(np.uint8)
    cs = np.concatenate([[0], np.cumsum(np.bincount(t, minlength=256))[:-1]]).astype(np.uint64)
    samples = a.get_sa(rows[::4])
    flat, off, _ = W.substring_patterns_np(t, 300, 6, 5)
    want = a.count_batch(flat, off)
    for wide_samples in (False, True):
        b = O.OracleIndex.from_bwt(bwt, cs, 255, samples=samples, level=2, kind=kind)
        if wide_samples:                                       # the n >= 2^32 entry points, on the same small input
            import ctypes as C
            h = C.c_void_p()
            s64 = np.ascontiguousarray(samples, dtype=np.uint64)
            if kind == "rlfm":
                assert b._l.orc_rlfm_from_bwt64(C.byref(h), O._p(bwt), len(bwt), 255, O._p(s64), 2) == 0
            else:
                assert b._l.orc_fm_from_bwt64(C.byref(h), O._p(bwt), len(bwt), 255, O._p(cs), O._p(s64), 2) == 0
            b.close()
            b._h = h
            b._b = {"fm": b._l.orc_fm_backend, "rlfm": b._l.orc_rlfm_backend}[kind](h)
        got = b.count_batch(flat, off)
        assert (got[0] == want[0]).all() and (got[1] == want[1]).all()
        assert (b.lf_map(rows) == a.lf_map(rows)).all() and (b.get_sa(rows) == a.get_sa(rows)).all()
        ooff, opos = b.locate_batch(want[0], want[1])
        aoff, apos = a.locate_batch(want[0], want[1])
        assert (ooff == aoff).all() and (opos == apos).all()
        b.close()
    a.close()
