"""Multi-pieces index (src/multi_pieces.rs, tests/test_multi_pieces.rs, examples/multi_pieces.rs):
oracle pinned on the reference's known answers and unit-test properties (CPU), then the GPU
path against the oracle and a brute-force scan with piece ids (tests/testutil/mod.rs:62-86)."""
import numpy as np
import pytest

from fm_index_amd import workload as W
from oracle import fm_oracle as O


def b(s):
    return s.encode("latin-1")


def multi_text(n, alphabet, seed):
    """tests/testutil/mod.rs:7-32 rules: no leading zero, no double zero, nonzero at n-2, zero at n-1."""
    r = (W.splitmix64_np(seed, 0, 2 * n) % np.uint64(alphabet)).astype(np.uint8)
    t = r[:n].copy()
    for i in range(n - 1):
        if t[i] == 0 and (i == 0 or t[i - 1] == 0):
            t[i] = 1 + r[n + i] % (alphabet - 1)
    if t[n - 2] == 0:
        t[n - 2] = 1 + r[n] % (alphabet - 1)
    if n >= 3 and t[n - 3] == 0 and t[n - 2] == 0:
        t[n - 2] = 2
    t[n - 1] = 0
    return t


def naive(text, pattern, prefix=False, suffix=False):
    """NaiveSearchIndex::do_search (tests/testutil/mod.rs:62-86): [(position, piece_id)]."""
    text, pattern = bytes(text), bytes(pattern)
    out, piece = [], 0
    for i in range(len(text) - len(pattern) + 1):
        if text[i] == 0:
            piece += 1
        if (not prefix or i == 0 or text[i - 1] == 0) and \
           (not suffix or i + len(pattern) == len(text) or text[i + len(pattern)] == 0) and \
           text[i:i + len(pattern)] == pattern:
            out.append((i, piece))
    return out


# ----------------------------------------------------------------- oracle (CPU) -------------
def test_oracle_multi_example_known_answers(golden):
    g = golden["multi_pieces_example"]
    idx = O.OracleIndex(b(g["text"]), 255, level=g["level"], kind="multi")
    assert idx.pieces_count() == 3
    assert idx.count(b("star")) == g["count_star"]
    s, e = idx.search(b("How I wonder"))
    assert sorted(idx.piece_id(idx.match_rows(s, e)).tolist()) == g["piece_ids_how_i_wonder_sorted"]
    s, e = idx.search(b("Twinkle"))                          # search_prefix: (0, len), filter
    assert sorted(idx.piece_id(idx.match_rows(s, e, True)).tolist()) == g["prefix_twinkle_piece_ids_sorted"]
    s, e = idx.search(b("what you are!\n"), s0e0=(0, idx.pieces_count()))   # search_suffix
    assert sorted(idx.piece_id(idx.match_rows(s, e)).tolist()) == g["suffix_what_you_are_piece_ids_sorted"]


def test_oracle_multi_unit_properties(golden):
    t = b(golden["multi_pieces_foo_bar_baz"]["text"])      # multi_pieces.rs:277-297
    idx = O.OracleIndex(t, 255, level=0, kind="multi")
    sa = O.suffix_array(t)
    assert idx.piece_id(np.arange(len(t))).tolist() == [t[:p].count(0) for p in sa]
    for seed in range(10):                                   # multi_pieces.rs:251-275, 299-324
        tt = multi_text(512, 8, 100 + seed)
        idx = O.OracleIndex(tt, 255, level=0, kind="multi")
        sa = [int(x) for x in O.suffix_array(tt, naive=True)]
        assert [int(x) for x in O.suffix_array(tt)] == sa
        inv = np.zeros(512, dtype=int)
        inv[sa] = np.arange(512)
        assert idx.lf_map(np.arange(512)).tolist() == [int(inv[(p - 1) % 512]) for p in sa]
        assert idx.piece_id(np.arange(512)).tolist() == [bytes(tt)[:p].count(0) for p in sa]


# ----------------------------------------------------------------- GPU ---------------------
# wide = True: the 64-bit engine (FMX_FLAG_FORCE_WIDE; multi-pieces on it since round 4: multi_pieces.rs is usize throughout)
@pytest.mark.gpu
@pytest.mark.parametrize("wide", [False, True])
def test_gpu_multi_example(golden, wide):
    import fm_index_amd as F
    g = golden["multi_pieces_example"]
    index = F.FMIndexMultiPiecesWithLocate(F.Text(b(g["text"])), g["level"], force_wide=wide)
    assert index.pieces_count() == 3 and index.is_wide() == wide
    assert index.search(b("star")).count() == g["count_star"]
    ids = sorted(m.piece_id() for m in index.search(b("How I wonder")).iter_matches())
    assert ids == g["piece_ids_how_i_wonder_sorted"]
    pre = []
    for m in index.search(b(" in the dark")).iter_matches():
        chars = []
        for c in m.iter_chars_backward():
            if c == ord(" "):
                break
            chars.append(c)
        pre.append(bytes(chars).decode())
    assert pre == g["backward_until_space_from_in_the_dark"]
    post = []
    for m in index.search(b("ing ")).iter_matches():
        chars = []
        for c in m.iter_chars_forward():
            if c == ord(","):
                break
            chars.append(c)
        post.append(bytes(chars).decode())
    assert post == g["forward_until_comma_from_ing"]
    assert sorted(index.search_prefix(b("Twinkle")).piece_ids()) == g["prefix_twinkle_piece_ids_sorted"]
    assert sorted(index.search_suffix(b("what you are!\n")).piece_ids()) == \
        g["suffix_what_you_are_piece_ids_sorted"]
    sm = golden["multi_pieces_small"]
    idx = F.FMIndexMultiPiecesWithLocate(F.Text(b(sm["text"])), sm["level"], force_wide=wide)
    s = idx.search(b("a"))
    assert s.count() == sm["count"] and s.locate_all() == sm["positions"] and s.piece_ids() == sm["piece_ids"]


@pytest.mark.gpu
@pytest.mark.parametrize("wide", [False, True])
def test_gpu_multi_vs_oracle_and_bruteforce(wide, tmp_path):
    """tests/test_multi_pieces.rs:44-272: count, locate, piece ids, prefix / suffix / exact."""
    import fm_index_amd as F
    for ti in range(12):
        size = 20 + int(W.splitmix64_np(700 + ti, 0, 1)[0] % np.uint64(1000))
        if wide and ti >= 10:                               # texts that cross many superblocks of 2^12 rows
            size = 30000 + 4001 * ti
        t = multi_text(size, 8, 800 + ti)
        level = ti % 4
        maxc = 255 if ti % 3 else 7                         # one 3-bit level / two 4-bit levels
        gi = F.FMIndexMultiPiecesWithLocate(F.Text.with_max_character(t, maxc), level, force_wide=wide)
        oi = O.OracleIndex(t, maxc, level=level, kind="multi")
        assert gi.is_wide() == wide
        assert gi.pieces_count() == oi.pieces_count() == int((t == 0).sum())
        rows = np.arange(size)
        assert (gi.get_l(rows) == oi.get_l(rows)).all()
        assert (gi.lf_map(rows) == oi.lf_map(rows)).all()
        assert (gi.get_sa(rows) == oi.get_sa(rows)).all()
        assert (gi.piece_id(rows) == oi.piece_id(rows)).all()
        assert (gi.get_f(rows) == oi.get_f(rows)).all()
        assert (gi.fl_map(rows) == oi.fl_map(rows)).all()
        cc, ii = np.meshgrid(np.arange(min(maxc, 8) + 1), np.arange(size + 1))
        assert (gi.lf_map2(cc.ravel(), ii.ravel()) == oi.lf_map2(cc.ravel(), ii.ravel())).all()
        flat, off = W.ragged_patterns_np(60, 6, 7, 900 + ti)
        pc = oi.pieces_count()
        for mode, (pre, suf) in {"search": (False, False), "prefix": (True, False),
                                 "suffix": (False, True), "exact": (True, True)}.items():
            se = np.tile(np.array([0, pc], dtype=np.uint64), 60) if suf else None
            gb = gi.search_many(flat=flat, off=off, s0e0=se)
            os_, oe = oi.count_batch(flat, off, se)
            assert (gb.s == os_).all() and (gb.e == oe).all()
            goff, grows = gi.match_rows_many(gb.s, gb.e, pre)
            for k in range(60):
                p = flat[int(off[k]):int(off[k + 1])]
                exp_rows = oi.match_rows(int(os_[k]), int(oe[k]), pre)
                got = grows[int(goff[k]):int(goff[k + 1])]
                assert (got == exp_rows).all(), (mode, k)
                if len(p) == 0:
                    continue
                if size > 2000 and k % 7:                   # (the brute-force scan is a Python loop over the text)
                    continue
                exp = naive(t, p, pre, suf)
                pos = gi.get_sa(got)
                ids = gi.piece_id(got)
                assert sorted(zip(pos.tolist(), ids.tolist())) == sorted(exp), (mode, ti, k)
        # locate through the batched walk; iter_chars_forward ends at a piece's end marker (wrapper.rs:172-183)
        gb = gi.search_many(flat=flat, off=off)
        goff, gpos = gb.locate()
        ooff, opos = oi.locate_batch(gb.s, gb.e)
        assert (goff == ooff).all() and (gpos == opos).all()
        some = np.arange(0, size, max(1, size // 200), dtype=np.uint64)
        syms, lens, nxt = gi.extract_many(some, 12, forward=True)
        for q, row in enumerate(some.tolist()):
            i, want = row, []
            while len(want) < 12:
                nx = int(oi.fl_map(np.array([i], np.uint64))[0])
                if nx == 0xFFFFFFFFFFFFFFFF:
                    break
                want.append(int(oi.get_f(np.array([i], np.uint64))[0]))
                i = nx
            assert int(lens[q]) == len(want) and syms[q, :len(want)].tolist() == want, (ti, row)
            assert int(nxt[q]) == (i if len(want) == 12 else 0xFFFFFFFFFFFFFFFF)
        if ti % 4 == 1:                                     # the index file
            path = str(tmp_path / ("multi%d.fmx" % ti))
            gi.save(path)
            li = type(gi).load(path)
            assert li.is_wide() == wide and li.pieces_count() == gi.pieces_count()
            assert (li.piece_id(rows) == oi.piece_id(rows)).all() and (li.lf_map(rows) == oi.lf_map(rows)).all()
            lb = li.search_many(flat=flat, off=off)
            assert (lb.s == gb.s).all() and (lb.e == gb.e).all()
            li.close()
        gi.close()
