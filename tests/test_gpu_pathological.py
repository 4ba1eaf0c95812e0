"""Suffix sorter / RLFM structures on pathological texts at sizes beyond the oracle's comfort:
the suffix array is checked ON THE DEVICE (sortedness of every adjacent pair + permutation),
counts against bytes.count() of the text, RLFM (s, e) against FM (s, e)."""
import numpy as np
import pytest

import fm_index_amd as F

pytestmark = pytest.mark.gpu

N = 1 << 20


def fib_word(n):
    a, b = b"\x01", b"\x01\x02"
    while len(b) < n:
        a, b = b, b + a
    return np.frombuffer(b[:n], dtype=np.uint8).copy()


def texts():
    yield "all-equal", np.full(N, 1, dtype=np.uint8)
    yield "period-2", np.tile(np.array([1, 2], dtype=np.uint8), N // 2)
    yield "fibonacci", fib_word(N)
    blk = (np.arange(4096) * 2654435761 % 251 + 1).astype(np.uint8)
    yield "block-repeat", np.tile(blk, N // 4096)


@pytest.mark.parametrize("name,t", list(texts()), ids=[n for n, _ in texts()])
def test_pathological_text(name, t):
    t = t.copy()
    t[-1] = 0
    fm = F.FMIndexWithLocate(F.Text.with_max_character(t, 255), 4, keep_sa=True)
    assert fm.verify_sa() == 0
    rl = F.RLFMIndexWithLocate(F.Text.with_max_character(t, 255), 4)
    runs = int(rl._lib.fmx_num_runs(rl.handle()))
    assert runs < N // 8, runs          # these texts compress: few runs
    tb = t.tobytes()
    pats = [tb[k:k + m] for k, m in ((0, 1), (5, 7), (1000, 33), (N // 2, 64), (N - 200, 100), (7, 3))]
    pats += [b"\x03\x03", b"\x01\x02\x02\x01", b"\x02" * 9]
    a = fm.search_many(pats)
    b = rl.search_many(pats)
    assert (a.s == b.s).all() and (a.e == b.e).all()
    for p, c in zip(pats, a.counts):
        # overlapping occurrences: count by scanning
        k, pos = 0, tb.find(p)
        while pos != -1 and k <= 50000:
            k += 1
            pos = tb.find(p, pos + 1)
        if k <= 50000:
            assert int(c) == k, (name, p[:8], int(c), k)
    # locate: positions of a handful of matches really hold the pattern (FM and RLFM agree)
    sel = [i for i, c in enumerate(a.counts) if 0 < int(c) <= 5000][:4]
    if sel:
        off, pos = fm.locate_many(a.s[sel], a.e[sel])
        off2, pos2 = rl.locate_many(b.s[sel], b.e[sel])
        assert (pos == pos2).all()
        for j, i in enumerate(sel):
            for q in pos[int(off[j]):int(off[j + 1])][:50]:
                assert tb[int(q):int(q) + len(pats[i])] == pats[i]
