"""get_f / fl_map / iter_chars_forward (backend.rs:17-19, wrapper.rs:175-183) through the ABI."""
import numpy as np
import pytest

import fm_index_amd as F
from fm_index_amd import workload as W
from oracle import fm_oracle as O

pytestmark = pytest.mark.gpu


def b(s):
    return s.encode("latin-1")


@pytest.mark.parametrize("cls,kind", [(F.FMIndexWithLocate, "fm"), (F.RLFMIndexWithLocate, "rlfm")])
def test_known_answers_forward(golden, cls, kind):
    g = golden["mississippi"]
    idx = cls(F.Text(b(g["text"])), 2)
    assert idx.fl_map(np.arange(12)).tolist() == g["fl_map"]["expected"]      # fm_index.rs:163-173
    assert bytes(int(x) for x in idx.get_f(np.arange(12))) == bytes(sorted(b(g["text"])))
    r = golden["readme"]
    index = cls(F.Text(b(r["text"])), r["level"])
    search = index.search(b(r["pattern"]))
    m = list(search.iter_matches())[3]
    it = m.iter_chars_forward()
    assert bytes(next(it) for _ in range(20)) == b(r["forward_20_from_match_3"])  # README.md:78-85


@pytest.mark.parametrize("maxc,alphabet,n,dtype", [(4, 4, 3000, np.uint8), (255, 255, 3000, np.uint8),
                                                   (49, 2, 3000, np.uint8), (255, 3, 70000, np.uint8),
                                                   (3000, 2500, 4000, np.uint16)])
@pytest.mark.parametrize("kind", ["fm", "rlfm"])
def test_forward_every_row(maxc, alphabet, n, dtype, kind):
    t = ((W.splitmix64_np(maxc + alphabet, 0, n) % np.uint64(alphabet)) + np.uint64(48 if maxc == 49 else 1)).astype(dtype)
    t[-1] = 0
    cls = F.FMIndex if kind == "fm" else F.RLFMIndex
    gi = cls(F.Text.with_max_character(t, maxc))
    oi = O.OracleIndex(t if dtype == np.uint8 else t.astype(np.uint32), maxc, kind=kind)
    rows = np.arange(n) if n <= 4000 else np.arange(0, n, 17)
    assert (gi.get_f(rows) == oi.get_f(rows)).all()
    assert (gi.fl_map(rows) == oi.fl_map(rows)).all()
    # FL is the inverse of LF
    assert (gi.lf_map(gi.fl_map(rows)) == rows).all()
