"""Character = u16 / u32 / u64 (character.rs:38-42): same kernels, wider symbols; parity against
the oracle's wide-symbol path.  L up to 17 bits -> up to five wavelet levels."""
import numpy as np
import pytest

import fm_index_amd as F
from fm_index_amd import workload as W
from oracle import fm_oracle as O

pytestmark = pytest.mark.gpu


def _text(n, alphabet, dtype, seed, spread=1):
    t = ((W.splitmix64_np(seed, 0, n) % np.uint64(alphabet)) * np.uint64(spread) + np.uint64(1)).astype(dtype)
    t[-1] = 0
    return t


def _patterns(text, npat, mmax, seed):
    n = len(text)
    pos = (W.splitmix64_np(seed, 0, npat) % np.uint64(n - 1)).astype(np.int64)
    lens = (W.splitmix64_np(seed + 1, 0, npat) % np.uint64(mmax + 1)).astype(np.int64)
    pats = []
    for k in range(npat):
        p = text[pos[k]:min(pos[k] + lens[k], n - 1)].copy()
        if k % 3 == 0 and len(p):          # mutate one symbol: exercises the early exit
            p[len(p) // 2] = text[(pos[k] * 7 + 3) % (n - 1)]
        pats.append(p)
    return pats


@pytest.mark.parametrize("dtype,maxc,alphabet,spread", [
    (np.uint16, 1000, 1000, 1),          # L = 10: [4,3,3]
    (np.uint16, 65535, 300, 200),        # L = 16: [4,4,4,4]   (Text::new on u16)
    (np.uint32, 100000, 5000, 19),       # L = 17: five levels
    (np.uint32, 6, 6, 1),                # small alphabet in a wide type: single level
    (np.uint64, 70000, 900, 77),         # usize/u64 symbols are narrowed on the host
])
@pytest.mark.parametrize("kind", ["fm", "rlfm"])
def test_wide_symbols_count_locate_and_trait(dtype, maxc, alphabet, spread, kind):
    n = 6000
    t = _text(n, alphabet, dtype, 11 + maxc, spread)
    cls = F.FMIndexWithLocate if kind == "fm" else F.RLFMIndexWithLocate
    gi = cls(F.Text.with_max_character(t, maxc), 2)
    t32 = t.astype(np.uint32)
    oi = O.OracleIndex(t32, maxc, level=2, kind=kind)
    pats = _patterns(t, 600, 6, 5)
    flat, off = F.pack_patterns(pats, dtype)
    gb = gi.search_many(flat=flat, off=off)
    os_, oe = oi.count_batch(flat.astype(np.uint32), off)
    assert (gb.s == os_).all() and (gb.e == oe).all()
    goff, gpos = gb.locate()
    ooff, opos = oi.locate_batch(os_, oe)
    assert (goff == ooff).all() and (gpos == opos).all()
    for k in range(0, 600, 41):
        if len(pats[k]):
            assert int(gb.counts[k]) == len(O.naive_search(t32, pats[k].astype(np.uint32)))
    rows = np.arange(0, n, 7)
    assert (gi.get_l(rows) == oi.get_l(rows)).all()
    assert (gi.lf_map(rows) == oi.lf_map(rows)).all()
    assert (gi.get_sa(rows) == oi.get_sa(rows)).all()
    syms = np.unique(t32)[:50]
    cc, ii = np.meshgrid(syms, np.arange(0, n + 1, 13))
    assert (gi.lf_map2(cc.ravel(), ii.ravel()) == oi.lf_map2(cc.ravel(), ii.ravel())).all()
    if dtype != np.uint64:
        assert (gi.export_bwt() == O.OracleIndex(t32, maxc).get_l(np.arange(n))).all() if kind == "fm" else True


def test_wide_symbol_errors():
    t = np.array([5, 70000, 3, 0], dtype=np.uint32)
    with pytest.raises(F.Error) as ei:
        F.FMIndex(F.Text.with_max_character(t, 1000))
    assert ei.value.code == F._lib.ERR_SYMBOL_RANGE
    gi = F.FMIndex(F.Text.with_max_character(t, 70000))
    with pytest.raises(F.Error) as ei:
        gi.search(np.array([70001], dtype=np.uint32))
    assert ei.value.code == F._lib.ERR_SYMBOL_RANGE
    bad = np.array([0, 4, 0], dtype=np.uint16)
    with pytest.raises(F.Error) as ei:
        F.FMIndex(F.Text.with_max_character(bad, 9))
    assert str(ei.value) == "invalid text: the given text must not start with zero character"


def test_u16_suffix_array_matches_oracle():
    t = _text(30000, 40000, np.uint16, 3)
    gi = F.FMIndex(F.Text(t), keep_sa=True)
    assert (gi.export_sa() == O.suffix_array(t.astype(np.uint32))).all()
    assert gi.verify_sa() == 0


def test_u64_text_resident_in_hbm():
    """Character = u64 / usize text already on the device (fmx_build_dev, sym_bytes 8; round 4): narrowed by a kernel; same
    index as the host path, and a symbol above max_character is reported the same way."""
    import torch
    n = 20000
    t = _text(n, 900, np.uint64, 5, 77)
    maxc = int(t.max())
    dt = torch.from_numpy(t.view(np.int64)).cuda()
    gi = F.FMIndexWithLocate.from_device_text(dt.data_ptr(), n, maxc, level=2, sym_bytes=8)
    hi = F.FMIndexWithLocate(F.Text.with_max_character(t, maxc), 2)
    assert gi._lib.fmx_sym_bytes(gi.handle()) == 4 == hi._lib.fmx_sym_bytes(hi.handle())
    flat, off = F.pack_patterns(_patterns(t, 500, 6, 9), np.uint64)     # host-pointer queries take the caller's u64
    a, b_ = gi.search_many(flat=flat, off=off), hi.search_many(flat=flat, off=off)
    assert (a.s == b_.s).all() and (a.e == b_.e).all() and int(a.counts.sum()) > 0
    assert (a.locate()[1] == b_.locate()[1]).all()
    assert gi.heap_size() == hi.heap_size()
    gi.close(); hi.close()
    dt[100] = maxc + 1
    with pytest.raises(F.Error) as ei:
        F.FMIndex.from_device_text(dt.data_ptr(), n, maxc, sym_bytes=8)
    assert ei.value.code == F._lib.ERR_SYMBOL_RANGE
