"""Walk records (round 4; fmx_internal.h FmxDev::walk): a text-order FM index over one 3-bit
wavelet level with max_character <= 5 at levels 1..3 carries a second encoding of the BWT -- 112 rows per 128-byte
record with every row's phase SA[row] mod 2^level, the phase-0 rank and the per-symbol phase-1 ranks next to its
symbol -- and the batched locate walk (fmx_locate_f3t_kernel) reads nothing else: max(phase, 1) records and one sample
per hit (the sample index of a walk's last row comes out of the record of the row before it).  get_sa is unchanged as a
function (fm_index.rs:127-140): every row of every index is compared with the oracle's row-order answer, the step
count is sum(SA[row] mod 2^level) exactly, and the index without walk records (FMX_FLAG_NO_WALK_RECORDS) and the
row-order index give the same sequence."""
import os

import numpy as np
import pytest

import fm_index_amd as F
from fm_index_amd import workload as W
from oracle import fm_oracle as O

pytestmark = pytest.mark.gpu


def _text(seed, n, alpha):
    t = (W.splitmix64_np(seed, 0, n) % np.uint64(alpha)).astype(np.uint8) + 1
    t[-1] = 0
    return t


def _locate_steps(idx, s, e):
    lib = idx._lib
    lib.fmx_set_timing(idx.handle(), 1)
    _, pos = idx.locate_many(np.asarray(s, np.uint64), np.asarray(e, np.uint64))
    steps = int(lib.fmx_last_steps(idx.handle()))
    lib.fmx_set_timing(idx.handle(), 0)
    return np.asarray(pos, np.uint64), steps


@pytest.mark.parametrize("level", [1, 2, 3])
@pytest.mark.parametrize("maxc,alpha,n", [(4, 4, 5003), (4, 4, (1 << 17) + 77), (5, 5, 70001), (5, 3, 4099), (4, 2, 33000),
                                           (4, 4, 111), (4, 4, 112), (4, 4, 113), (4, 4, 225), (4, 4, 3)])
def test_walk_record_locate_equals_oracle_on_every_row(level, maxc, alpha, n):
    t = _text(300 + level + alpha + n % 7, n, alpha)
    gi = F.FMIndexWithLocate(F.Text.with_max_character(t, maxc), level, sampling="text")
    if n <= (1 << level):                                           # sample.rs:28-31: the level is forced to 0
        assert gi.level() == 0 and not gi.text_order() and not gi.walk_records()
        level = 0
    else:
        assert gi.text_order() and gi.walk_records()
    oi = O.OracleIndex(t, maxc, level=level, kind="fm")
    rows = np.arange(n)
    want = oi.get_sa(rows).astype(np.uint64)
    # the whole index as one interval (>= 2^16 hits: four walks per group and the write-combining ring)
    pos, steps = _locate_steps(gi, [0], [n])
    assert (pos == want).all()
    assert steps == int((want & np.uint64((1 << level) - 1)).sum())       # a walk is SA[row] mod 2^level steps, exactly
    # few hits (one walk per group, direct stores), ragged intervals, an empty one
    s = np.array([0, min(5, n), min(5, n), n // 2, max(n - 9, 0)], np.uint64)
    e = np.array([min(3, n), min(5, n), min(40, n), min(n // 2 + 70, n), n], np.uint64)
    pos, _ = _locate_steps(gi, s, e)
    assert (pos == np.concatenate([want[int(a):int(b)] for a, b in zip(s, e)])).all()
    # the scalar trait call and the reference's samples are untouched by the extra array
    assert (gi.get_sa(rows[:2000]) == want[:2000]).all()
    assert (gi.export_sa_samples() == want[::1 << level]).all()
    gi.close()


def test_same_sequence_with_and_without_walk_records_and_in_row_order():
    n, level = (1 << 18) + 5, 2
    t = _text(11, n, 4)
    tx = F.Text.with_max_character(t, 4)
    with_w = F.FMIndexWithLocate(tx, level, sampling="text")
    without = F.FMIndexWithLocate(tx, level, sampling="text", walk_records=False)
    row = F.FMIndexWithLocate(tx, level, sampling="row")
    assert with_w.walk_records() and not without.walk_records() and not row.walk_records()
    assert without.text_order() and not row.text_order()
    # walk records cost one byte per row
    assert with_w.heap_size() - without.heap_size() == (n // 112 + 1) * 128
    pats = W.substring_patterns_np(t, 4096, 9, 3)
    sb = with_w.search_many(flat=pats[0], off=pats[1])
    s, e = sb.s, sb.e
    p0, st0 = _locate_steps(with_w, s, e)
    p1, st1 = _locate_steps(without, s, e)
    p2, st2 = _locate_steps(row, s, e)
    assert (p0 == p1).all() and (p0 == p2).all()
    assert st0 == st1 and st0 < st2                                     # half the LF steps of row order
    for ix in (with_w, without, row):
        ix.close()


@pytest.mark.parametrize("maxc,level,sampling,expect", [(7, 2, "text", False), (6, 2, "text", False), (4, 4, "text", False),
                                                        (4, 2, "row", False), (4, 0, "text", False), (255, 2, "text", False),
                                                        (5, 3, "text", True), (1, 1, "text", True)])
def test_eligibility(maxc, level, sampling, expect):
    n = 9000
    t = _text(5 + maxc, n, min(maxc, 200))
    gi = F.FMIndexWithLocate(F.Text.with_max_character(t, maxc), level, sampling=sampling)
    assert gi.walk_records() == expect
    oi = O.OracleIndex(t, maxc, level=level, kind="fm")
    pos, _ = _locate_steps(gi, [0], [n])
    assert (pos == oi.get_sa(np.arange(n)).astype(np.uint64)).all()
    gi.close()


def test_save_load_rebuilds_the_walk_records(tmp_path):
    n, level = 50021, 3
    t = _text(9, n, 4)
    gi = F.FMIndexWithLocate(F.Text.with_max_character(t, 4), level, sampling="text")
    assert gi.walk_records()
    path = os.path.join(tmp_path, "w.fmx")
    gi.save(path)
    # the file does not hold them (they are derived from the records and the phase pieces)
    assert os.path.getsize(path) < gi.heap_size() - (n // 112) * 128 + 4096
    li = F.FMIndexWithLocate.load(path)
    assert li.walk_records() and li.text_order() and li.heap_size() == gi.heap_size()
    p0, s0 = _locate_steps(gi, [0], [n])
    p1, s1 = _locate_steps(li, [0], [n])
    assert (p0 == p1).all() and s0 == s1
    # an index saved WITHOUT them loads without them
    g2 = F.FMIndexWithLocate(F.Text.with_max_character(t, 4), level, sampling="text", walk_records=False)
    p2 = os.path.join(tmp_path, "nw.fmx")
    g2.save(p2)
    l2 = F.FMIndexWithLocate.load(p2)
    assert not l2.walk_records()
    assert (_locate_steps(l2, [0], [n])[0] == p0).all()
    for ix in (gi, li, g2, l2):
        ix.close()


def test_multi_pieces_index_gets_no_walk_records():
    n = 8000
    t = _text(3, n, 4)
    t[np.arange(101, n - 1, 307)] = 0
    gi = F.FMIndexMultiPiecesWithLocate(F.Text.with_max_character(t, 4), 2, sampling="text")
    assert gi.text_order() and not gi.walk_records()
    oi = O.OracleIndex(t, 4, level=2, kind="multi")
    pos, _ = _locate_steps(gi, [0], [n])
    assert (pos == oi.get_sa(np.arange(n)).astype(np.uint64)).all()


@pytest.mark.parametrize("n,sigma,level,m", [((1 << 18) + 77, 4, 2, 4), (200003, 5, 3, 3), (150000, 4, 1, 4), (300000, 2, 2, 9),
                                             (131072, 3, 3, 5)])
def test_long_intervals_take_the_lane_per_walk_kernel(n, sigma, level, m):
    """batches that average 64+ hits per pattern (2^16+ hits in all) on an index with walk records go through
    fmx_locate_walk_lane_kernel -- a lane per walk on consecutive hits, every lane decoding its row's record alone
    (round 4): the oracle's exact sequences, and those of the same text's row-order index (group-cooperative walk)"""
    t = ((W.splitmix64_np(n + sigma, 0, n) % np.uint64(sigma)) + np.uint64(1)).astype(np.uint8)
    t[-1] = 0
    gi = F.FMIndexWithLocate(F.Text.with_max_character(t, sigma), level)
    assert gi.walk_records() and gi.text_order()
    oi = O.OracleIndex(t, sigma, level=level)
    flat, off, _ = W.substring_patterns_np(t, 400, m, 31)
    gb = gi.search_many(flat=flat, off=off)
    os_, oe = oi.count_batch(flat, off, nthreads=8)
    assert (gb.s == os_).all() and (gb.e == oe).all()
    total = int((oe - os_).sum())
    assert total >= (1 << 16) and total // 400 >= 64, total
    goff, gpos = gb.locate()
    ooff, opos = oi.locate_batch(os_, oe, nthreads=8)
    assert (goff == ooff).all() and (gpos == opos).all()
    ri = F.FMIndexWithLocate(F.Text.with_max_character(t, sigma), level, sampling="row")
    _, rpos = ri.search_many(flat=flat, off=off).locate()
    assert (rpos == opos).all()
    # a batch of the same patterns that does NOT reach 64 hits per pattern on average takes the cooperative kernel
    few = gi.search_many(flat=flat[: 8 * m], off=off[:9])
    foff, fpos = few.locate()
    assert (fpos == opos[: int(ooff[8])]).all()
    gi.close(); ri.close()
