"""Locate batches of millions of hits take other launch shapes than the BASELINE's 2^20-hit batch
(DESIGN.md 4.1 / 4.1b): the DNA walk kernel runs two 1024-thread blocks per CU from 2^22 hits, the
one-walk-per-lane kernel 640-thread blocks.  Same ordered positions as the oracle
(wrapper.rs:203-217 order, fm_index.rs:127-140 / rlfmi.rs:172-190 values)."""
import numpy as np
import pytest

import fm_index_amd as F
from fm_index_amd import workload as W
from oracle import fm_oracle as O

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("level", [1, 2, 4])
def test_dna_locate_millions_of_hits(level):
    n = 1 << 16
    t = (W.splitmix64_np(61 + level, 0, n) % np.uint64(4)).astype(np.uint8) + 1
    t[-1] = 0
    gi = F.FMIndexWithLocate(F.Text.with_max_character(t, 4), level)
    oi = O.OracleIndex(t, 4, level=level)
    # 1200 patterns of 2 symbols: ~4 000 hits each -> ~4.9e6 hits, every row located ~75 times
    flat, off, _ = W.substring_patterns_np(t, 1200, 2, 17)
    gb = gi.search_many(flat=flat, off=off)
    os_, oe = oi.count_batch(flat, off)
    assert (gb.s == os_).all() and (gb.e == oe).all()
    total = int((oe - os_).sum())
    assert total >= (1 << 22), total
    goff, gpos = gb.locate()
    ooff, opos = oi.locate_batch(os_, oe, nthreads=8)
    assert (goff == ooff).all() and (gpos == opos).all()
    # the same rows as ONE interval list with empty and one-row intervals mixed in
    s = np.concatenate([os_[:50], os_[:50], np.arange(50, dtype=np.uint64)])
    e = np.concatenate([oe[:50], os_[:50], np.arange(50, dtype=np.uint64) + 1])
    goff, gpos = gi.locate_many(s, e)
    ooff, opos = oi.locate_batch(s, e, nthreads=8)
    assert (goff == ooff).all() and (gpos == opos).all()


def test_rlfm_locate_millions_of_hits():
    n = 1 << 16
    t = W.repetitive_text_np(n, 7, base_len=1 << 9, mut_per_1024=20)
    gi = F.RLFMIndexWithLocate(F.Text(t), 2)
    oi = O.OracleIndex(t, 255, level=2, kind="rlfm")
    flat, off, _ = W.substring_patterns_np(t, 40000, 4, 23)
    gb = gi.search_many(flat=flat, off=off)
    os_, oe = oi.count_batch(flat, off)
    assert (gb.s == os_).all() and (gb.e == oe).all()
    total = int((oe - os_).sum())
    assert total >= (1 << 22), total
    goff, gpos = gb.locate()
    ooff, opos = oi.locate_batch(os_, oe, nthreads=8)
    assert (goff == ooff).all() and (gpos == opos).all()


@pytest.mark.parametrize("extra", [0, 37, 4099])
def test_dna_locate_just_above_the_write_combining_threshold(extra):
    """From 64 hits per wave on (>= 2^18 hits or so) the DNA walk kernel hands out whole 64-hit tickets and
    sends the positions through its LDS write-combining ring; a batch right at that size, with a ragged
    last ticket, must give the same ordered positions (every row of the index, then `extra` more rows)."""
    n = 1 << 18
    t = (W.splitmix64_np(71, 0, n) % np.uint64(4)).astype(np.uint8) + 1
    t[-1] = 0
    gi = F.FMIndexWithLocate(F.Text.with_max_character(t, 4), 2)
    oi = O.OracleIndex(t, 4, level=2)
    s = np.array([0, 1000], dtype=np.uint64)
    e = np.array([n, 1000 + extra], dtype=np.uint64)
    goff, gpos = gi.locate_many(s, e)
    want = oi.get_sa(np.concatenate([np.arange(n), np.arange(1000, 1000 + extra)]))
    assert int(goff[-1]) == n + extra and (gpos == want.astype(np.uint64)).all()
