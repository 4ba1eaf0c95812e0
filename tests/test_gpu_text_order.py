"""Text-order suffix-array sampling: indexes whose LF step costs several
requests (RLFM; FM / multi-pieces over two or more wavelet levels) sample the rows whose SA value is a
multiple of 2^level.  Same answers as the reference's row-order sampling (suffix_array/sample.rs:22-60)
on every row, the samples the reference would hold are still exported, and a walk is SA[row] mod
2^level LF steps -- never more than 2^level - 1."""
import os
import subprocess
import sys

import numpy as np
import pytest

import fm_index_amd as F
from fm_index_amd import workload as W
from oracle import fm_oracle as O

pytestmark = pytest.mark.gpu


def _text(seed, n, alpha, base=1):
    t = (W.splitmix64_np(seed, 0, n) % np.uint64(alpha)).astype(np.uint8) + base
    t[-1] = 0
    return t


def _locate_steps(idx, s, e):
    lib = idx._lib
    lib.fmx_set_timing(idx.handle(), 1)
    _, pos = idx.locate_many(s, e)
    steps = int(lib.fmx_last_steps(idx.handle()))
    lib.fmx_set_timing(idx.handle(), 0)
    return pos, steps


CASES = [("rlfm", 4, 4), ("rlfm", 255, 200), ("fm", 255, 200), ("fm", 255, 17), ("multi", 255, 60)]


@pytest.mark.parametrize("kind,maxc,alpha", CASES)
@pytest.mark.parametrize("level", [0, 1, 2, 3, 4, 5, 7])
def test_every_row_and_step_bound(kind, maxc, alpha, level):
    n = 6000 + 37 * level
    t = _text(900 + level + alpha, n, alpha)
    if kind == "multi":
        t[np.arange(97, n - 1, 211)] = 0                          # several pieces
    cls = {"rlfm": F.RLFMIndexWithLocate, "fm": F.FMIndexWithLocate, "multi": F.FMIndexMultiPiecesWithLocate}[kind]
    gi = cls(F.Text.with_max_character(t, maxc), level)
    oi = O.OracleIndex(t, maxc, level=level, kind=kind)
    rows = np.arange(n)
    assert (gi.get_sa(rows) == oi.get_sa(rows)).all()             # every row, through the scalar kernel
    # the whole index as one interval through the batched walk; the steps it took
    pos, steps = _locate_steps(gi, np.array([0], np.uint64), np.array([n], np.uint64))
    assert (np.asarray(pos, np.uint64) == oi.get_sa(rows).astype(np.uint64)).all()
    lv = level                                                     # n > 2^level: the level is kept (sample.rs:28-31)
    assert gi.level() == lv
    if 1 <= lv <= 4:
        # text order: the walk from row i is SA[i] mod 2^level steps exactly
        sa = oi.get_sa(rows).astype(np.uint64)
        assert steps == int((sa & np.uint64((1 << lv) - 1)).sum())
    # the reference's samples (row-order, sample.rs:33-43) are what the export returns either way
    want = oi.get_sa(np.arange(0, n, 1 << lv))
    assert (gi.export_sa_samples() == want).all()


@pytest.mark.parametrize("kind,maxc,alpha", [("rlfm", 4, 4), ("fm", 255, 90)])
def test_row_order_dna_walks_are_geometric_and_text_order_steps_halve(kind, maxc, alpha):
    """Row-order sampling (the reference's; FMX_FLAG_ROW_ORDER on a DNA index, whose default became text order with
    walk records in round 4) walks a geometric number of steps, mean 2^level - 1; text order walks SA[row] mod
    2^level steps, half that mean."""
    n, level = 1 << 16, 2
    t = _text(77, n, alpha)
    cls = {"rlfm": F.RLFMIndexWithLocate, "fm": F.FMIndexWithLocate}[kind]
    gi = cls(F.Text.with_max_character(t, maxc), level)
    _, steps = _locate_steps(gi, np.array([0], np.uint64), np.array([n], np.uint64))
    assert steps <= 3 * n                                          # at most 2^level - 1 per row
    assert abs(steps / n - 1.5) < 0.05                             # uniform phases: mean 1.5
    dna = F.FMIndexWithLocate(F.Text.with_max_character(_text(78, n, 4), 4), level, sampling="row")
    assert not dna.text_order()
    _, steps_dna = _locate_steps(dna, np.array([0], np.uint64), np.array([n], np.uint64))
    assert steps_dna > 2 * n                                       # geometric, mean 3 (row-order sampling)
    # the default DNA index: text order + walk records (2.75 requests per hit at level 2)
    dflt = F.FMIndexWithLocate(F.Text.with_max_character(_text(78, n, 4), 4), level)
    assert dflt.text_order() and dflt.walk_records()
    _, steps_dflt = _locate_steps(dflt, np.array([0], np.uint64), np.array([n], np.uint64))
    assert abs(steps_dflt / n - 1.5) < 0.05


@pytest.mark.parametrize("kind", ["rlfm", "fm"])
def test_save_load_keeps_text_order(kind, tmp_path):
    n = 20000
    t = _text(5, n, 150)
    cls = {"rlfm": F.RLFMIndexWithLocate, "fm": F.FMIndexWithLocate}[kind]
    gi = cls(F.Text(t), 3)
    path = os.path.join(tmp_path, "ix.fmx")
    gi.save(path)
    li = cls.load(path)
    rows = np.arange(n)
    assert (li.get_sa(rows) == gi.get_sa(rows)).all()
    p0, s0 = _locate_steps(gi, np.array([0], np.uint64), np.array([n], np.uint64))
    p1, s1 = _locate_steps(li, np.array([0], np.uint64), np.array([n], np.uint64))
    assert (np.asarray(p0) == np.asarray(p1)).all() and s0 == s1 and s0 <= 7 * n
    assert li.heap_size() == gi.heap_size()


def test_text_order_forced_on_one_level_indexes():
    """FMX_FLAG_TEXT_ORDER (opt-in, shipped library): one-level indexes sampled in text order -- the text-order
    branches of the DNA walk kernel and of the generic walk against the oracle, levels 1-4, FM and multi-pieces."""
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import text_order_forced
    assert text_order_forced.main() == 12


def test_row_order_can_be_forced_where_text_order_is_the_default():
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import text_order_forced
    assert text_order_forced.row_order_forced()
