"""The one-launch DNA locate (round 5; fmx_locate_f3u_kernel): a block expands its slice of the
hits into LDS itself (no rows array, no fmx_expand_kernel) and picks the walk per 64-hit TICKET -- a lane per hit for
tickets of adjacent rows, the group-cooperative walk for the others.  Whatever the classification the result must be
the reference's positions in the reference's order (wrapper.rs:203-217, fm_index.rs:127-140): mixed and skewed batches,
empty patterns between the hits, more than 2^20 patterns (two rounds of the bracket search), offsets with gaps, and the
same batches with the classification forced either way and through the round-4 pair of launches (measurement build)."""
import os
import subprocess
import sys

import numpy as np
import pytest

import fm_index_amd as F
from fm_index_amd import workload as W
from oracle import fm_oracle as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _text(seed, n, alpha=4):
    t = (W.splitmix64_np(seed, 0, n) % np.uint64(alpha)).astype(np.uint8) + 1
    t[-1] = 0
    return t


def _index(n, seed=7, level=2):
    t = _text(seed, n)
    gi = F.FMIndexWithLocate(F.Text.with_max_character(t, 4), level)
    assert gi.walk_records() and gi.text_order()
    oi = O.OracleIndex(t, 4, level=level)
    return t, gi, oi


def _mixed_intervals(n, rng, singles, longs, long_len, empties=0):
    """(s, e) of a batch: `singles` one-row intervals at random rows, `longs` intervals of ~long_len rows, `empties` empty
    ones, shuffled"""
    s1 = rng.integers(0, n, singles).astype(np.uint64)
    e1 = s1 + np.uint64(1)
    s2 = rng.integers(0, max(n - long_len, 1), longs).astype(np.uint64)
    e2 = np.minimum(s2 + rng.integers(long_len // 2, long_len + 1, longs).astype(np.uint64), np.uint64(n))
    s3 = rng.integers(0, n, empties).astype(np.uint64)
    s, e = np.concatenate([s1, s2, s3]), np.concatenate([e1, e2, s3])
    p = rng.permutation(len(s))
    return s[p], e[p]


def _expect(want, s, e):
    return np.concatenate([want[int(a):int(b)] for a, b in zip(s, e)] + [np.zeros(0, np.uint64)])


@pytest.mark.parametrize("singles,longs,long_len,empties", [
    (70000, 40, 5000, 0),          # singletons with long intervals in between: both classes in most slices
    (100, 3, 120000, 5),           # almost everything adjacent
    (200000, 0, 0, 0),             # no adjacent ticket at all
    (30000, 2000, 70, 3000),       # intervals just above a ticket, many empty patterns
    (5000, 300, 33, 100),          # ranges of 17..33 rows: the long-range queue of the expansion and its thread-written cousin
    (9, 1, 200, 0), (1, 0, 0, 0), (0, 1, 64, 0), (0, 1, 65, 3)])
def test_mixed_batches_equal_the_oracle(singles, longs, long_len, empties):
    n = (1 << 17) + 311
    _, gi, oi = _index(n)
    want = oi.get_sa(np.arange(n)).astype(np.uint64)
    rng = np.random.default_rng(singles + longs + long_len)
    s, e = _mixed_intervals(n, rng, singles, longs, long_len, empties)
    off, pos = gi.locate_many(s, e)
    assert int(off[-1]) == int((e - s).sum()) and (np.asarray(pos, np.uint64) == _expect(want, s, e)).all()
    # step census: a walk is SA[row] mod 2^level steps whichever kernel path took it
    gi._lib.fmx_set_timing(gi.handle(), 1)
    gi.locate_many(s, e)
    assert int(gi._lib.fmx_last_steps(gi.handle())) == int((_expect(want, s, e) & np.uint64(3)).sum())
    gi._lib.fmx_set_timing(gi.handle(), 0)
    gi.close()


@pytest.mark.parametrize("longs,long_len", [(0, 0), (6, 3000)])
def test_partial_last_ticket_of_scattered_batches(longs, long_len):
    """>= 3e5 scattered singletons (tickets of 64 hits, Q = 4, the ring) whose total is NOT a multiple of
    64 -- remainders below and above the 32 walk slots of a wave -- so that the last slice holds a partial ticket among
    more than 16 non-adjacent ones.  The ticket list fills in the order the waves' atomics land; the partial ticket must
    still be the last one drawn or a wave retires its slots on the ticket's missing tail and drops the ticket behind it.
    Repeated launches (the order is a race), with and without long intervals in the last slice."""
    n = (1 << 17) + 311
    _, gi, oi = _index(n)
    want = oi.get_sa(np.arange(n)).astype(np.uint64)
    rng = np.random.default_rng(longs + 11)
    for rem in (1, 9, 31, 32, 33, 63):
        s, e = _mixed_intervals(n, rng, 300000, longs, long_len)
        extra = (rem - int((e - s).sum())) % 64
        xs = rng.integers(0, n, extra).astype(np.uint64)
        s, e = np.concatenate([s, xs]), np.concatenate([e, xs + np.uint64(1)])
        assert int((e - s).sum()) % 64 == rem
        for _ in range(4):
            # another order every time: the device scratch the positions land in still holds the previous launch's
            # (correct) positions, and an unwritten slot must not read as a right answer
            p = rng.permutation(len(s))
            s, e = s[p], e[p]
            off, pos = gi.locate_many(s, e)
            assert (np.asarray(pos, np.uint64) == _expect(want, s, e)).all(), rem
    gi.close()


def test_more_than_2_20_patterns_mostly_empty():
    """1.3 M patterns (two rounds of the 1024-probe bracket search), almost all of them empty, the hits in clusters: a
    slice's first pattern sits thousands of patterns behind the previous slice's last"""
    n = 1 << 16
    _, gi, oi = _index(n, seed=9)
    want = oi.get_sa(np.arange(n)).astype(np.uint64)
    npat = (1 << 20) + (1 << 18) + 77
    rng = np.random.default_rng(5)
    s = rng.integers(0, n, npat).astype(np.uint64)
    e = s.copy()
    hot = np.sort(rng.choice(npat, 3000, replace=False))
    e[hot] = np.minimum(s[hot] + rng.integers(1, 90, len(hot)).astype(np.uint64), np.uint64(n))
    e[hot[:3]] = np.minimum(s[hot[:3]] + np.uint64(9000), np.uint64(n))
    off, pos = gi.locate_many(s, e)
    assert (np.asarray(pos, np.uint64) == _expect(want, s[hot], e[hot])).all()
    gi.close()


def test_offsets_with_gaps_and_foreign_ranges_are_reported_and_stay_inside_the_index():
    n = 70001
    _, gi, oi = _index(n, seed=3)
    want = oi.get_sa(np.arange(n)).astype(np.uint64)
    lib = gi._lib
    k = 3000
    rng = np.random.default_rng(1)
    s = rng.integers(0, n - 50, k).astype(np.uint64)
    e = s + rng.integers(0, 40, k).astype(np.uint64)
    cnt = (e - s).astype(np.uint64)
    off = np.zeros(k + 1, np.uint64)
    off[1:] = np.cumsum(cnt + (np.arange(k) % 7 == 0).astype(np.uint64) * np.uint64(3))   # a 3-slot gap behind every 7th range
    off += np.uint64(5)                                                                     # and one before the first
    total = int(off[-1])
    pos = np.full(total, 2 ** 64 - 1, np.uint64)
    assert lib.fmx_locate_batch(gi.handle(), F._p(s), F._p(e), k, F._p(off), F._p(pos)) == F._lib.ERR_ARG
    assert (pos < n).all()                                    # gap slots hold a position of this text (row 0's)
    for j in (0, 1, 6, 7, 8, k - 1):
        assert (pos[int(off[j]):int(off[j]) + int(cnt[j])] == want[int(s[j]):int(e[j])]).all(), j
    # a range beyond the index: reported, its slots filled from row 0 on; the others located as usual
    e2 = e.copy()
    e2[10] = np.uint64(n + 1000)
    off2 = np.zeros(k + 1, np.uint64)
    off2[1:] = np.cumsum((e2 - s).astype(np.uint64))
    pos2 = np.full(int(off2[-1]), 2 ** 64 - 1, np.uint64)
    assert lib.fmx_locate_batch(gi.handle(), F._p(s), F._p(e2), k, F._p(off2), F._p(pos2)) == F._lib.ERR_ARG
    assert (pos2 < n).all() and (pos2[int(off2[11]):int(off2[12])] == want[int(s[11]):int(e2[11])]).all()
    # and the handle keeps working
    o3, p3 = gi.locate_many(s, e)
    assert (np.asarray(p3, np.uint64) == _expect(want, s, e)).all()
    gi.close()


_FORCED = r"""
import sys, numpy as np
sys.path.insert(0, %r)
import fm_index_amd as F
from fm_index_amd import workload as W
n = (1 << 17) + 311
t = (W.splitmix64_np(7, 0, n) %% np.uint64(4)).astype(np.uint8) + 1
t[-1] = 0
gi = F.FMIndexWithLocate(F.Text.with_max_character(t, 4), 2)
d = np.load(sys.argv[1])
off, pos = gi.locate_many(d["s"], d["e"])
np.save(sys.argv[2], np.asarray(pos, np.uint64))
"""


@pytest.mark.parametrize("env", [{"FMX_ADJ_CLUSTERS": "0"}, {"FMX_ADJ_CLUSTERS": "65"}, {"FMX_ADJ_CLUSTERS": "1"},
                                 {"FMX_VARIANT": "28"}])
def test_any_classification_gives_the_same_positions(env, tmp_path):
    """measurement build: every ticket through the cooperative walk (0), every ticket a lane per hit (65), only perfect
    runs a lane per hit (1), and the round-4 pair of launches (FMX_VARIANT=28) -- the shipped library's positions"""
    lib = os.path.join(ROOT, "fm_index_amd", "libfmx_measure.so")
    assert os.path.exists(lib), "fm_index_amd/libfmx_measure.so is missing: run `make -C fm_index_amd/csrc measure`"
    n = (1 << 17) + 311
    _, gi, oi = _index(n)
    want = oi.get_sa(np.arange(n)).astype(np.uint64)
    rng = np.random.default_rng(77)
    s, e = _mixed_intervals(n, rng, 90000, 60, 4000, 500)
    gi.close()
    np.savez(tmp_path / "in.npz", s=s, e=e)
    p = subprocess.run([sys.executable, "-c", _FORCED % ROOT, str(tmp_path / "in.npz"), str(tmp_path / "out.npy")],
                       env=dict(os.environ, FMX_LIB=lib, **env), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert p.returncode == 0, p.stderr.decode(errors="replace")[-2000:]
    assert (np.load(tmp_path / "out.npy") == _expect(want, s, e)).all()


def test_workspace_form_and_two_streams_share_nothing():
    """fmx_locate_batch_ws_dev on an index with walk records: the workspace is not needed any more (the rows live in LDS)
    but the entry point keeps its contract; two streams, same positions"""
    import ctypes as C
    import torch
    n = 1 << 17
    t, gi, oi = _index(n, seed=21)
    lib = gi._lib
    flat, off, _ = W.substring_patterns_np(t, 20000, 7, 5)
    b = gi.search_many(flat=flat, off=off)
    ooff, opos = oi.locate_batch(b.s, b.e, nthreads=8)
    dev = torch.device("cuda", 0)
    d_s = torch.from_numpy(b.s.astype(np.int64)).to(dev)
    d_e = torch.from_numpy(b.e.astype(np.int64)).to(dev)
    d_off = torch.from_numpy(np.asarray(ooff).astype(np.int64)).to(dev)
    total = int(ooff[-1])
    wsb = int(lib.fmx_locate_workspace_bytes(gi.handle(), total))
    outs = []
    for _ in range(2):
        st = torch.cuda.Stream(device=dev)
        ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device=dev)
        pos = torch.empty(total, dtype=torch.int64, device=dev)
        rc = lib.fmx_locate_batch_ws_dev(gi.handle(), C.c_void_p(d_s.data_ptr()), C.c_void_p(d_e.data_ptr()), len(b.s),
                                         C.c_void_p(d_off.data_ptr()), total, C.c_void_p(pos.data_ptr()),
                                         C.c_void_p(ws.data_ptr()), wsb, C.c_void_p(st.cuda_stream))
        assert rc == 0
        outs.append((st, pos, ws))
    torch.cuda.synchronize()
    for _, pos, _ in outs:
        assert (pos.cpu().numpy().astype(np.uint64) == np.asarray(opos, np.uint64)).all()
    gi.close()


@pytest.mark.parametrize("sampling,level", [(None, 2), ("row", 2), (None, 3), ("row", 0)])
@pytest.mark.parametrize("singles,longs,long_len", [(60000, 60, 6000), (100000, 0, 0), (0, 8, 70000), (150000, 3, 90000)])
def test_rlfm_run_table_mixed_batches(sampling, level, singles, longs, long_len):
    """RLFM with the run table (round 5): batches of 2^18+ hits that average two or more hits per pattern take
    fmx_locate_rl_rounds_kernel (a lane per walk on consecutive hits, four tickets per wave in rounds) for every hit -- mixed
    and skewed batches included; batches of about one hit per pattern take the queue kernel.  The oracle's ordered
    positions either way, in text and in row order."""
    n = 300000
    t = W.repetitive_text_np(n, 11, base_len=512, mut_per_1024=4)
    gi = F.RLFMIndexWithLocate(F.Text(t), level, sampling=sampling)
    assert gi.walk_records()
    oi = O.OracleIndex(t, 255, level=level, kind="rlfm")
    want = oi.get_sa(np.arange(n)).astype(np.uint64)
    rng = np.random.default_rng(singles + longs)
    s, e = _mixed_intervals(n, rng, singles + 300000 * int(longs == 0), longs, long_len, 100)
    off, pos = gi.locate_many(s, e)
    assert int(off[-1]) >= (1 << 18)
    assert (np.asarray(pos, np.uint64) == _expect(want, s, e)).all()
    gi.close()


@pytest.mark.parametrize("shape", ["slices", "deferred"])
@pytest.mark.parametrize("kind", ["rlfm", "fm_bytes", "dna_row_order"])
def test_rows_of_large_batches_are_expanded_in_consecutive_slices(kind, shape):
    """The paths that keep a rows array expand it with fmx_expand_slices_kernel from 1024 hits per pattern on: at most
    2048 blocks, each a run of consecutive 4096-hit slices -- only a block's first slice probes off[] for its first
    pattern, the others start from the pattern the previous slice ended on, and slices inside one long range follow from
    it.  ~10^7 hits (2400+ slices): long intervals with clusters of singletons and empty patterns in between, so that a
    slice's first pattern sits anywhere from 0 to thousands of patterns behind the hint."""
    n = 250000
    rng = np.random.default_rng(17)
    if kind == "rlfm":
        t = W.repetitive_text_np(n, 13, base_len=400, mut_per_1024=6)
        gi = F.RLFMIndexWithLocate(F.Text(t), 2)
        oi = O.OracleIndex(t, 255, level=2, kind="rlfm")
    elif kind == "fm_bytes":
        t = W.byte_text_np(n, 8)
        gi = F.FMIndexWithLocate(F.Text(t), 2)
        oi = O.OracleIndex(t, 255, level=2)
    else:
        t = _text(23, n)
        gi = F.FMIndexWithLocate(F.Text.with_max_character(t, 4), 2, sampling="row")
        oi = O.OracleIndex(t, 4, level=2)
    want = oi.get_sa(np.arange(n)).astype(np.uint64)
    if shape == "slices":
        s, e = _mixed_intervals(n, rng, 6000, 62, 240000, 2000)     # ~1500 hits per pattern on average
    else:
        # round 6, per RANGE: a low batch average (~40 hits per pattern: the lane-per-pattern expansion) with a few
        # ranges of 2^16+ rows in it -- those are listed and written by a second, grid-wide pass (fmx_expand_long_kernel),
        # ranges just below the threshold by their pattern's wave
        s, e = _mixed_intervals(n, rng, 50000, 9, 240000, 3000)
        s2, e2 = _mixed_intervals(n, rng, 0, 4, 65000, 0)
        s, e = np.concatenate([s, s2]), np.concatenate([e, e2])
        p = rng.permutation(len(s))
        s, e = s[p], e[p]
        assert ((e - s) >= (1 << 16)).sum() >= 5 and int((e - s).sum()) // len(s) < 1024
    off, pos = gi.locate_many(s, e)
    assert int(off[-1]) > (2100 * 4096 if shape == "slices" else 1 << 20)
    exp = _expect(want, s, e)
    assert (np.asarray(pos, np.uint64) == exp).all()
    gi.close()
