"""BASELINE config 5 behind the C ABI: fmx_replicate + fmx_count_batch_multi / fmx_locate_batch_multi /
fmx_count_batch_multi_resident.  A batch shards contiguously over G replicas of an index (pattern k of N -> replica
floor(k G / N), wrapper.rs:103-124 reads only immutable index state) and every shard's results land in place in the
caller's arrays: G handles on device 0 (1, 2, 3 -- ragged shards) must give the one-handle results bit for bit, the
oracle's, and the committed oracle-made golden hashes (tests/golden/config5_counts.json)."""
import ctypes as C
import json
import os
import subprocess

import numpy as np
import pytest

import fm_index_amd as F
from fm_index_amd import _lib as L
from fm_index_amd import workload as W
from benchmarks.legs.common import counts_sha256, golden_key, positions_sha256, ranges_sha256
from oracle import fm_oracle as O
from test_abi_cpu import build_c_example

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _golden(key):
    with open(os.path.join(ROOT, "tests", "golden", "config5_counts.json")) as f:
        return json.load(f)["entries"][key]


def _config5_patterns(t, total, m, seed=7):
    n = len(t)
    src = (W.splitmix64_np(seed, 0, total) % np.uint64(n - 1 - m)).astype(np.int64)
    flat = t[src[:, None] + np.arange(m, dtype=np.int64)[None, :]].reshape(-1)
    return np.ascontiguousarray(flat), np.arange(total + 1, dtype=np.uint64) * np.uint64(m)


@pytest.mark.parametrize("g", [1, 2, 3])
@pytest.mark.parametrize("total", [8192, 8193, 65536])
def test_config5_small_golden_through_g_replicas(g, total):
    """the committed golden sets at n = 2^16 (seed 7; 8193 = ragged at G = 2, 65536 ragged at G = 3; 65536 patterns take
    the chunked batch path): counts, (s, e) and the ordered positions hash to what the CPU oracle computed over all
    patterns"""
    n, m = 1 << 16, 32
    t = W.dna_text_np(n, 1)
    gi = F.FMIndexWithLocate(F.Text.with_max_character(t, 4), 2)
    reps = F.Replicas.of(gi, [0] * (g - 1))
    assert len(reps) == g and all(r.device() == 0 for r in reps.indexes)
    flat, off = _config5_patterns(t, total, m)
    b = reps.search_many(flat=flat, off=off)
    ent = _golden(golden_key("dna", 16, 7, total, m))
    assert counts_sha256(b.counts) == ent["counts_sha256"] and ranges_sha256(b.s, b.e) == ent["ranges_sha256"]
    one = gi.search_many(flat=flat, off=off)
    assert (one.s == b.s).all() and (one.e == b.e).all() and (one.counts == b.counts).all()
    hoff, pos = reps.locate_many(b.s, b.e)
    assert int(hoff[-1]) == ent["locate"]["hits"] and positions_sha256(pos) == ent["locate"]["positions_sha256"]
    cuts = [reps.shard_range(total, r) for r in range(g)]
    assert cuts[0][0] == 0 and cuts[-1][1] == total and all(cuts[r][1] == cuts[r + 1][0] for r in range(g - 1))
    assert cuts == [((total * r + g - 1) // g, (total * (r + 1) + g - 1) // g) for r in range(g)]   # sharding.shard_range
    reps.close()


def _kinds():
    n = 70001
    return {
        "dna_walk_records": lambda: (W.dna_text_np(n, 3), 4, lambda t: F.FMIndexWithLocate(F.Text.with_max_character(t, 4), 2), "fm"),
        "dna_auto": lambda: (W.dna_text_np(1 << 24, 5), 4,
                             lambda t: F.FMIndexWithLocate(F.Text.with_max_character(t, 4), 2, auto=True), "fm"),
        "bytes_fm": lambda: (W.byte_text_np(n, 4), 255, lambda t: F.FMIndexWithLocate(F.Text(t), 3), "fm"),
        "rlfm_run_table": lambda: (W.repetitive_text_np(n, 11, base_len=512, mut_per_1024=4), 255,
                                   lambda t: F.RLFMIndexWithLocate(F.Text(t), 2), "rlfm"),
        "rlfm_count_only": lambda: (W.byte_text_np(n, 6), 255, lambda t: F.RLFMIndex(F.Text(t)), "rlfm"),
        "wide_dna": lambda: (W.dna_text_np(n, 8), 4,
                             lambda t: F.FMIndexWithLocate(F.Text.with_max_character(t, 4), 2, force_wide=True), "fm"),
        "wide_rlfm": lambda: (W.repetitive_text_np(n, 12, base_len=300, mut_per_1024=5), 255,
                              lambda t: F.RLFMIndexWithLocate(F.Text(t), 2, force_wide=True), "rlfm"),
    }


@pytest.mark.parametrize("kind", list(_kinds()))
def test_a_replica_is_the_same_index(kind):
    """fmx_replicate copies every array the query path reads -- accelerators, walk records, run table, select
    structures, superblock bases of the wide engine: the replica answers like the source (and like the oracle) after the
    source is gone"""
    t, maxc, make, okind = _kinds()[kind]()
    gi = make(t)
    n = len(t)
    rp = gi.replicate()
    lib = gi._lib
    for f in ("fmx_len", "fmx_kind", "fmx_level", "fmx_max_character", "fmx_sym_bytes", "fmx_is_wide", "fmx_text_order",
              "fmx_walk_records", "fmx_kmer_k", "fmx_has_pair_index", "fmx_num_runs", "fmx_num_samples"):
        assert getattr(lib, f)(gi.handle()) == getattr(lib, f)(rp.handle()), f
    assert 0.9 * gi.heap_size() <= rp.heap_size() <= 1.1 * gi.heap_size() + 4096
    bwt = gi.export_bwt()
    gi.close()                                                  # the replica owns its arrays
    assert (rp.export_bwt() == bwt).all()
    small = n <= 100000
    oi = O.OracleIndex(t, maxc, level=rp.level(), kind=okind) if small else None
    flat, off = W.ragged_patterns_np(5000, 9, min(maxc, 6), 21)
    sflat, soff, _ = W.substring_patterns_np(t, 3000, 14, 22)
    for fl, of in ((flat, off), (sflat, soff)):
        b = rp.search_many(flat=fl, off=of)
        if oi is not None:
            so, eo = oi.count_batch(fl, of, nthreads=8)
            assert (so == b.s).all() and (eo == b.e).all()
        else:
            assert (b.counts[np.diff(of) >= 14] >= 1).all()
    rows = (W.splitmix64_np(5, 0, 2000) % np.uint64(n)).astype(np.uint64)
    if oi is not None:
        assert (rp.lf_map(rows) == oi.lf_map(rows)).all() and (rp.get_l(rows) == oi.get_l(rows)).all()
        if rp.level() is not None:
            b = rp.search_many(flat=sflat, off=soff)
            hoff, pos = rp.locate_many(b.s, b.e)
            ooff, opos = oi.locate_batch(b.s, b.e, nthreads=8)
            assert (np.asarray(opos, np.uint64) == pos).all()
        oi.close()
    else:                                                        # n = 2^24: the accelerated index against its plain twin
        assert rp.has_pair_index() and rp.kmer_k() >= 8      # (k = 10 at n = 2^24: the table stays below n / 2 bytes)
        plain = F.FMIndexWithLocate(F.Text.with_max_character(t, 4), 2)
        rf, ro = W.random_patterns_np(20000, 32, 4, 9)
        for fl, of in ((sflat, soff), (rf, ro)):
            a, b = plain.search_many(flat=fl, off=of), rp.search_many(flat=fl, off=of)
            assert (a.s == b.s).all() and (a.e == b.e).all()
        plain.close()
    rp.close()


@pytest.mark.parametrize("g", [2, 3, 5])
def test_sharded_batches_equal_one_handle_on_every_path(g):
    """ragged and empty patterns, refinement pairs, 2-byte offsets that do not start at 0, page-locked arrays (the chunk
    pipeline with results written over the host link), Character = u64 patterns: same (s, e, count) as one handle"""
    import torch
    n = 150001
    t = W.dna_text_np(n, 13)
    gi = F.FMIndexWithLocate(F.Text.with_max_character(t, 4), 2)
    oi = O.OracleIndex(t, 4, level=2)
    reps = F.Replicas.of(gi, [0] * (g - 1))
    lib = gi._lib
    hs = reps._handles()
    # (a) small ragged batch with empty patterns + refinement from a first search
    flat, off = W.ragged_patterns_np(1001, 6, 4, 31)
    first = reps.search_many(flat=flat, off=off)
    so, eo = oi.count_batch(flat, off, nthreads=8)
    assert (so == first.s).all() and (eo == first.e).all()
    se = np.stack([first.s, first.e], axis=1).reshape(-1).copy()
    flat2, off2 = W.ragged_patterns_np(1001, 4, 4, 32)
    second = reps.search_many(flat=flat2, off=off2, s0e0=se)
    ref = gi.search_many(flat=flat2, off=off2, s0e0=se)
    assert (second.s == ref.s).all() and (second.e == ref.e).all() and (second.counts == ref.counts).all()
    # (b) 200 003 patterns: every shard takes the batch path; offsets of the whole buffer, shards start mid-buffer
    flat, off, _ = W.substring_patterns_np(t, 200003, 20, 33)
    big = reps.search_many(flat=flat, off=off)
    so, eo = oi.count_batch(flat, off, nthreads=8)
    assert (so == big.s).all() and (eo == big.e).all() and (big.counts == eo - so).all()
    # (c) page-locked arrays: the chunk pipeline of every shard writes straight into the caller's arrays
    npat = len(off) - 1
    pf = torch.from_numpy(flat).pin_memory()
    po = torch.from_numpy(off.astype(np.int64)).pin_memory()
    ps, pe, pc = (torch.zeros(npat, dtype=torch.int64).pin_memory() for _ in range(3))
    rc = lib.fmx_count_batch_multi(hs, g, C.c_void_p(pf.data_ptr()), C.c_void_p(po.data_ptr()), npat, None,
                                   C.c_void_p(ps.data_ptr()), C.c_void_p(pe.data_ptr()), C.c_void_p(pc.data_ptr()))
    assert rc == 0, lib.fmx_last_error()
    assert (ps.numpy().view(np.uint64) == so).all() and (pe.numpy().view(np.uint64) == eo).all()
    assert (pc.numpy().view(np.uint64) == eo - so).all()
    # (d) patterns resident on the device, results into pinned and into pageable host arrays
    dev = torch.device("cuda", 0)
    keep, d_pat, d_off = [], (C.c_void_p * g)(), (C.c_void_p * g)()
    for r in range(g):
        a, b = reps.shard_range(npat, r)
        fp = torch.from_numpy(flat[int(off[a]):int(off[b])].copy()).to(dev)
        fo = torch.from_numpy((off[a:b + 1] - off[a]).astype(np.int64)).to(dev)
        keep += [fp, fo]
        d_pat[r], d_off[r] = fp.data_ptr(), fo.data_ptr()
    for pinned in (True, False):
        outs = [torch.zeros(npat, dtype=torch.int64) for _ in range(3)]
        if pinned:
            outs = [o.pin_memory() for o in outs]
        rc = lib.fmx_count_batch_multi_resident(hs, g, d_pat, d_off, npat, None, *[C.c_void_p(o.data_ptr()) for o in outs])
        assert rc == 0, lib.fmx_last_error()
        assert (outs[0].numpy().view(np.uint64) == so).all() and (outs[1].numpy().view(np.uint64) == eo).all()
        assert (outs[2].numpy().view(np.uint64) == eo - so).all()
    # counts only (the other two arrays NULL)
    oc = torch.zeros(npat, dtype=torch.int64).pin_memory()
    assert lib.fmx_count_batch_multi_resident(hs, g, d_pat, d_off, npat, None, None, None, C.c_void_p(oc.data_ptr())) == 0
    assert (oc.numpy().view(np.uint64) == eo - so).all()
    # (e) locate of a mixed batch: long intervals, singletons, empty ranges
    rng = np.random.default_rng(g)
    s1 = rng.integers(0, n, 30000).astype(np.uint64)
    ls = rng.integers(0, n - 9000, 7).astype(np.uint64)
    s = np.concatenate([s1, ls, s1[:50]])
    e = np.concatenate([s1 + np.uint64(1), ls + np.uint64(9000), s1[:50]])
    p = rng.permutation(len(s))
    s, e = s[p], e[p]
    hoff, pos = reps.locate_many(s, e)
    ooff, opos = oi.locate_batch(s, e, nthreads=8)
    assert (hoff == np.asarray(ooff, np.uint64)).all() and (pos == np.asarray(opos, np.uint64)).all()
    oi.close()
    reps.close()


def test_u64_symbols_and_errors_come_from_the_failing_shard():
    n = 50001
    t = W.dna_text_np(n, 17)
    gi = F.FMIndex(F.Text.with_max_character(t.astype(np.uint64), 4))
    reps = F.Replicas.of(gi, [0, 0])
    lib = gi._lib
    flat, off, _ = W.substring_patterns_np(t, 3001, 11, 41)
    b = reps.search_many(flat=flat.astype(np.uint64), off=off)
    one = gi.search_many(flat=flat.astype(np.uint64), off=off)
    assert (b.s == one.s).all() and (b.e == one.e).all() and (b.counts >= 1).all()
    # a symbol above max_character in the LAST shard only: the reference panics on cs[c] (fm_index.rs:94); the call
    # reports it, the other shards' results are complete
    bad = flat.astype(np.uint64)
    bad[int(off[2900]) + 3] = 9
    with pytest.raises(F.Error) as ei:
        reps.search_many(flat=bad, off=off)
    assert ei.value.code == L.ERR_SYMBOL_RANGE and "max_character" in str(ei.value)
    # handles of different indexes are refused; so are zero handles, locate without samples, a locate batch whose
    # offsets do not start at 0
    other = F.FMIndex(F.Text.with_max_character(W.dna_text_np(n + 1, 17), 4))
    hs = (C.c_void_p * 2)(gi.handle().value, other.handle().value)
    o = np.zeros(4, np.uint64)
    assert lib.fmx_count_batch_multi(hs, 2, F._p(flat), F._p(off), 3, None, F._p(o), None, None) == L.ERR_ARG
    assert b"replicas of one index" in lib.fmx_last_error()
    assert lib.fmx_count_batch_multi(hs, 0, F._p(flat), F._p(off), 3, None, F._p(o), None, None) == L.ERR_ARG
    s = np.zeros(3, np.uint64)
    assert lib.fmx_locate_batch_multi(reps._handles(), 3, F._p(s), F._p(s + 1), 3, F._p(np.arange(4, dtype=np.uint64)),
                                      F._p(o)) == L.ERR_NO_LOCATE
    other.close()
    reps.close()
    gl = F.FMIndexWithLocate(F.Text.with_max_character(t, 4), 1)
    r2 = F.Replicas.of(gl, [0])
    assert lib.fmx_locate_batch_multi(r2._handles(), 2, F._p(s), F._p(s + 1), 3,
                                      F._p(np.arange(1, 5, dtype=np.uint64)), F._p(np.zeros(8, np.uint64))) == L.ERR_ARG
    r2.close()


def test_concurrent_multi_calls_share_the_worker_pool():
    """several host threads issue sharded batches on the same replicas at once: the per-replica workers serve them in
    turn, every call gets its own results"""
    import threading
    n = 90001
    t = W.dna_text_np(n, 19)
    gi = F.FMIndexWithLocate(F.Text.with_max_character(t, 4), 2)
    reps = F.Replicas.of(gi, [0, 0])
    oi = O.OracleIndex(t, 4, level=2)
    errs = []

    def work(seed):
        try:
            flat, off, _ = W.substring_patterns_np(t, 20011 + seed, 13, seed)
            so, eo = oi.count_batch(flat, off, nthreads=2)
            for _ in range(4):
                b = reps.search_many(flat=flat, off=off)
                assert (b.s == so).all() and (b.e == eo).all()
                k = 4000
                hoff, pos = reps.locate_many(b.s[:k], b.e[:k])
                _, opos = oi.locate_batch(so[:k], eo[:k], nthreads=2)
                assert (pos == np.asarray(opos, np.uint64)).all()
        except Exception as ex:      # noqa: BLE001 -- reported by the main thread
            errs.append(repr(ex))
    th = [threading.Thread(target=work, args=(50 + i,)) for i in range(4)]
    [x.start() for x in th]
    [x.join() for x in th]
    assert not errs, errs
    oi.close()
    reps.close()


def test_c99_caller_shards_over_three_replicas(tmp_path):
    out = subprocess.run([build_c_example(tmp_path, "abi_multi")], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.startswith("ok multi replicas=3 patterns=1003 ")
