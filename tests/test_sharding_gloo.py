"""world_size-2 (and 3, ragged) CPU tests of the N>1 path over gloo: sharding + gathers
reproduce the single-rank output exactly.  The per-rank search runs on the CPU oracle here
(this is a test of the distribution logic; the GPU search itself is covered by -m gpu)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, npat, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from fm_index_amd import sharding as S
    from fm_index_amd import workload as W
    from oracle import fm_oracle as O
    text = W.dna_text_np(20000, 1)
    idx = O.OracleIndex(text, 4, level=2)            # replicated index
    flat, off, _ = W.substring_patterns_np(text, npat, 6, 7)
    lo, hi = S.shard_range(npat, rank, world)        # this rank's contiguous shard
    sub_off = off[lo:hi + 1] - off[lo]
    s, e = idx.count_batch(flat[int(off[lo]):int(off[hi])] if hi > lo else flat[:1], sub_off)
    loff, lpos = idx.locate_batch(s, e)
    cnt = torch.from_numpy((e - s).astype(np.int64))
    allc = S.gather_counts(cnt, npat)
    goff, gpos = S.gather_positions(cnt, torch.from_numpy(lpos.astype(np.int64)), npat)
    # the planned form: sizes exchanged once, then repeated gathers of the same shape with new values
    plan = S.PositionGatherPlan(cnt, npat)
    lp = torch.from_numpy(lpos.astype(np.int64))
    plan.gather(lp)
    again = plan.compact().clone()
    plan.gather(lp + 5)
    shifted = plan.compact().clone()
    plan_ok = bool((again == gpos).all()) and bool((shifted == gpos + 5).all()) and bool((plan.off == goff).all()) \
        and sum(plan.totals) == int(goff[-1])
    if rank == 0:
        fs, fe = idx.count_batch(flat, off)
        foff, fpos = idx.locate_batch(fs, fe)
        ok = bool((allc.numpy() == (fe - fs).astype(np.int64)).all()) and \
            bool((goff.numpy() == foff.astype(np.int64)).all()) and \
            bool((gpos.numpy() == fpos.astype(np.int64)).all()) and plan_ok
        q.put(ok)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,npat", [(2, 1000), (2, 1001), (3, 1000)])
def test_sharded_equals_single(world, npat):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, npat, q)) for r in range(world)]
    for p in procs:
        p.start()
    ok = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert ok


def test_shard_range_partitions():
    from fm_index_amd.sharding import shard_range
    for n in (0, 1, 7, 8, 1000, 1 << 20):
        for w in (1, 2, 3, 4, 8):
            cuts = [shard_range(n, r, w) for r in range(w)]
            assert cuts[0][0] == 0 and cuts[-1][1] == n
            assert all(cuts[i][1] == cuts[i + 1][0] for i in range(w - 1))
            assert max(c[1] - c[0] for c in cuts) - min(c[1] - c[0] for c in cuts) <= 1
