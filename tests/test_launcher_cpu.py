"""The N > 1 entry of bench.py without hardware: `python bench.py --gpus N` must start N ranks
itself, refuse a WORLD_SIZE that disagrees with --gpus, fail loudly -- not fall
back to a CPU path -- when there is no GPU, and the gather pipeline it ships (double-buffered async
all-gather with the int32 wire format) must reproduce single-rank counts over gloo."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_spawn_ranks_sets_the_rank_environment(tmp_path):
    from fm_index_amd import launcher
    script = tmp_path / "r.py"
    script.write_text(
        "import os,sys\n"
        "open(os.path.join(sys.argv[1], 'rank%s' % os.environ['RANK']), 'w').write(\n"
        "    ' '.join(os.environ[k] for k in ('RANK','LOCAL_RANK','WORLD_SIZE','MASTER_ADDR','MASTER_PORT')))\n")
    rc = launcher.spawn_ranks(3, [str(script), str(tmp_path)], timeout=60)
    assert rc == 0
    seen = [open(tmp_path / ("rank%d" % r)).read().split() for r in range(3)]
    assert [s[0] for s in seen] == ["0", "1", "2"] and [s[1] for s in seen] == ["0", "1", "2"]
    assert all(s[2] == "3" and s[3] == "127.0.0.1" for s in seen)
    assert len({s[4] for s in seen}) == 1


def test_spawn_ranks_propagates_a_failing_rank(tmp_path):
    from fm_index_amd import launcher
    script = tmp_path / "f.py"
    script.write_text("import os,sys,time\n"
                      "if os.environ['RANK'] == '1': sys.exit(7)\n"
                      "time.sleep(30)\n")
    import time
    t0 = time.time()
    rc = launcher.spawn_ranks(2, [str(script)], timeout=60)
    assert rc == 7
    assert time.time() - t0 < 20          # rank 0 was terminated, not waited for


def test_bench_gpus_flag_starts_ranks_and_fails_loudly_without_a_gpu():
    """No GPU here: each of the 2 ranks must die on the missing device (no CPU fallback), and the
    parent must return their failure."""
    if torch.cuda.is_available():
        pytest.skip("this is the no-GPU half; the GPU half is tests/test_gpu_bench_multirank.py")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dist-backend", "gloo",
                        "--log2n", "16", "--npat", "4096", "--no-cpu-baseline"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert p.returncode != 0
    assert p.stdout.strip() == b""            # no result line was invented
    err = p.stderr.decode(errors="replace")
    assert "HIP" in err or "GPU" in err or "cuda" in err.lower()


def test_bench_refuses_a_world_size_that_disagrees_with_gpus():
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    assert p.returncode != 0 and b"--gpus 4 but WORLD_SIZE=1" in p.stderr


def test_wire_dtype_refuses_truncation():
    from fm_index_amd import sharding
    assert sharding.wire_dtype(1 << 30) == torch.int32
    assert sharding.wire_dtype(1 << 31) == torch.int64
    assert sharding.wire_dtype((1 << 32) - 17) == torch.int64
    with pytest.raises(ValueError):
        sharding.wire_dtype(1 << 31, force=torch.int32)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _pipe_worker(rank, world, port, npat, steps, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from fm_index_amd import sharding as S
    from fm_index_amd import workload as W
    from oracle import fm_oracle as O
    text = W.dna_text_np(30000, 1)
    idx = O.OracleIndex(text, 4)
    ok = True
    pipe = S.CountGatherPipeline(npat, world, len(text), "cpu", backend="gloo")
    assert pipe.wire == torch.int32 and pipe.nbuf == 2
    expect = []
    got = []
    for k in range(steps):               # a different global pattern set every step
        flat, off, _ = W.substring_patterns_np(text, npat * world, 5, 100 + k)
        fs, fe = idx.count_batch(flat, off)
        expect.append((fe - fs).astype(np.int64))
        lo, hi = rank * npat, (rank + 1) * npat
        sub = off[lo:hi + 1] - off[lo]
        s, e = idx.count_batch(flat[int(off[lo]):int(off[hi])], sub)

        def launch(out64, s=s, e=e):     # the per-rank search (the HIP kernel on a GPU box)
            out64.copy_(torch.from_numpy((e - s).astype(np.int64)))
        got.append(pipe.step(launch))
        if k >= 1:                       # buffer k-1 is complete once step k has been issued on the other one
            pass
    pipe.drain()
    # every buffer holds the gather of the LAST step issued on it
    for k in (steps - 2, steps - 1):
        ok = ok and bool((got[k].numpy().astype(np.int64) == expect[k]).all())
    if rank == 0:
        q.put(ok)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_count_gather_pipeline_over_gloo(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_pipe_worker, args=(r, world, port, 500, 5, q)) for r in range(world)]
    for p in procs:
        p.start()
    ok = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert ok


def _one_rank_worker(port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=0, world_size=1)
    from fm_index_amd import sharding as S
    pipe = S.CountGatherPipeline(100, 1, 1 << 20, "cpu", backend="gloo", force_collective=True)
    assert pipe.collective and pipe.nbuf == 2 and len(pipe.gathered) == 2
    plain = S.CountGatherPipeline(100, 1, 1 << 20, "cpu", backend="gloo")
    assert not plain.collective and plain.nbuf == 1
    outs = []
    for k in range(4):
        outs.append(pipe.step(lambda o, k=k: o.copy_(torch.arange(100, dtype=torch.int64) + k)))
    pipe.drain()
    ok = bool((outs[3].to(torch.int64) == torch.arange(100) + 3).all())
    ok = ok and bool((outs[2].to(torch.int64) == torch.arange(100) + 2).all())
    off, pos = S.gather_positions(torch.tensor([2, 0, 1]), torch.tensor([7, 8, 9]), 3)
    ok = ok and off.tolist() == [0, 2, 2, 3] and pos.tolist() == [7, 8, 9]
    q.put(ok)
    dist.destroy_process_group()


def test_forced_collective_on_a_one_rank_group():
    """`bench.py --force-dist`: the gather really runs (double-buffered, async) on a 1-rank group"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_one_rank_worker, args=(_free_port(), q))
    p.start()
    assert q.get(timeout=120)
    p.join(timeout=60)
    assert p.exitcode == 0


def test_distinct_device_check():
    from fm_index_amd import sharding as S
    a = {"pci_bus_id": "0000:05:00.0", "uuid": None, "name": "x"}
    b = {"pci_bus_id": "0000:15:00.0", "uuid": None, "name": "x"}
    assert S.assert_distinct_devices([a, b]) == ["0000:05:00.0", "0000:15:00.0"]
    with pytest.raises(RuntimeError, match="share a physical GPU"):
        S.assert_distinct_devices([a, dict(a)])
    with pytest.raises(RuntimeError, match="cannot identify"):
        S.assert_distinct_devices([a, {"pci_bus_id": None, "uuid": None}])
    # the rank environment always carries the dmabuf-IPC switch RCCL needs on this image
    from fm_index_amd import launcher
    assert launcher.rank_env(1, 2, 1234, base={})["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    assert launcher.rank_env(1, 2, 1234, base={"HSA_ENABLE_IPC_MODE_LEGACY": "1"})["HSA_ENABLE_IPC_MODE_LEGACY"] == "1"
    assert launcher.rank_env(0, 2, 1234, base={})["GPU_MAX_HW_QUEUES"] == "8"
    assert launcher.rank_env(0, 2, 1234, base={"GPU_MAX_HW_QUEUES": "4"})["GPU_MAX_HW_QUEUES"] == "4"


def _strong_worker(rank, world, port, total, q):
    """strong scaling (bench.py --total-patterns): a FIXED global pattern set in contiguous, possibly ragged shards
    (sharding.shard_range), every rank's slot padded to the largest shard, the compacted gather hashed"""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    from fm_index_amd import sharding as S
    from fm_index_amd import workload as W
    from oracle import fm_oracle as O
    text = W.dna_text_np(30000, 1)
    idx = O.OracleIndex(text, 4)
    flat, off, _ = W.substring_patterns_np(text, total, 6, 7)
    fs, fe = idx.count_batch(flat, off)
    expect = (fe - fs).astype(np.int64)
    sizes = [S.shard_range(total, r, world)[1] - S.shard_range(total, r, world)[0] for r in range(world)]
    slot = max(sizes)
    lo, hi = S.shard_range(total, rank, world)
    assert hi - lo == sizes[rank] and sum(sizes) == total
    s, e = idx.count_batch(flat[int(off[lo]):int(off[hi])], off[lo:hi + 1] - off[lo])
    pipe = S.CountGatherPipeline(slot, world, len(text), "cpu", backend="gloo")

    def launch(out64):                   # a rank writes only its shard's entries of its (padded) slot
        out64[:hi - lo].copy_(torch.from_numpy((e - s).astype(np.int64)))
    for _ in range(3):
        g = pipe.step(launch)
    pipe.drain()
    allc = S.compact_padded(g, sizes, slot).numpy().astype(np.int64)
    ok = allc.shape == (total,) and bool((allc == expect).all())
    ok = ok and bench.counts_sha256(allc) == bench.counts_sha256(expect)
    if rank == 0:
        q.put((ok, bench.counts_sha256(allc)))
    dist.barrier()
    dist.destroy_process_group()


def test_strong_scaling_shards_hash_like_one_rank():
    """config 5: the hash of the gathered counts is the same at G = 1, 2 (equal shards) and 3 (ragged shards)"""
    shas = []
    for world in (1, 2, 3):
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=_strong_worker, args=(r, world, port, 1000, q)) for r in range(world)]
        for p in procs:
            p.start()
        ok, sha = q.get(timeout=240)
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
        assert ok
        shas.append(sha)
    assert len(set(shas)) == 1


def test_counts_sha256_is_of_the_int64_little_endian_counts():
    import hashlib
    import bench
    c = np.array([1, 2, 70000, 0], dtype=np.int32)
    assert bench.counts_sha256(c) == hashlib.sha256(c.astype("<i8").tobytes()).hexdigest()
    assert bench.counts_sha256(c) == bench.counts_sha256(c.astype(np.uint64))
    assert bench.golden_key("dna", 30, 7, 8 << 20, 32) == "dna:n=2^30:seed=7:patterns=8388608:len=32"
