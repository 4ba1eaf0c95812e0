"""The N > 1 legs of bench.py: process group, the self-verifying report of the gathers, the config-5 step through a
1-rank RCCL communicator and config 5's whole pattern set on one GPU."""
import ctypes as C
import datetime
import os
import sys
import time

from .common import counts_sha256, flush_c_stdio, golden_counts_sha, ranges_sha256, wl_oracle
from .cpu import host_cpu

# rank 0 reaches the rendezvous after its counter passes (two rocprofv3 runs over a child that builds the index),
# and is waited for in a gloo barrier while it measures the CPU baseline
RENDEZVOUS_TIMEOUT = datetime.timedelta(minutes=30)


def open_process_group(torch, local, rank, world, gloo):
    """one process per GPU: backend "nccl" IS RCCL on ROCm (communicator bound to this rank's device);
    "gloo" is the rehearsal in which all ranks share cuda:0 and gather through host memory"""
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if "MASTER_PORT" not in os.environ:            # --force-dist at N=1, started without a launcher
        from fm_index_amd import launcher
        os.environ["MASTER_PORT"] = str(launcher.free_port())
    os.environ.setdefault("RANK", str(rank))
    os.environ.setdefault("WORLD_SIZE", str(world))
    # RCCL prints a version banner through C stdio to stdout; the result line must be alone there: while the group
    # comes up, file descriptor 1 points at stderr, and the C buffer is flushed before it is put back
    sys.stdout.flush()
    saved = os.dup(1)
    os.dup2(2, 1)
    try:
        if gloo:
            torch.cuda.set_device(0)
            dist.init_process_group("gloo", rank=rank, world_size=world, timeout=RENDEZVOUS_TIMEOUT)
        else:
            torch.cuda.set_device(local)
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local),
                                    timeout=RENDEZVOUS_TIMEOUT)
        flush_c_stdio()
    finally:
        os.dup2(saved, 1)
        os.close(saved)
    return dist



def rccl_version_string(torch):
    try:
        v = torch.cuda.nccl.version()
        return ".".join(str(x) for x in v) if isinstance(v, (tuple, list)) else str(v)
    except Exception:  # noqa: BLE001
        return None


def timed_sync_gather(torch, dist, pipe, gloo, reps=5):
    """one count gather on its own (not overlapped), ms: HIP events on the launch stream around the
    synchronous collective (nccl) / wall clock (gloo through host memory)"""
    src, dst = pipe.local_w[0], pipe.gathered[0]
    dist.all_gather_into_tensor(dst, src)
    torch.cuda.synchronize()
    if gloo:
        t0 = time.perf_counter()
        for _ in range(reps):
            dist.all_gather_into_tensor(dst, src)
        return (time.perf_counter() - t0) / reps * 1e3
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        dist.all_gather_into_tensor(dst, src)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def dist_report(out, torch, dist, sharding, pipe, wl, args, world, rank, local, gloo, dt_rank, ev_ms,
                kernel_ms_single, step):
    """the self-verifying part of the N>1 line: which backend really carried the gathers, which physical
    GPU every rank sat on (asserted distinct under nccl), per-rank kernel / gather / step times, and the
    event timeline showing gather k in flight under search k+1"""
    backend = dist.get_backend()
    nccl = backend == "nccl"
    npat = wl.npat
    gather_ms = timed_sync_gather(torch, dist, pipe, gloo)
    mine = torch.tensor([kernel_ms_single, gather_ms, dt_rank / args.steps * 1e3, ev_ms / args.steps],
                        dtype=torch.float64, device="cpu" if gloo else wl.dev)
    allr = torch.empty(4 * world, dtype=torch.float64, device=mine.device)
    dist.all_gather_into_tensor(allr, mine)
    allr = allr.cpu().view(world, 4).tolist()
    idents = sharding.gather_device_identities(local)
    if nccl:                     # under gloo all ranks share cuda:0 on purpose (rehearsal)
        sharding.assert_distinct_devices(idents)
    # event timeline of a few traced steps (outside the timed region)
    tr = traced_steps(torch, sharding, wl, world) if nccl else None
    # rccl_ranks: ranks of the RCCL communicator that carried the gathers -- null unless the backend is nccl
    out["rccl_ranks"] = dist.get_world_size() if nccl else None
    out["rccl_version"] = rccl_version_string(torch) if nccl else None
    out["dist_backend"] = backend
    out["devices"] = idents
    out["per_rank"] = [{"rank": r, "kernel_ms": round(v[0], 4), "gather_ms": round(v[1], 4),
                        "wall_ms_per_step": round(v[2], 4), "stream_ms_per_step": round(v[3], 4)}
                       for r, v in enumerate(allr)]
    out["gather"] = {"backend": "gloo (rehearsal through host memory)" if gloo else "nccl (RCCL)",
                     "counts_wire_dtype": str(pipe.wire).replace("torch.", ""),
                     "bytes_per_rank_per_step": wl.npat_pad * (4 if pipe.wire == torch.int32 else 8),
                     "shard_sizes": wl.shard_sizes if len(set(wl.shard_sizes)) > 1 else wl.shard_sizes[0],
                     "pipelined": pipe.nbuf > 1, "trace": tr}


def traced_steps(torch, sharding, wl, world, steps=8):
    """event timeline of a few steps of the count + gather pipeline (sharding.CountGatherPipeline.trace_report)"""
    tp = sharding.CountGatherPipeline(wl.npat_pad, world, wl.n, wl.dev, backend="nccl", force_collective=True, trace=True)
    for _ in range(steps):
        tp.step(lambda out64: wl.count(out_cnt=out64))
    tp.drain()
    torch.cuda.synchronize()
    return tp.trace_report()


def rccl_1rank_leg(out, wl, args, dev, local):
    """default N=1 run: the step that ships for N>1 (sharding.CountGatherPipeline + gather_positions) through
    a 1-rank RCCL communicator on this GPU, so that the driver's single-GPU line carries hardware evidence of
    the RCCL path (communicator, device-side all_gather_into_tensor, async Work ordering under the next
    search) even when no multi-GPU node is available.  Counts and positions must equal the ungathered ones."""
    import torch
    from fm_index_amd import sharding
    dist = open_process_group(torch, local, 0, 1, False)
    try:
        npat, m = wl.npat, wl.m
        wl.count()
        torch.cuda.synchronize()
        ref_c = wl.d_c.clone()
        pipe = sharding.CountGatherPipeline(npat, 1, wl.n, dev, backend="nccl", force_collective=True)
        for _ in range(args.warmup):
            pipe.step(lambda o: wl.count(out_cnt=o))
        pipe.drain()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            g = pipe.step(lambda o: wl.count(out_cnt=o))
        pipe.drain()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        assert bool((g.to(torch.int64) == ref_c).all()), "counts gathered over RCCL differ"
        o = {"backend": dist.get_backend(), "rccl_version": rccl_version_string(torch), "ranks": dist.get_world_size(),
             "device": sharding.device_identity(local), "value": npat * m * args.steps / dt,
             "unit": "pattern-chars/s", "ms_per_step": dt / args.steps * 1e3,
             "counts_wire_dtype": str(pipe.wire).replace("torch.", ""),
             "gather_ms": round(timed_sync_gather(torch, dist, pipe, False), 4), "trace": traced_steps(torch, sharding, wl, 1),
             "note": "config-5 step at one rank: pipelined all_gather_into_tensor of the counts over a 1-rank RCCL "
                     "communicator on this GPU; counts identical to the ungathered run"}
        if wl.level is not None and getattr(wl, "total_hits", None):
            wl.locate()
            cnt = (wl.d_e - wl.d_s)
            goff, gpos = sharding.gather_positions(cnt, wl.d_pos[:wl.total_hits], npat)
            torch.cuda.synchronize()
            assert int(goff[-1].item()) == wl.total_hits and bool((gpos == wl.d_pos[:wl.total_hits]).all()), \
                "positions gathered over RCCL differ"
            o["positions_gathered"] = wl.total_hits
        out["rccl_1rank"] = o
        # ---- BASELINE config 5 at G = 1: the whole 8 M-pattern set on this GPU, through the same communicator ----
        if wl.dna and not args.no_config5:
            try:
                config5_g1_leg(out, wl, args, dev)
            except Exception as ex:  # noqa: BLE001 -- never lose the headline line to an extra leg
                out["config5_g1"] = {"error": repr(ex)}
    finally:
        dist.destroy_process_group()


CONFIG5_PATTERNS = 8 << 20          # BASELINE.json configs[4]: 8M length-32 patterns; SURVEY 8d: seed 7
CONFIG5_SEED = 7


def config5_g1_leg(out, wl, args, dev):
    """BASELINE config 5 at one GPU: ALL 8 388 608 length-32 substring patterns (seed 7; the set `--gpus G
    --total-patterns 8388608` shards over G ranks, and the set the default `--gpus 8` weak run searches) in one batch
    on this GPU, the int32 counts all-gathered through the 1-rank RCCL communicator every step.  counts_sha256 is the
    hash every G must reproduce; (s, e) of a 2^15-pattern sample (every 256th pattern) is compared with the CPU oracle."""
    import numpy as np
    import torch
    from fm_index_amd import sharding
    from fm_index_amd import workload as W
    lib, n, m = wl.lib, wl.n, wl.m
    T = CONFIG5_PATTERNS if args.log2n >= 30 else max(args.npat * 8, 1 << 15)
    pat = torch.empty(T * m, dtype=torch.uint8, device=dev)
    ar = torch.arange(m, dtype=torch.int64, device=dev)[None, :]
    chunk = 1 << 20
    for lo in range(0, T, chunk):                      # in chunks: the (patterns x m) int64 index tensor is 2 GB at once
        k = min(chunk, T - lo)
        src = W.umod_torch(W.splitmix64_torch(CONFIG5_SEED, lo, k, dev), n - 1 - m)
        pat[lo * m:(lo + k) * m] = wl.text[src[:, None] + ar].reshape(-1)
    del src
    off = (torch.arange(T + 1, dtype=torch.int64, device=dev) * m).contiguous()
    s = torch.empty(T, dtype=torch.int64, device=dev)
    e = torch.empty(T, dtype=torch.int64, device=dev)
    pipe = sharding.CountGatherPipeline(T, 1, n, dev, backend="nccl", force_collective=True)

    def launch(out64):
        rc = lib.fmx_count_batch_dev(wl.h, C.c_void_p(pat.data_ptr()), C.c_void_p(off.data_ptr()), T, None,
                                     C.c_void_p(s.data_ptr()), C.c_void_p(e.data_ptr()), C.c_void_p(out64.data_ptr()), wl.sp)
        if rc != 0:
            raise RuntimeError(lib.fmx_last_error().decode())
    for _ in range(max(2, args.warmup // 2)):
        pipe.step(launch)
    pipe.drain()
    torch.cuda.synchronize()
    steps = max(5, args.steps // 2)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record(wl.stream)
    for _ in range(steps):
        g = pipe.step(launch)
    pipe.drain()
    ev1.record(wl.stream)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert lib.fmx_stream_status(wl.h) == 0
    lib.fmx_set_timing(wl.h, 1)
    launch(pipe.local64[0])
    torch.cuda.synchronize()
    kms, executed = lib.fmx_last_kernel_ms(wl.h), int(lib.fmx_last_steps(wl.h))
    lib.fmx_set_timing(wl.h, 0)
    assert executed == T * m, (executed, T * m)
    cnt = g.to(torch.int64)
    assert bool((cnt == e - s).all()) and bool((cnt >= 1).all())
    sha = counts_sha256(cnt.cpu().numpy())
    rsha = ranges_sha256(s.cpu().numpy(), e.cpu().numpy())
    # the weak run's rank-0 shard is the first 2^20 patterns of this set
    o = {"workload": "config5 at G=1: %d x len-%d substring patterns (seed %d) in one batch, counts all-gathered (int32) "
                     "through a 1-rank RCCL communicator every step" % (T, m, CONFIG5_SEED),
         "total_patterns": T, "value": T * m * steps / dt, "unit": "pattern-chars/s", "steps": steps,
         "ms_per_step": dt / steps * 1e3, "stream_ms_per_step": ev0.elapsed_time(ev1) / steps, "kernel_ms": round(kms, 4),
         "executed_steps": executed, "vs_headline_value": round(T * m * steps / dt / out["value"], 4),
         "counts_sha256": sha, "ranges_sha256": rsha, "counts_sum": int(cnt.sum().item())}
    gold = golden_counts_sha(wl, args, total=T, seed=CONFIG5_SEED)
    if gold is not None:
        o["matches_golden"] = {"counts_sha256": gold[0] == sha, "ranges_sha256": gold[1] == rsha,
                               "source": "tests/golden/config5_counts.json (CPU oracle over all patterns)"}
        assert gold[0] == sha and gold[1] in (None, rsha), "config-5 results differ from tests/golden/config5_counts.json"
    if not args.no_cpu_baseline:
        oi, _ = wl_oracle(wl, "fm")
        k = 1 << 15
        idx = torch.arange(0, T, T // k, device=dev)[:k]
        ph = pat.view(T, m)[idx].reshape(-1).cpu().numpy()
        so, eo = oi.count_batch(ph, np.arange(k + 1, dtype=np.uint64) * np.uint64(m), nthreads=host_cpu()["effective_cpus"])
        ok = (so == s[idx].cpu().numpy().view(np.uint64)).all() and (eo == e[idx].cpu().numpy().view(np.uint64)).all()
        assert ok, "config5_g1: GPU (s, e) != oracle on the sample"
        o["oracle_sample"] = {"patterns": k, "stride": T // k, "identical_s_e": True}
    out["config5_g1"] = o

