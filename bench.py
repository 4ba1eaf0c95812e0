#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on MI355X: pattern-chars/s of batched backward search
(count) on a 1 GB sigma=4 DNA text (config 2: FMIndex, 2^20 length-32 patterns that are
substrings of the text, so all 32 steps execute) as `value`, with every other BASELINE config in
the same line: `locate` (config 3), `locate_3b` (64 K short patterns, wide intervals), `rlfm`
(config 4: RLFMIndex over a 1 GB sigma=255 byte text, 2^20 length-16 patterns, with its own
roofline and CPU baseline), `value_incl_d2h` (the same batch through the host-pointer entry point)
and, with --gpus N, config 5: N x 2^20 patterns sharded over N ranks, index replicated, counts and
positions gathered over RCCL inside the timed region.

`python bench.py --gpus N` starts the N ranks itself (one process per GPU, before anything touches
the GPU); under `python -m torch.distributed.run` it is one of the ranks.  A "step" = one pass of
the count kernel over this rank's pattern batch, inputs and outputs resident in HBM.  Rank 0
prints ONE JSON line.
"""
import argparse
import csv
import ctypes as C
import datetime
import glob
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# Process-level runtime settings, fixed before torch, HIP or OpenMP are initialised (none is a machine setting):
#  * HIP maps streams onto GPU_MAX_HW_QUEUES hardware queues (4 by default).  Streams that share one run in
#    submission order: the count gather of step k then sits BETWEEN searches k and k+1 instead of under k+1
#    (profiles/r03/rccl_overlap_probe_hwqueues.jsonl).  8 queues; the headline is unaffected
#    (profiles/r03/hw_queues_ab.txt).
# (The CPU baseline's threads are placed by the oracle itself, per batch -- OMP_PROC_BIND would narrow the
# affinity mask of this Python thread for good, and with it the mask of every HIP / RCCL helper thread.)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
#  * dmabuf IPC between the ranks of one node (the host driver of this pool supports nothing else: without it RCCL fails
#    with `hipIpcGetMemHandle: invalid argument`); exported on the pool's boxes already, kept when the caller set it
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


from benchmarks.legs.common import (PRETOUCH, Workload, counts_sha256, flush_c_stdio, golden_counts_sha,  # noqa: E402,F401
                                    golden_key, pretouch_device, ranges_sha256)
from benchmarks.legs.cabi import config5_cabi_leg, results_on_host_leg, single_call_leg  # noqa: E402
from benchmarks.legs.cpu import cpu_baseline, host_cpu  # noqa: E402,F401
from benchmarks.legs.dist import dist_report, open_process_group, rccl_1rank_leg  # noqa: E402
from benchmarks.legs.extra import accel_legs, d2h_leg, ic_ab_leg, rlfm_leg, wide_leg  # noqa: E402
from benchmarks.legs.locate import locate_3b, locate_leg, locate_row_order_leg  # noqa: E402
from benchmarks.legs.roofline import (PMC_LEGS, PMC_WHICH, WALK_KERNEL, apply_pmc, make_roofline,  # noqa: E402,F401
                                      pmc_aggregate, pmc_child, pmc_per_dispatch, run_census, run_pmc_passes,
                                      stored_traffic, two_stream_roofline)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--log2n", type=int, default=30, help="text length 2^k incl. terminator")
    ap.add_argument("--npat", type=int, default=1 << 20, help="patterns per GPU (weak scaling, the default)")
    ap.add_argument("--total-patterns", type=int, default=None,
                    help="strong scaling (BASELINE config 5): a FIXED global set of this many patterns (seed 7), "
                         "rank r searching the contiguous shard [ceil(T r / G), ceil(T (r+1) / G)); --npat is ignored.  "
                         "`--gpus G --total-patterns 8388608` is config 5; at G = 1 the step still goes through a 1-rank "
                         "RCCL communicator so that every point of the curve includes the gather")
    ap.add_argument("--plen", type=int, default=32)
    ap.add_argument("--level", type=int, default=2, help="SA sampling level for the locate legs")
    ap.add_argument("--workload", default="dna", choices=["dna", "bytes-fm", "bytes-rlfm", "rep-fm", "rep-rlfm"],
                    help="headline workload: dna = config 2/3/5; bytes-rlfm = config 4 as the headline "
                         "(the default run reports it in the `rlfm` object); bytes-fm = FMIndex on the "
                         "config-4 text; rep-* = config 4b: 1 MiB random block repeated with point mutations")
    ap.add_argument("--mut-per-1024", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-locate", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    ap.add_argument("--no-accel", action="store_true", help="skip the opt-in index legs (pair index, k-mer table)")
    ap.add_argument("--no-early-exit", action="store_true", help="skip the config-2b legs")
    ap.add_argument("--no-rlfm", action="store_true", help="skip the config-4 object of the default run")
    ap.add_argument("--no-3b", action="store_true", help="skip the config-3b object")
    ap.add_argument("--no-wide", action="store_true", help="skip the n = 2^32 + 2^20 object (the 64-bit engine)")
    ap.add_argument("--pretouch", action="store_true",
                    help="have a child process write the device's free memory once before the run (see pretouch_device); "
                         "off by default since round 5: the builder allocates its buffers before it touches any")
    ap.add_argument("--no-ic-ab", action="store_true",
                    help="skip `count_n31`: the headline kernel on an index four times the Infinity Cache (n = 2^31)")
    ap.add_argument("--detail-out", default=None,
                    help="where the full result object goes (every leg, every roofline, thread sweeps, traces); default "
                         "bench_detail.json next to bench.py (bench_detail_gN.json for --gpus N > 1).  stdout's last line is "
                         "the compact headline object (< 4 KB) that refers to it")
    ap.add_argument("--no-d2h", action="store_true", help="skip value_incl_d2h")
    ap.add_argument("--no-census", action="store_true", help="skip the requested / distinct line census")
    ap.add_argument("--no-pmc", action="store_true",
                    help="do not run the rocprofv3 --pmc child passes that measure roofline.traffic live")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--cabi-child", action="store_true",
                    help="no bench line: config 5 through the C ABI over every device this ONE process sees (fmx_replicate + "
                         "fmx_count_batch_multi[_resident] at G = 1, 2, 4, 8 replicas on distinct devices); prints one JSON "
                         "object.  The default N = 1 run starts it as a child when it sees more than one device")
    ap.add_argument("--dist-backend", default="nccl",
                    help="nccl (= RCCL, the real path) | gloo (rehearsal of the N>1 code path: all ranks "
                         "share cuda:0 and gather through host memory)")
    ap.add_argument("--force-dist", action="store_true",
                    help="with --gpus 1: still open the process group (a 1-rank RCCL communicator on this GPU) and "
                         "drive the N>1 step -- pipelined all-gather of the counts, counts-then-positions gather of "
                         "locate -- through it; the line is the config-5 line at one rank")
    ap.add_argument("--no-config5", action="store_true",
                    help="default N=1 run: skip the `config5_g1` object (the 8 388 608-pattern set of config 5 on one GPU)")
    ap.add_argument("--no-rccl-check", action="store_true",
                    help="default N=1 run: skip the `rccl_1rank` object (the config-5 step through a 1-rank RCCL "
                         "communicator, after the headline measurement)")
    ap.add_argument("--headline-of", default=None, metavar="DETAIL.json",
                    help="no run: print the compact line of an existing detail file (the last JSON line of the file "
                         "when it holds several) exactly as a run would, and exit -- the print path without a GPU")
    ap.add_argument("--dump-counts", default=None, help=argparse.SUPPRESS)   # tests: gathered counts -> .npy
    ap.add_argument("--pattern-seed", type=int, default=None, help=argparse.SUPPRESS)   # tests: same global set at any N
    return ap.parse_args(argv)



# --------------------------------------------------------------------------------------------
def main():
    args = parse_args()
    if args.headline_of:
        with open(args.headline_of) as f:
            txt = f.read().strip()
        try:
            full = json.loads(txt)
        except ValueError:
            full = json.loads([ln for ln in txt.splitlines() if ln.startswith("{")][-1])
        print(json.dumps(headline(full, args.headline_of), separators=(",", ":")))
        return
    if args.gpus > 1 and "RANK" not in os.environ:
        # one process per GPU, started before anything in THIS process touches the GPU
        from fm_index_amd import launcher
        sys.exit(launcher.spawn_ranks(args.gpus, [os.path.abspath(__file__)] + sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and not args.pmc_child and not args.cabi_child:
        sys.exit("bench.py: --gpus %d but WORLD_SIZE=%d (start it as `python bench.py --gpus N`, or under "
                 "torch.distributed.run with --nproc-per-node equal to --gpus)" % (args.gpus, world))
    if args.pmc_child:
        pmc_child(args)
        return
    if args.cabi_child:
        from benchmarks.legs.cabi import cabi_child
        cabi_child(args)
        return
    # fabric traffic (L2 -> Infinity Cache / HBM) of this build, measured now: rocprofv3 --pmc passes over a child of this
    # script.  Started BEFORE this process touches the GPU (no torch import yet): the children are
    # then spawned from a process that holds no device state.
    pmc = None
    rank = int(os.environ.get("RANK", "0"))
    if (world == 1 and not args.total_patterns and args.workload == "dna" and not args.no_wide and args.log2n >= 30
            and args.pretouch):
        pretouch_device(int(os.environ.get("LOCAL_RANK", "0")))
    t_pmc = time.perf_counter()
    dist_line = world > 1 or args.force_dist or bool(args.total_patterns)
    if rank == 0 and args.workload == "dna" and not args.no_pmc and args.dist_backend != "gloo":
        try:
            if not dist_line:
                pmc = run_pmc_passes(args)
            else:
                # the config-5 line: counters of RANK 0's count launch (its shard's shape) on rank 0's GPU, collected
                # now -- the other ranks build their indexes meanwhile and wait for rank 0 in the rendezvous
                shard0 = (args.total_patterns + world - 1) // world if args.total_patterns else args.npat
                pmc = run_pmc_passes(args, npat=shard0, count_only=True)
        except Exception as ex:  # noqa: BLE001 -- never lose the line to the counter passes
            pmc = ({}, repr(ex))
    # ... and of the same kernel on an index four times the Infinity Cache (`count_n31`, the cache A/B of the headline)
    pmc31 = None
    if pmc is not None and pmc[0] and ic_ab_wanted(args, world, dist_line):
        try:
            a31 = argparse.Namespace(**vars(args))
            a31.log2n = args.log2n + 1
            pmc31 = run_pmc_passes(a31, count_only=True, seed7=False)
        except Exception as ex:  # noqa: BLE001
            pmc31 = ({}, repr(ex))
    run(args, world, pmc, pmc_seconds=time.perf_counter() - t_pmc, pmc31=pmc31)


def ic_ab_wanted(args, world, dist_line):
    return (world == 1 and not dist_line and args.workload == "dna" and args.log2n == 30 and not args.no_ic_ab)



def run(args, world, pmc=None, pmc_seconds=0.0, pmc31=None):
    import torch
    import numpy as np
    import fm_index_amd as F
    from fm_index_amd import workload as W
    from fm_index_amd import sharding

    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if rank != 0:            # the ranks share one stdout: only rank 0 may write to it (libraries' banners included)
        os.dup2(2, 1)
    # wall-clock seconds of every part of the run (what the driver's clock around `python bench.py` is made of)
    leg_seconds = {"pretouch": PRETOUCH.get("seconds", 0.0), "pmc_passes": round(pmc_seconds, 1)}
    lap_t = [time.perf_counter()]

    def lap(name):
        now = time.perf_counter()
        leg_seconds[name] = round(leg_seconds.get(name, 0.0) + now - lap_t[0], 1)
        lap_t[0] = now

    gloo = args.dist_backend == "gloo"
    strong = bool(args.total_patterns)
    # the N>1 step, also on a 1-rank communicator when forced -- and at the G = 1 point of the strong-scaling curve,
    # so that every point of config 5 includes the gather ("throughput at G = 1,2,4,8 incl. gather")
    use_dist = world > 1 or args.force_dist or strong
    if use_dist:
        dist = open_process_group(torch, local, rank, world, gloo)
        if gloo:
            local = 0
    else:
        dist = None
        torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    # a throw-away 4 KiB build first: runtime / code-object initialisation is not index construction
    F.FMIndex(F.Text.with_max_character(W.dna_text_np(4096, 9), 4), device=local).close()
    wl = Workload(args.workload, args, dev, local, rank, world)
    lib, npat, m, n = wl.lib, wl.npat, wl.m, wl.n
    total_pat = wl.total_patterns
    stream = wl.stream

    # ---- headline: count (+ gather of every rank's counts for N > 1, config 5) ----
    # every rank's slot in the gathered buffer is npat_pad = the largest shard (ragged shards: T not a multiple of G)
    pipe = sharding.CountGatherPipeline(wl.npat_pad, world, n, dev, backend="gloo" if gloo else "nccl",
                                        pipelined=not os.environ.get("FMX_BENCH_SYNC_GATHER"),
                                        force_collective=use_dist)

    def step():
        return pipe.step(lambda out64: wl.count(out_cnt=out64))

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    pipe.drain()
    barrier()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record(stream)
    for _ in range(args.steps):
        step()
    pipe.drain()                 # every gather of the timed steps has completed
    ev1.record(stream)
    barrier()
    dt = time.perf_counter() - t0
    dt_rank = dt
    if use_dist:
        tt = torch.tensor([dt], dtype=torch.float64, device="cpu" if gloo else dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    ev_ms = ev0.elapsed_time(ev1)
    assert lib.fmx_stream_status(wl.h) == 0

    # ---- validation + step census (outside the timed region) ----
    kernel_ms_single, steps_exec = wl.timed_kernel(lambda: wl.count())
    last = step()
    pipe.drain()
    torch.cuda.synchronize()
    if use_dist:
        # gathered counts: rank r's shard sits at [r * npat_pad, r * npat_pad + shard_sizes[r]); this rank's
        # slot must equal its own counts, and the compacted buffer IS the one-GPU output (input order)
        mine = last[rank * wl.npat_pad:rank * wl.npat_pad + npat].to(torch.int64).to(dev)
        assert bool((mine == wl.d_c).all()), "gathered counts differ from this rank's"
        allc = sharding.compact_padded(last, wl.shard_sizes, wl.npat_pad).cpu().numpy().astype(np.int64)
        # the ranges themselves, gathered ONCE for validation (the timed step gathers the counts only: SURVEY 8e)
        se = torch.zeros(wl.npat_pad, 2, dtype=torch.int64, device="cpu" if gloo else dev)
        se[:npat, 0], se[:npat, 1] = wl.d_s, wl.d_e
        allse = torch.empty(wl.npat_pad * world, 2, dtype=torch.int64, device=se.device)
        dist.all_gather_into_tensor(allse, se)
        allse = sharding.compact_padded(allse, wl.shard_sizes, wl.npat_pad).cpu().numpy()
        alls, alle = allse[:, 0], allse[:, 1]
        del se, allse
    else:
        allc, alls, alle = wl.d_c.cpu().numpy(), wl.d_s.cpu().numpy(), wl.d_e.cpu().numpy()
    assert allc.shape == (total_pat,) and bool((alle - alls == allc).all())
    counts_sha, ranges_sha = counts_sha256(allc), ranges_sha256(alls, alle)
    del alls, alle
    if args.dump_counts and rank == 0:
        np.save(args.dump_counts, allc)
    assert steps_exec == npat * m, (steps_exec, npat * m)   # substrings: every step executes
    assert bool((wl.d_c >= 1).all()), "a substring of the text must occur at least once"

    chars_per_step_rank = npat * m
    value = total_pat * m * args.steps / dt
    # dominant kernel's average launch duration: the event bracket of the timed region at N=1
    # (launches back to back on one stream), the library's per-launch events at N>1
    avg_kernel_ms = ev_ms / args.steps if not use_dist else kernel_ms_single
    stream_bytes = npat * m + (npat + 1) * 8 + 3 * npat * 8      # pattern bytes + offsets + (s, e, count)
    cen = None
    if not args.no_census and rank == 0:
        cen = run_census(wl, lambda cl: wl.count(lib=cl), npat * m * (8 if wl.rlfm else 3) + (1 << 20))
    key = "%s:%d:%d:%d" % (args.workload, npat, m, args.log2n)
    # the DEFAULT DNA index of 2^24+ symbols carries the pair index and the k-mer start table (round 6: same (s, e), 2.2 x
    # the rate); the plain index -- the reference's loop step for step -- is the `plain` object / `value_plain`
    accelerated = wl.accelerated()
    kname = {"dna": "fmx_count_pair_kernel<true>" if accelerated else "fmx_count_f3_kernel<1,false,false>"}.get(
        args.workload, "fmx_count_ep_kernel" if wl.rlfm else "fmx_count_kernel<FMX_KIND_FM>")
    roofline = make_roofline(kname, avg_kernel_ms, chars_per_step_rank, wl.ref_bytes_per_char(), stream_bytes,
                             cen, stored_traffic(key, "count"), table_bytes=wl.count_table_bytes())

    out = {
        # BASELINE.json's metric, verbatim; `value` is its count half (pattern-chars/s), the locate
        # half (hits/s) is out["locate"]["hits_per_s"]
        "metric": "pattern-chars/sec backward search (count) + locate hits/sec, 1 GB text",
        "value": value, "unit": "pattern-chars/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
        "scaling": "strong" if strong else "weak", "vs_baseline": None, "dtype": "u32", "data": "synthetic",
        "config": {"workload": wl.describe(world), "text_len": n, "patterns_per_gpu": npat, "pattern_len": m,
                   "total_patterns": total_pat, "pattern_seed": wl.pattern_seed,
                   "parallelism": "patterns sharded x%d, index replicated" % world,
                   "index_bytes": wl.index.heap_size(), "build_ms": round(wl.build_ms, 1),
                   "textgen_s": round(wl.textgen_s, 2),
                   "index": ("default index: pair index + k-mer start table (k = %d) next to the plain count records "
                             "(FMX_FLAG_PLAIN vetoes; `plain` / `value_plain` is that index)" % wl.index.kmer_k())
                   if accelerated else "plain index"},
        "roofline": roofline,
        # sha256 over the int64 little-endian counts of ALL patterns of the global set in input order: equal at
        # every G for the same global set ("multi-GPU output identical to 1-GPU output")
        "counts_sha256": counts_sha,
        # ... and over the (s, e) pairs, gathered once outside the timed region
        "ranges_sha256": ranges_sha,
    }
    gold = golden_counts_sha(wl, args)
    if gold is not None:
        # tests/golden/config5_counts.json: made by the CPU oracle over ALL patterns of this set
        out["matches_golden"] = {"counts_sha256": gold[0] == counts_sha, "ranges_sha256": gold[1] == ranges_sha,
                                 "source": "tests/golden/config5_counts.json (CPU oracle over all patterns)"}
        assert gold[0] == counts_sha and gold[1] in (None, ranges_sha), \
            "results on the config-5 pattern set differ from tests/golden/config5_counts.json"
    if use_dist:
        dist_report(out, torch, dist, sharding, pipe, wl, args, world, rank, local, gloo, dt_rank, ev_ms,
                    kernel_ms_single, step)

    single = world == 1 and rank == 0 and not use_dist          # the default line: every BASELINE config
    # ---- config 2b: uniform random patterns -> the early exit of wrapper.rs:111-113 ----
    rflat = None
    if wl.dna and single and not args.no_early_exit:
        rflat = ((W.splitmix64_torch(5, 0, npat * m, dev) & 3) + 1).to(torch.uint8)
        wl.count(pat=rflat)
        torch.cuda.synchronize()
        rms, rsteps = wl.timed_kernel(lambda: wl.count(pat=rflat))
        out["early_exit"] = {"workload": "config 2b: %d uniform random len-%d patterns" % (npat, m),
                             "executed_steps": rsteps, "offered_chars": npat * m,
                             "mean_steps_per_pattern": round(rsteps / npat, 2),
                             "executed_steps_per_s": rsteps / (rms / 1e3), "kernel_ms": round(rms, 4),
                             "nonzero_counts": int((wl.d_e > wl.d_s).sum().item())}
        wl.count()                                    # restore the config-2 (s, e)
        torch.cuda.synchronize()

    lap("import + text + build + headline")
    # ---- the plain index (FMX_FLAG_PLAIN: the reference's loop step for step) and each accelerator alone: same
    # patterns, (s, e) asserted identical to the headline's on all of them ----
    if accelerated:
        out["value_auto"] = value                      # (rounds 4-5 printed the accelerated index under this key)
    if single and not args.no_accel:
        accel_legs(out, wl, args, rflat)
    del rflat
    lap("accelerators")

    # ---- locate (config 3; gathered over the ranks for N > 1) ----
    if wl.level is not None:
        locate_leg(out, wl, args, world, rank, dist, gloo, key)

    # ---- the reference's own row-order sampling on the same index (FMX_FLAG_ROW_ORDER): same positions ----
    if single and wl.dna and wl.level is not None and not args.no_accel:
        try:
            locate_row_order_leg(out, wl, args)
        except Exception as ex:  # noqa: BLE001 -- never lose the headline line to an optional leg
            out["locate_row_order"] = {"error": repr(ex)}
    lap("locate")

    # ---- beyond 2^32 rows: the 64-bit engine on the config-2 / config-3 shapes.  EARLY in the run: its builder needs
    # 137 GB of scratch beyond what the scratch cache holds, and on this runtime a process that has cycled through about
    # the device's memory pays ~30 ms per GiB for every further hipMalloc -- the legs below
    # allocate and free tens of GB between them (round 3 ran this leg last: build_ms 0.57 s in some runs, 3-8 s in others)
    if single and wl.dna and not args.no_wide and args.log2n >= 30:      # only next to the full-size configs
        try:
            wide_leg(out, args, dev)
        except Exception as ex:  # noqa: BLE001 -- never lose the headline line to an extra leg
            out["wide"] = {"error": repr(ex)}
    lap("wide")

    # ---- the config-5 step through a 1-rank RCCL communicator on this GPU (default N=1 run) ----
    if single and not args.no_rccl_check:
        try:
            rccl_1rank_leg(out, wl, args, dev, local)
        except Exception as ex:  # noqa: BLE001 -- never lose the headline line to an extra leg
            out["rccl_1rank"] = {"error": repr(ex)}
    lap("rccl_1rank + config5_g1")

    # ---- config 3b: short patterns, wide intervals ----
    if single and wl.dna and wl.level is not None and not args.no_3b:
        try:
            locate_3b(out, wl, args, key)
        except Exception as ex:  # noqa: BLE001 -- never lose the headline line to an extra leg
            out["locate_3b"] = {"error": repr(ex)}
    lap("locate_3b")

    # ---- the same batch through the host-pointer entry point (PCIe both ways) ----
    if single and not args.no_d2h:
        try:
            d2h_leg(out, wl, args)
        except Exception as ex:  # noqa: BLE001
            out["value_incl_d2h"] = None
            out["incl_d2h"] = {"error": repr(ex)}
    lap("incl_d2h")

    # ---- SURVEY 8(d)'s protocol (patterns resident, results landed in host memory) and config 5 behind the C ABI
    # (one host caller, fmx_count_batch_multi over the replicas of the index) ----
    if single and not args.no_d2h:
        try:
            results_on_host_leg(out, wl, args)
        except Exception as ex:  # noqa: BLE001
            out["results_on_host"] = {"error": repr(ex)}
    if single and wl.dna and not args.no_config5 and not args.no_d2h:
        try:
            config5_cabi_leg(out, wl, args, dev)
        except Exception as ex:  # noqa: BLE001
            out["config5_cabi"] = {"error": repr(ex)}
    lap("results_on_host + config5_cabi")

    # ---- CPU baseline of the headline: rank 0 only, after the timed regions.  At N > 1 the other ranks wait in a
    # gloo barrier (a socket wait): an RCCL barrier would have their host threads spin on a stream and take CPU time
    # from the very cores the baseline is measured on ----
    if not args.no_cpu_baseline and (single or use_dist):
        side = None
        if use_dist and world > 1:
            side = dist.group.WORLD if gloo else dist.new_group(backend="gloo")
            dist.barrier(group=side)
        if rank == 0:
            wl.count()
            torch.cuda.synchronize()
            out["cpu_baseline"] = cpu_baseline(wl, args, "rlfm" if wl.rlfm else "fm")
        if side is not None:
            dist.barrier(group=side)
    lap("cpu_baseline")

    # ---- one-at-a-time callers: latency of one pattern, break-even batch size against the one-thread CPU port ----
    if single and wl.dna and not args.no_cpu_baseline and not args.no_d2h:
        try:
            single_call_leg(out, wl, args)
        except Exception as ex:  # noqa: BLE001
            out["single_call"] = {"error": repr(ex)}
    lap("single_call")

    # ---- config 4 (RLFMIndex, sigma = 255) as its own object ----
    wr = None
    if single and wl.dna and not args.no_rlfm:
        try:
            wr = rlfm_leg(out, args, dev, local)
        except Exception as ex:  # noqa: BLE001
            out["rlfm"] = {"error": repr(ex)}
        if wr is not None:
            wr.close()
            del wr
    lap("rlfm")

    # ---- fabric traffic measured by the counter passes at the start of this run ----
    if pmc is not None and rank == 0:
        apply_pmc(out, pmc[0], pmc[1])

    # ---- Infinity-Cache A/B: the headline kernel on n = 2^31 (after apply_pmc: it compares with the headline's requests) ----
    if single and ic_ab_wanted(args, world, use_dist):
        try:
            ic_ab_leg(out, args, dev, local, pmc31)
        except Exception as ex:  # noqa: BLE001 -- never lose the headline line to an extra leg
            out["count_n31"] = {"error": repr(ex)}
    lap("count_n31")
    two_stream_roofline(out.get("locate"))
    two_stream_roofline((out.get("rlfm") or {}).get("locate"))

    if use_dist:             # every rank is done before the line is printed; nothing follows it on stdout
        dist.barrier()
        dist.destroy_process_group()
    flush_c_stdio()
    if rank == 0:
        out["leg_seconds"] = leg_seconds
        path = write_detail(out, args, world)
        print(json.dumps(headline(out, path), separators=(",", ":")))
        sys.stdout.flush()


# --------------------------------------------------------------------------------------------
# the line the driver parses: compact (< 4 KB), every BASELINE config as one scalar; the rest goes to a file
# --------------------------------------------------------------------------------------------
LINE_LIMIT = 4096


def write_detail(out, args, world):
    """the full result object -> --detail-out (default bench_detail.json next to this script); returns the path as the
    line names it, or None when it could not be written (a read-only checkout: the object then goes to stderr)"""
    path = args.detail_out or os.path.join(ROOT, "bench_detail.json" if world == 1 else "bench_detail_g%d.json" % world)
    try:
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        with open(path, "w") as f:
            json.dump(out, f, indent=1)
            f.write("\n")
        return os.path.relpath(path, ROOT) if path.startswith(ROOT + os.sep) else path
    except OSError:
        sys.stderr.write(json.dumps(out) + "\n")
        return None


def _get(d, *keys):
    for k in keys:
        if not isinstance(d, dict):
            return None
        d = d.get(k)
    return d


def _cabi_gmax(out):
    """config 5 through the C ABI at the largest G of the child run (patterns resident, counts to the host), or None"""
    pts = _get(out, "config5_cabi", "multi_device", "points") or {}
    if not pts:
        return None
    g = max(pts, key=lambda k: int(k[1:]))
    return _get(pts[g], "resident_patterns", "value")


def _sig(v, digits=4):
    """floats of the side legs to `digits` significant digits (the headline's own numbers stay exact)"""
    if isinstance(v, float) and v == v and v not in (float("inf"), float("-inf")) and v != 0.0:
        from math import floor, log10
        return round(v, digits - 1 - int(floor(log10(abs(v)))))
    return v


ROOF_KEYS = ("bound", "kernel", "avg_kernel_ms", "achieved", "peak", "unit", "frac", "traffic", "hbm_frac_min",
             "ic_hit_share_max", "frac_of_gather_ceiling", "traffic_over_min_bytes", "algorithmic_ref_bytes")
CPU_KEYS = ("value", "unit", "cores", "threads_used", "single_thread_value", "cpu_model", "kind")


def headline(out, detail_path):
    """the compact object: the contract's keys verbatim from `out`, `roofline` and `cpu_baseline` cut to their figures,
    one scalar per side leg.  Nothing is recomputed here -- every number is the one in the detail file."""
    keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data")
    h = {k: out.get(k) for k in keep}
    cfg = out.get("config") or {}
    h["config"] = {k: cfg.get(k) for k in ("workload", "text_len", "patterns_per_gpu", "pattern_len", "total_patterns",
                                            "parallelism", "index_bytes", "build_ms")}
    if isinstance(h["config"]["workload"], str):
        h["config"]["workload"] = h["config"]["workload"][:300]
    r = out.get("roofline") or {}
    h["roofline"] = {k: r.get(k) for k in ROOF_KEYS}
    h["roofline"]["basis"] = (("fabric bytes (L2 -> Infinity Cache / HBM): read requests counted by width (TCC_EA0_RDREQ_"
                               "32B/64B/128B) + WRITE_SIZE, / kernel ms" if r.get("read_request_widths") else
                               "fabric bytes (L2 -> Infinity Cache / HBM): 2 x FETCH_SIZE + WRITE_SIZE, / kernel ms")
                              if r.get("traffic") else "no PMC traffic in this run")
    cb = out.get("cpu_baseline")
    if cb:
        h["cpu_baseline"] = {k: cb.get(k) for k in CPU_KEYS}
        h["cpu_baseline"]["sample"] = (cb.get("sample") or "")[:120]
    side = {
        "counts_sha256": (out.get("counts_sha256") or "")[:16] or None,
        "matches_golden": (all(v for k, v in out["matches_golden"].items() if k.endswith("sha256"))
                           if out.get("matches_golden") else None),
        "early_exit_steps_per_s": _get(out, "early_exit", "executed_steps_per_s"),
        "locate_hits_per_s": _get(out, "locate", "hits_per_s"),
        "locate_ms_per_batch": _get(out, "locate", "ms_per_batch"),
        "locate_kernel": _get(out, "locate", "roofline", "kernel"),
        "locate_frac": _get(out, "locate", "roofline", "frac"),
        "locate_frac_of_gather_ceiling": _get(out, "locate", "roofline", "frac_of_gather_ceiling"),
        "locate_two_streams_hits_per_s": _get(out, "locate", "two_streams", "hits_per_s"),
        "locate_row_order_hits_per_s": _get(out, "locate_row_order", "hits_per_s"),
        "locate_3b_hits_per_s": _get(out, "locate_3b", "hits_per_s"),
        "locate_3b_kernel": _get(out, "locate_3b", "roofline", "kernel"),
        "locate_3b_frac": _get(out, "locate_3b", "roofline", "frac"),
        "rlfm_value": _get(out, "rlfm", "value"),
        "rlfm_frac": _get(out, "rlfm", "roofline", "frac"),
        "rlfm_value_plain": _get(out, "rlfm", "plain", "value"),
        "rlfm_index_bytes": _get(out, "rlfm", "config", "index_bytes"),
        "rlfm_locate_hits_per_s": _get(out, "rlfm", "locate", "hits_per_s"),
        "rlfm_locate_frac": _get(out, "rlfm", "locate", "roofline", "frac"),
        "rlfm_count_only_index_bytes": _get(out, "rlfm", "config", "count_only_index_bytes"),
        "rlfm_locate_run_table_hits_per_s": _get(out, "rlfm", "locate_run_table", "hits_per_s"),
        "rlfm_cpu_value": _get(out, "rlfm", "cpu_baseline", "value"),
        "count_n31_value": _get(out, "count_n31", "value"),
        "count_n31_frac": _get(out, "count_n31", "roofline", "frac"),
        "count_n31_hbm_frac_min": _get(out, "count_n31", "roofline", "hbm_frac_min"),
        "config5_g1_value": _get(out, "config5_g1", "value"),
        "config5_g1_matches_golden": _get(out, "config5_g1", "matches_golden", "counts_sha256"),
        "rccl_1rank_value": _get(out, "rccl_1rank", "value"),
        "value_plain": out.get("value_plain"),
        "plain_kernel": _get(out, "plain", "roofline", "kernel"),
        "plain_frac": _get(out, "plain", "roofline", "frac"),
        "plain_frac_of_gather_ceiling": _get(out, "plain", "roofline", "frac_of_gather_ceiling"),
        "value_auto": out.get("value_auto"),
        "value_incl_d2h": out.get("value_incl_d2h"),
        "value_results_on_host": out.get("value_results_on_host"),
        "value_ranges_on_host": out.get("value_ranges_on_host"),
        "one_pattern_call_us": _get(out, "single_call", "n_2^30", "one_pattern_us"),
        "break_even_batch": _get(out, "single_call", "n_2^30", "break_even_batch"),
        "break_even_batch_n50000": _get(out, "single_call", "n_50000", "break_even_batch"),
        "config5_cabi_g1_value": _get(out, "config5_cabi", "g1", "value"),
        "config5_cabi_g1_resident_value": _get(out, "config5_cabi", "g1_resident", "value"),
        "config5_cabi_matches_golden": _get(out, "config5_cabi", "matches_golden"),
        "config5_cabi_devices": _get(out, "config5_cabi", "multi_device", "devices_visible"),
        "config5_cabi_gmax_resident_value": _cabi_gmax(out),
        "wide_value": _get(out, "wide", "value"),
        "wide_locate_hits_per_s": _get(out, "wide", "locate", "hits_per_s"),
        "wide_build_ms": _get(out, "wide", "build_ms"),
        "rccl_ranks": out.get("rccl_ranks"), "rccl_version": out.get("rccl_version"),
        "dist_backend": out.get("dist_backend"),
        "gather_ms_max": (max(p["gather_ms"] for p in out["per_rank"]) if out.get("per_rank") else None),
        "kernel_ms_max": (max(p["kernel_ms"] for p in out["per_rank"]) if out.get("per_rank") else None),
        "pmc": _get(out, "pmc", "status"),
    }
    h.update({k: _sig(v) for k, v in side.items() if v is not None})
    errs = [k for k, v in out.items() if isinstance(v, dict) and "error" in v]
    if errs:
        h["legs_failed"] = errs
    h["run_seconds"] = round(sum(v for v in (out.get("leg_seconds") or {}).values() if isinstance(v, (int, float))), 1)
    h["detail"] = detail_path
    line = json.dumps(h, separators=(",", ":"))
    if len(line) >= LINE_LIMIT:        # never again a line the driver cannot parse: drop side legs from the end
        for k in list(side)[::-1]:
            h.pop(k, None)
            if len(json.dumps(h, separators=(",", ":"))) < LINE_LIMIT:
                break
    return h



if __name__ == "__main__":
    main()
