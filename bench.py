#!/usr/bin/env python3
"""bench.py -- BASELINE.json metric on MI355X: pattern-chars/s of batched backward search
(count) on a 1 GB sigma=4 DNA text (config 2: FMIndex, 1 Mi length-32 patterns that are
substrings of the text, so all 32 steps execute), plus locate hits/s (config 3) as an
extra field.  One process per GPU; for N > 1 the patterns are sharded contiguously
(N x 1 Mi patterns, weak scaling), the index is replicated, and the counts are gathered
with one RCCL all-gather inside the timed region (config 5).

A "step" = one pass of the count kernel over this rank's pattern batch, inputs and
outputs resident in HBM.  Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# algorithmic bytes: one 512-bit block per level per endpoint
BYTES_PER_CHAR_L3 = 2 * 3 * 64      # 384 B per executed backward-search step at L = 3
HBM_PEAK_GBS = 8000.0               # MI355X_MICROARCH.md: 8.0 TB/s spec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--log2n", type=int, default=30, help="text length 2^k incl. terminator")
    ap.add_argument("--npat", type=int, default=1 << 20, help="patterns per GPU")
    ap.add_argument("--plen", type=int, default=32)
    ap.add_argument("--level", type=int, default=2, help="SA sampling level for the locate leg")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-locate", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--workload", default="dna", choices=["dna", "bytes-fm", "bytes-rlfm", "rep-fm", "rep-rlfm"],
                    help="dna = config 2/3 (headline); bytes-fm / bytes-rlfm = config 4 text "
                         "(sigma=255, L=8, len-16 patterns) on FMIndex / RLFMIndex; rep-* = config 4b: "
                         "1 MiB random block repeated with 1 % point mutations (the case RLFM exists for)")
    ap.add_argument("--mut-per-1024", type=int, default=10,
                    help="rep-* workloads: point mutations per 1024 symbols (10 = the 1 %% of config 4b)")
    ap.add_argument("--pair-index", action="store_true",
                    help="also build the opt-in 2-step index (FMX_FLAG_PAIR_INDEX) and report its "
                         "count rate in an extra 'pair_index' object (the headline stays 1-step)")
    ap.add_argument("--kmer-table", action="store_true",
                    help="also build the opt-in k-mer start table (FMX_FLAG_KMER_TABLE) and report "
                         "its count rate in 'kmer_table' (and with --pair-index the combination in "
                         "'kmer_table+pair_index'); the headline stays the plain index")
    ap.add_argument("--no-accel", action="store_true",
                    help="skip the extra legs that time the opt-in indexes (pair index, k-mer start table); "
                         "by default they are built and reported next to the plain-index headline")
    ap.add_argument("--no-early-exit", action="store_true",
                    help="skip the config-2b (uniform random patterns) legs, so that a profile of this "
                         "run holds only config-2 launches of the count kernels")
    ap.add_argument("--dist-backend", default="nccl",
                    help="nccl (= RCCL, the real path) | gloo (single-GPU rehearsal of the N>1 code "
                         "path: all ranks share cuda:0 and gather through host memory)")
    args = ap.parse_args()

    import torch
    import numpy as np
    import fm_index_amd as F
    from fm_index_amd import workload as W
    from fm_index_amd import _lib as L

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dist_backend == "gloo":
            local = 0
            torch.cuda.set_device(0)
            dist.init_process_group("gloo")
        else:
            torch.cuda.set_device(local)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    else:
        dist = None
        torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    lib = L.lib()

    n = 1 << args.log2n
    npat, m = args.npat, args.plen
    dna = args.workload == "dna"
    if not dna and args.plen == 32:
        m = 16                                   # config 4: length-16 patterns
    maxc = 4 if dna else 255
    Lbits = 3 if dna else 8
    # 2 endpoints x L levels x 64 B (FM); 2 x (2L+4) probes x 64 B (RLFM)
    rlfm = args.workload.endswith("rlfm")
    bytes_per_char = 2 * (2 * Lbits + 4) * 64 if rlfm else 2 * Lbits * 64
    # ---- synthetic inputs: text seed 1, patterns seed 3 / 7 ----
    t0 = time.time()
    if dna:
        text = W.dna_text_torch(n, 1, dev)
    elif args.workload.startswith("rep"):
        text = W.repetitive_text_torch(n, 5, dev, base_len=1 << 20, mut_per_1024=args.mut_per_1024)
    else:
        text = W.byte_text_torch(n, 4, dev)
    torch.cuda.synchronize()
    t_gen = time.time() - t0
    level = None if args.no_locate else args.level
    # a throw-away 4 KiB build first: runtime / code-object initialisation is not index construction
    F.FMIndex(F.Text.with_max_character(W.dna_text_np(4096, 9), 4), device=local).close()
    if rlfm:
        cls = F.RLFMIndexWithLocate if level is not None else F.RLFMIndex
    else:
        cls = F.FMIndexWithLocate if level is not None else F.FMIndex
    index = cls.from_device_text(text.data_ptr(), n, maxc, level=level, device=local)
    build_ms = lib.fmx_build_ms(index.handle())
    # global pattern set = world * npat substrings; this rank owns a contiguous shard
    seed = (3 if world == 1 else 7) if dna else 6
    total_pat = npat * world
    z = W.splitmix64_torch(seed, rank * npat, npat, dev)
    pos = W.umod_torch(z, n - 1 - m)
    idx2d = pos[:, None] + torch.arange(m, dtype=torch.int64, device=dev)[None, :]
    pat = text[idx2d].reshape(-1).contiguous()
    off = (torch.arange(npat + 1, dtype=torch.int64, device=dev) * m).contiguous()
    del idx2d
    d_s = torch.empty(npat, dtype=torch.int64, device=dev)
    d_e = torch.empty(npat, dtype=torch.int64, device=dev)
    from fm_index_amd import sharding
    stream = torch.cuda.current_stream()
    sp = C.c_void_p(stream.cuda_stream)
    h = index.handle()

    # N > 1 (config 5): the per-pattern counts of every rank are all-gathered over RCCL/xGMI.
    # The gather of step k overlaps the search kernel of step k+1: two result buffers alternate,
    # the collective is issued async (it runs on RCCL's stream once the kernel that produced its
    # input has finished) and a buffer is only reused after its gather has completed.  Counts
    # travel as int32 (n < 2^32): 4 MiB per rank per step instead of 8.
    pipelined = world > 1 and args.dist_backend == "nccl" and not os.environ.get("FMX_BENCH_SYNC_GATHER")
    nbuf = 2 if pipelined else 1
    d_cs = [torch.empty(npat, dtype=torch.int64, device=dev) for _ in range(nbuf)]
    d_c32 = [torch.empty(npat, dtype=torch.int32, device=dev) for _ in range(nbuf)]
    g_out = [torch.empty(total_pat, dtype=torch.int32, device=dev) for _ in range(nbuf)] if world > 1 else []
    pending = [None] * nbuf
    d_c = d_cs[0]
    step_no = [0]

    def step():
        b = step_no[0] % nbuf
        step_no[0] += 1
        if pending[b] is not None:          # this buffer's previous gather must be done
            pending[b].wait()
            pending[b] = None
        rc = lib.fmx_count_batch_dev(h, C.c_void_p(pat.data_ptr()), C.c_void_p(off.data_ptr()), npat,
                                     None, C.c_void_p(d_s.data_ptr()), C.c_void_p(d_e.data_ptr()),
                                     C.c_void_p(d_cs[b].data_ptr()), sp)
        if rc != 0:
            raise RuntimeError(lib.fmx_last_error().decode())
        if world > 1:
            if args.dist_backend == "gloo":     # single-GPU rehearsal of the control flow
                return sharding.gather_counts(d_cs[b].cpu(), total_pat)
            d_c32[b].copy_(d_cs[b])
            if pipelined:
                try:
                    pending[b] = dist.all_gather_into_tensor(g_out[b], d_c32[b], async_op=True)
                except Exception:   # noqa: BLE001 -- fall back to the blocking collective
                    pending[b] = None
                    dist.all_gather_into_tensor(g_out[b], d_c32[b])
            else:
                dist.all_gather_into_tensor(g_out[b], d_c32[b])
            return g_out[b]
        return d_cs[b]

    def drain():
        for b in range(nbuf):
            if pending[b] is not None:
                pending[b].wait()
                pending[b] = None

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    drain()
    barrier()
    # kernel-only time over the timed region: HIP events on the launch stream
    ev0 = torch.cuda.Event(enable_timing=True)
    ev1 = torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record(stream)
    for k in range(args.steps):
        step()
    drain()                      # every gather of the timed steps has completed
    ev1.record(stream)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64,
                          device="cpu" if args.dist_backend == "gloo" else dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    ev_ms = ev0.elapsed_time(ev1)
    assert lib.fmx_stream_status(h) == 0

    # ---- validation + step census (outside the timed region) ----
    lib.fmx_set_timing(h, 1)
    step_no[0] = 0
    last = step()
    drain()
    torch.cuda.synchronize()
    if world > 1 and args.dist_backend == "nccl":
        # gathered counts: this rank's shard sits at [rank*npat, (rank+1)*npat) and equals its own
        assert bool((last[rank * npat:(rank + 1) * npat].to(torch.int64) == d_cs[0]).all())
    kernel_ms_single = lib.fmx_last_kernel_ms(h)
    steps_exec = int(lib.fmx_last_steps(h))
    lib.fmx_set_timing(h, 0)
    assert steps_exec == npat * m, (steps_exec, npat * m)   # substrings: every step executes
    assert bool((d_c >= 1).all()), "a substring of the text must occur at least once"
    # every pattern's own source position must lie in its SA interval's located set (below)

    chars_per_step_rank = npat * m
    value = chars_per_step_rank * world * args.steps / dt
    # dominant kernel: fmx_count_kernel; avg launch duration from the event bracket of the timed
    # region at N=1 (launches are back to back on one stream); per-launch event at N>1
    avg_kernel_s = (ev_ms / 1e3) / args.steps if world == 1 else kernel_ms_single / 1e3
    achieved = chars_per_step_rank * bytes_per_char / avg_kernel_s / 1e9
    # HBM-side bytes per launch: PMC counters of the same command, collected in separate
    # rocprofv3 passes and corrected as profiles/README.md describes (None when not profiled)
    traffic = {}
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            traffic = json.load(f).get("%s:%d:%d:%d" % (args.workload, npat, m, args.log2n), {})
    except OSError:
        pass
    kname = "fmx_count_f3_kernel<1,false,false>" if dna else \
        ("fmx_count_kernel<FMX_KIND_RLFM>" if rlfm else "fmx_count_kernel<FMX_KIND_FM>")
    roofline = {"bound": "hbm", "kernel": kname, "achieved": round(achieved, 1),
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                "traffic": traffic.get("count", {}).get("bytes"),
                "algorithmic_bytes_per_launch": chars_per_step_rank * bytes_per_char,
                "algorithmic_bytes_per_char": bytes_per_char,
                "avg_kernel_ms": round(avg_kernel_s * 1e3, 4)}
    if roofline["traffic"]:
        # what the memory system really moved: measured HBM-side bytes / kernel time (frac above is
        # the contract's algorithmic-bytes figure and exceeds 1 because one 128-B record answers
        # what the reference reads three 64-B blocks for)
        roofline["traffic_rate"] = round(roofline["traffic"] / avg_kernel_s / 1e9, 1)
        roofline["traffic_frac"] = round(roofline["traffic"] / avg_kernel_s / 1e9 / HBM_PEAK_GBS, 4)

    out = {
        # BASELINE.json's metric, verbatim; `value` is its count half (pattern-chars/s), the locate
        # half (hits/s) is out["locate"]["hits_per_s"]
        "metric": "pattern-chars/sec backward search (count) + locate hits/sec, 1 GB text",
        "value": value, "unit": "pattern-chars/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "u32", "data": "synthetic",
        "config": {"workload": ("config2: FMIndex count, n=2^%d sigma=4 DNA text (L=3), %d x len-%d "
                                "substring patterns per GPU" % (args.log2n, npat, m)) if dna else
                               ("config4 (%s): n=2^%d sigma=255 byte text (L=8), %d x len-%d substring "
                                "patterns per GPU" % (args.workload, args.log2n, npat, m)),
                   "text_len": n, "patterns_per_gpu": npat, "pattern_len": m,
                   "parallelism": "patterns sharded x%d, index replicated" % world,
                   "index_bytes": index.heap_size(), "build_ms": round(build_ms, 1),
                   "textgen_s": round(t_gen, 2)},
        "roofline": roofline,
    }

    # ---- config 2b: uniform random patterns -> the early exit of wrapper.rs:111-113 ----
    if dna and rank == 0 and not args.no_early_exit:
        rflat = ((W.splitmix64_torch(5, 0, npat * m, dev) & 3) + 1).to(torch.uint8)
        rs_ = torch.empty(npat, dtype=torch.int64, device=dev)
        re_ = torch.empty(npat, dtype=torch.int64, device=dev)

        def rstep():
            rc = lib.fmx_count_batch_dev(h, C.c_void_p(rflat.data_ptr()), C.c_void_p(off.data_ptr()), npat,
                                         None, C.c_void_p(rs_.data_ptr()), C.c_void_p(re_.data_ptr()), None, sp)
            assert rc == 0
        rstep()
        torch.cuda.synchronize()
        lib.fmx_set_timing(h, 1)
        rstep()
        torch.cuda.synchronize()
        rms = lib.fmx_last_kernel_ms(h)
        rsteps = int(lib.fmx_last_steps(h))
        lib.fmx_set_timing(h, 0)
        out["early_exit"] = {"workload": "config 2b: %d uniform random len-%d patterns" % (npat, m),
                             "executed_steps": rsteps, "offered_chars": npat * m,
                             "mean_steps_per_pattern": round(rsteps / npat, 2),
                             "executed_steps_per_s": rsteps / (rms / 1e3), "kernel_ms": round(rms, 4),
                             "nonzero_counts": int((re_ > rs_).sum().item())}
        del rflat, rs_, re_

    # ---- opt-in accelerators: same patterns, results asserted identical to the plain index ----
    legs = []
    if not args.no_accel:
        args.pair_index = args.kmer_table = True
    if args.pair_index and dna:
        legs.append(("pair_index", dict(pair_index=True), "opt-in FMX_FLAG_PAIR_INDEX"))
    if args.kmer_table:
        legs.append(("kmer_table", dict(kmer_table=True), "opt-in FMX_FLAG_KMER_TABLE"))
    if args.kmer_table and args.pair_index and dna:
        legs.append(("kmer_table+pair_index", dict(kmer_table=True, pair_index=True),
                     "FMX_FLAG_KMER_TABLE | FMX_FLAG_PAIR_INDEX"))
    for leg_name, leg_kw, leg_note in legs:
        try:
            pidx = (F.RLFMIndex if rlfm else F.FMIndex).from_device_text(text.data_ptr(), n, maxc,
                                                                          device=local, **leg_kw)
            assert pidx.has_pair_index() == bool(leg_kw.get("pair_index"))
            if leg_kw.get("kmer_table") and pidx.kmer_k() == 0:
                out[leg_name] = {"skipped": "FMX_FLAG_KMER_TABLE is ignored for this kind / alphabet"}
                pidx.close()
                continue
            ps = torch.empty(npat, dtype=torch.int64, device=dev)
            pe = torch.empty(npat, dtype=torch.int64, device=dev)

            def pstep():
                rc = lib.fmx_count_batch_dev(pidx.handle(), C.c_void_p(pat.data_ptr()),
                                             C.c_void_p(off.data_ptr()), npat, None,
                                             C.c_void_p(ps.data_ptr()), C.c_void_p(pe.data_ptr()), None, sp)
                assert rc == 0
            for _ in range(args.warmup):
                pstep()
            torch.cuda.synchronize()
            p0 = torch.cuda.Event(enable_timing=True)
            p1 = torch.cuda.Event(enable_timing=True)
            p0.record(stream)
            for _ in range(args.steps):
                pstep()
            p1.record(stream)
            torch.cuda.synchronize()
            pms = p0.elapsed_time(p1) / args.steps
            assert bool((ps == d_s).all()) and bool((pe == d_e).all()), leg_name + " != plain index"
            out[leg_name] = {"value": chars_per_step_rank / (pms / 1e3), "unit": "pattern-chars/s",
                             "ms_per_step": pms, "index_bytes": pidx.heap_size(), "kmer_k": pidx.kmer_k(),
                             "build_ms": round(float(lib.fmx_build_ms(pidx.handle())), 1),
                             "traffic": traffic.get(leg_name, {}).get("bytes"),
                             "note": leg_note + "; (s,e) identical to the plain-index run; rate of rank 0's "
                                     "shard alone (not aggregated over ranks)"}
            if not args.no_early_exit and "early_exit" in out:
                # config 2b patterns (uniform random, mostly absent) through the same index
                rflat2 = ((W.splitmix64_torch(5, 0, npat * m, dev) & 3) + 1).to(torch.uint8)
                lib.fmx_count_batch_dev(h, C.c_void_p(rflat2.data_ptr()), C.c_void_p(off.data_ptr()), npat, None,
                                        C.c_void_p(d_s.data_ptr()), C.c_void_p(d_e.data_ptr()), None, sp)

                def rstep():
                    rc = lib.fmx_count_batch_dev(pidx.handle(), C.c_void_p(rflat2.data_ptr()),
                                                 C.c_void_p(off.data_ptr()), npat, None,
                                                 C.c_void_p(ps.data_ptr()), C.c_void_p(pe.data_ptr()), None, sp)
                    assert rc == 0
                for _ in range(args.warmup):
                    rstep()
                torch.cuda.synchronize()
                p0.record(stream)
                for _ in range(args.steps):
                    rstep()
                p1.record(stream)
                torch.cuda.synchronize()
                rms2 = p0.elapsed_time(p1) / args.steps
                assert bool((ps == d_s).all()) and bool((pe == d_e).all()), leg_name + " != plain index (2b)"
                out[leg_name]["early_exit_ms_per_step"] = rms2
                out[leg_name]["early_exit_offered_chars_per_s"] = npat * m / (rms2 / 1e3)
                # restore the config-2 (s, e) the locate leg starts from
                lib.fmx_count_batch_dev(h, C.c_void_p(pat.data_ptr()), C.c_void_p(off.data_ptr()), npat, None,
                                        C.c_void_p(d_s.data_ptr()), C.c_void_p(d_e.data_ptr()), None, sp)
                torch.cuda.synchronize()
                del rflat2
            pidx.close()
        except Exception as ex:  # never lose the headline line to an optional leg
            out[leg_name] = {"error": repr(ex)}

    # ---- locate leg (config 3), rank 0 reports ----
    if level is not None:
        d_off = torch.empty(npat + 1, dtype=torch.int64, device=dev)
        lib.fmx_offsets_dev(h, C.c_void_p(d_s.data_ptr()), C.c_void_p(d_e.data_ptr()), npat,
                            C.c_void_p(d_off.data_ptr()), sp)
        total_hits = int(d_off[-1].item())
        d_pos = torch.empty(max(total_hits, 1), dtype=torch.int64, device=dev)

        def locate_step():
            rc = lib.fmx_locate_batch_dev(h, C.c_void_p(d_s.data_ptr()), C.c_void_p(d_e.data_ptr()),
                                          npat, C.c_void_p(d_off.data_ptr()), total_hits,
                                          C.c_void_p(d_pos.data_ptr()), sp)
            if rc != 0:
                raise RuntimeError(lib.fmx_last_error().decode())
        locate_step()
        torch.cuda.synchronize()
        lsteps = max(3, args.steps // 2)
        # whole batches back to back (expand + walk kernels, stream-ordered scratch): wall time
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(lsteps):
            locate_step()
        torch.cuda.synchronize()
        ldt = time.perf_counter() - t0
        # the walk kernel alone, one launch at a time, HIP events on the launch stream
        lib.fmx_set_timing(h, 1)
        kms = []
        for _ in range(lsteps):
            locate_step()
            torch.cuda.synchronize()
            kms.append(lib.fmx_last_kernel_ms(h))
        lf_steps = int(lib.fmx_last_steps(h))
        lib.fmx_set_timing(h, 0)
        # property checks at full size: every located position really holds the pattern, and
        # each pattern's source position is among its hits
        hit_pat = torch.repeat_interleave(torch.arange(npat, device=dev), d_c)
        chk = torch.ones(total_hits, dtype=torch.bool, device=dev)
        for j in range(m):
            chk &= text[d_pos[:total_hits] + j] == pat.view(npat, m)[hit_pat, j]
        assert bool(chk.all()), "located position does not hold the pattern"
        found_src = torch.zeros(npat, dtype=torch.bool, device=dev)
        found_src[hit_pat[d_pos[:total_hits] == pos[hit_pat]]] = True
        assert bool(found_src.all()), "source position missing from locate output"
        kavg = sum(kms) / len(kms) / 1e3
        lbytes = lf_steps * Lbits * 64 + total_hits * 64   # steps*L*64 + 64 per hit
        out["locate"] = {"hits_per_s": total_hits * lsteps / ldt, "hits": total_hits,
                         "lf_steps": lf_steps, "level": args.level,
                         "ms_per_batch": ldt / lsteps * 1e3,
                         "roofline": {"bound": "hbm",
                                      "kernel": "fmx_locate_f3w_kernel<4>" if dna else
                                      ("fmx_locate_kernel<FMX_KIND_RLFM>" if rlfm else
                                       "fmx_locate_kernel<FMX_KIND_FM>"),
                                      "achieved": round(lbytes / kavg / 1e9, 1), "peak": HBM_PEAK_GBS,
                                      "unit": "GB/s", "frac": round(lbytes / kavg / 1e9 / HBM_PEAK_GBS, 4),
                                      "avg_kernel_ms": round(kavg * 1e3, 4),
                                      "traffic": traffic.get("locate", {}).get("bytes")}}
        del hit_pat, chk, found_src

    # ---- CPU baseline: the oracle (port of the reference algorithm) on this box's cores ----
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import fm_oracle as O
        t0 = time.time()
        bwt = index.export_bwt()
        cs = index.export_cs()
        oi = O.OracleIndex.from_bwt(bwt, cs, maxc, native=True)
        del bwt
        t_ob = time.time() - t0
        cores = os.cpu_count() or 1
        pat_h = pat.cpu().numpy()
        s_h = d_s.cpu().numpy().view(np.uint64)
        e_h = d_e.cpu().numpy().view(np.uint64)

        def cpu_run(k, threads):
            offk = np.arange(k + 1, dtype=np.uint64) * np.uint64(m)
            t = time.perf_counter()
            so, eo = oi.count_batch(pat_h[:k * m], offk, nthreads=threads)
            return time.perf_counter() - t, so, eo
        k0 = 1 << 14
        t_probe, so, eo = cpu_run(k0, cores)
        k = int(min(npat, max(k0, k0 * args.cpu_seconds / max(t_probe, 1e-6))))
        t_all, so, eo = cpu_run(k, cores)
        assert (so == s_h[:k]).all() and (eo == e_h[:k]).all(), "GPU != oracle on the CPU sample"
        times = [t_all]
        while sum(times) < args.cpu_seconds and len(times) < 25:   # ~10-30 s of CPU work in total
            times.append(cpu_run(k, cores)[0])
        t_all = sorted(times)[len(times) // 2]
        k1 = max(1024, k // cores)
        t_one, _, _ = cpu_run(k1, 1)
        out["cpu_baseline"] = {"value": k * m / t_all, "unit": "pattern-chars/s", "cores": cores,
                               "kind": "port",
                               "sample": "first %d of the %d patterns (same text, same index), "
                                         "median of %d runs of %.2f s; GPU (s,e) bit-identical on the sample" % (k, npat, len(times), t_all),
                               "single_thread_value": k1 * m / t_one,
                               "oracle_build_s": round(t_ob, 1)}
        oi.close()

    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
