"""The locate legs of bench.py: config 3 (one stream, two streams, row-order index) and config 3b."""
import ctypes as C
import time

from .common import golden_locate, positions_sha256
from .roofline import WALK_KERNEL, make_roofline, run_census, stored_traffic

def dna_walk_kernel(wl):
    return WALK_KERNEL + "<4>" if wl.index.walk_records() else "fmx_locate_f3p_kernel<4>"


def dna_walk_kernel_long(wl):
    """the same one-launch kernel since round 5 (the walk is chosen per 64-hit ticket inside it)"""
    return WALK_KERNEL + "<4>" if wl.index.walk_records() else "fmx_locate_f3p_kernel<4>"


def locate_leg(out, wl, args, world, rank, dist, gloo, key, dest=None, legname="locate"):
    torch, lib = wl.torch, wl.lib
    import numpy as np
    from fm_index_amd import sharding
    dest = out if dest is None else dest
    wl.count()
    wl.prepare_locate()
    total_hits, npat, m = wl.total_hits, wl.npat, wl.m
    wl.locate()
    torch.cuda.synchronize()
    lsteps = max(3, args.steps // 2)

    use_dist = dist is not None

    # config 5: positions of every rank, in input order, on every rank.  The variable-length gather is planned once
    # (sharding.PositionGatherPlan: the counts of all ranks, the offsets and the padded buffer size -- one host
    # synchronisation; the intervals are the same in every step); a step is then locate + the gather of its positions
    # through the double-buffered pipeline of the count leg (wire dtype int32 while len < 2^31, the collective on the
    # communication stream under the next step's walk).  (Gathering the counts again in every step through a second
    # pipeline made the loop CPU-bound on one GPU: 0.21 ms per step against 0.14.)
    plan = pipe_pos = None
    if use_dist:
        cnt = (wl.d_e - wl.d_s)
        plan = sharding.PositionGatherPlan(cnt.cpu() if gloo else cnt, wl.total_patterns)
        pipe_pos = sharding.CountGatherPipeline(plan.mx, world, wl.n, wl.dev, backend="gloo" if gloo else "nccl",
                                                force_collective=True)

    def lstep():
        if use_dist:
            return pipe_pos.step(lambda out64: wl.locate(out=out64))
        wl.locate()
        return None
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(lsteps):
        g = lstep()
    if use_dist:
        pipe_pos.drain()
    torch.cuda.synchronize()
    ldt = time.perf_counter() - t0
    all_hits = total_hits
    if use_dist:
        tt = torch.tensor([ldt], dtype=torch.float64, device="cpu" if gloo else wl.dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        ldt = float(tt.item())
        # g: the padded gather of the last step (wire dtype): rank r's positions at [r * mx, r * mx + totals[r])
        all_hits = int(plan.off[-1].item())
        assert all_hits == sum(plan.totals) and plan.totals[rank] == total_hits
        wl.locate()                                     # this rank's positions once more, in its own buffer
        torch.cuda.synchronize()
        mine = g[rank * plan.mx:rank * plan.mx + total_hits].to(wl.dev).to(torch.int64)
        assert bool((mine == wl.d_pos[:total_hits]).all()), "gathered positions differ from this rank's"
        allp = torch.cat([g[r * plan.mx:r * plan.mx + plan.totals[r]] for r in range(world)]).cpu().numpy().astype(np.int64)
        if args.dump_counts and rank == 0:                # tests: every rank's positions, compacted
            np.save(args.dump_counts.replace(".npy", "_pos.npy"), allp)
    else:
        allp = wl.d_pos[:total_hits].cpu().numpy()
    # the ORDERED positions of the whole global pattern set, hashed (input order; suffix-array order within a pattern)
    pos_sha = positions_sha256(allp)
    gold = golden_locate(wl, args)
    del allp
    if gold is not None:
        assert gold["hits"] == all_hits and gold["positions_sha256"] == pos_sha, \
            "located positions differ from tests/golden/config5_counts.json (the oracle's ordered positions)"
    # the walk kernel alone, one launch at a time, HIP events on the launch stream
    kms, lf_steps = [], 0
    for _ in range(lsteps):
        ms, lf_steps = wl.timed_kernel(wl.locate)
        kms.append(ms)
    # property checks at full size: every located position really holds the pattern, and each
    # pattern's source position is among its hits
    hit_pat = torch.repeat_interleave(torch.arange(npat, device=wl.dev), wl.d_e - wl.d_s)
    chk = torch.ones(total_hits, dtype=torch.bool, device=wl.dev)
    for j in range(m):
        chk &= wl.text[wl.d_pos[:total_hits] + j] == wl.pat.view(npat, m)[hit_pat, j]
    assert bool(chk.all()), "located position does not hold the pattern"
    found_src = torch.zeros(npat, dtype=torch.bool, device=wl.dev)
    found_src[hit_pat[wl.d_pos[:total_hits] == wl.src_pos[hit_pat]]] = True
    assert bool(found_src.all()), "source position missing from locate output"
    del hit_pat, chk, found_src
    kalone_ms = sum(kms) / len(kms)
    # ... and as it runs in the timed region: launches back to back, a pair of events around every walk kernel
    kavg_ms = wl.series_kernel_ms(wl.locate, max(8, lsteps)) or kalone_ms
    cen = None
    if not args.no_census and rank == 0:
        cen = run_census(wl, lambda cl: wl.locate(lib=cl), lf_steps * (8 if wl.rlfm else 2) + 4 * total_hits + (1 << 20))
    ref_bytes = lf_steps * wl.Lbits * 64 + total_hits * 64   # steps*L*64 + 64 per hit
    # RLFM: batches of two or more hits per pattern on an index with the run table take the lane walk in rounds
    rl_lane = wl.rlfm and wl.index.walk_records() and total_hits >= 2 * npat and total_hits >= (1 << 18)
    kname = dna_walk_kernel(wl) if wl.dna else (("fmx_locate_rl_rounds_kernel" if rl_lane else "fmx_locate_ep_kernel")
                                                if wl.rlfm else "fmx_locate_kernel<FMX_KIND_FM>")
    # streamed bytes: the positions, and the rows array (read) -- or, for the one-launch DNA kernel, s / e / off
    lstream = total_hits * 8 + (npat * 24 if (wl.dna and wl.index.walk_records()) else total_hits * 4)
    roof = make_roofline(kname, kavg_ms, 1, ref_bytes, lstream, cen,
                         stored_traffic(key, "locate"), table_bytes=wl.locate_table_bytes())
    two = None
    if not use_dist:
        try:
            two = locate_two_streams(wl, max(8, lsteps))
        except Exception as ex:  # noqa: BLE001 -- never lose the leg to its extra measurement
            two = {"error": repr(ex)}
    dest[legname] = {"hits_per_s": all_hits * lsteps / ldt, "hits": all_hits, "hits_per_gpu": total_hits,
                     "lf_steps": lf_steps, "level": wl.level, "ms_per_batch": ldt / lsteps * 1e3,
                     "sampling": "text order" + (" + walk records" if wl.index.walk_records() else "")
                     if wl.index.text_order() else "row order",
                     "includes": "row expansion + walk" + (" + gather of counts and positions over the ranks"
                                                           if use_dist else ""),
                     "walk_kernel_ms": round(kavg_ms, 4), "walk_kernel_ms_launched_alone": round(kalone_ms, 4),
                     "positions_sha256": pos_sha,
                     "matches_golden": ({"positions_sha256": True, "source": "tests/golden/config5_counts.json (the CPU oracle's "
                                         "ordered positions of every pattern)"} if gold is not None else None),
                     "roofline": roof}
    if two is not None:
        dest[legname]["two_streams"] = two


def locate_row_order_leg(out, wl, args):
    """config 3 on an index built with FMX_FLAG_ROW_ORDER: the reference's own sampling (the rows i with i mod 2^level
    == 0, sample.rs:21-44) and its geometric walks (fmx_locate_f3p_kernel) -- the default until round 3.  Positions must
    equal the default index's (text order + walk records) on every hit."""
    torch, F, lib = wl.torch, wl.F, wl.lib
    tix = F.FMIndexWithLocate.from_device_text(wl.text.data_ptr(), wl.n, wl.maxc, level=wl.level, device=wl.local,
                                               sampling="row")
    try:
        assert not tix.text_order() and not tix.walk_records()
        total, npat = wl.total_hits, wl.npat
        pos = torch.empty(max(total, 1), dtype=torch.int64, device=wl.dev)

        def lstep():
            rc = lib.fmx_locate_batch_dev(tix.handle(), C.c_void_p(wl.d_s.data_ptr()), C.c_void_p(wl.d_e.data_ptr()),
                                          npat, C.c_void_p(wl.d_off.data_ptr()), total, C.c_void_p(pos.data_ptr()), wl.sp)
            assert rc == 0
        for _ in range(3):
            lstep()
        torch.cuda.synchronize()
        reps = max(5, args.steps // 2)
        t0 = time.perf_counter()
        for _ in range(reps):
            lstep()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        lib.fmx_set_timing(tix.handle(), 1)
        lstep()
        torch.cuda.synchronize()
        kms, steps = lib.fmx_last_kernel_ms(tix.handle()), int(lib.fmx_last_steps(tix.handle()))
        lib.fmx_set_timing(tix.handle(), 0)
        assert bool((pos[:total] == wl.d_pos[:total]).all()), "row-order index locates differently"
        out["locate_row_order"] = {"hits_per_s": total / dt, "ms_per_batch": dt * 1e3, "walk_kernel_ms": round(kms, 4),
                                   "hits": total, "lf_steps": steps, "index_bytes": tix.heap_size(),
                                   "default_index_bytes": wl.index.heap_size(),
                                   "build_ms": round(float(lib.fmx_build_ms(tix.handle())), 1),
                                   "note": "FMX_FLAG_ROW_ORDER on the config-3 index (SOSampledSuffixArray's own rows, "
                                           "sample.rs:21-44); positions identical to the default index on every hit"}
    finally:
        tix.close()



def locate_two_streams(wl, reps):
    """the same batch alternating between two streams through the caller-workspace entry point
    (fmx_locate_batch_ws_dev: kernel launches only, nothing shared between the streams but the index), so
    that one batch's longest walks run under the next batch's bulk.  Positions of both streams must equal
    the single-stream result."""
    torch, lib = wl.torch, wl.lib
    total, npat = wl.total_hits, wl.npat
    wsb = int(lib.fmx_locate_workspace_bytes(wl.h, total))
    streams = [torch.cuda.Stream(device=wl.dev), torch.cuda.Stream(device=wl.dev)]
    ws = [torch.empty(wsb, dtype=torch.uint8, device=wl.dev) for _ in range(2)]
    pos = [torch.empty(max(total, 1), dtype=torch.int64, device=wl.dev) for _ in range(2)]
    torch.cuda.synchronize()

    def launch(i):
        rc = lib.fmx_locate_batch_ws_dev(wl.h, C.c_void_p(wl.d_s.data_ptr()), C.c_void_p(wl.d_e.data_ptr()), npat,
                                         C.c_void_p(wl.d_off.data_ptr()), total, C.c_void_p(pos[i].data_ptr()),
                                         C.c_void_p(ws[i].data_ptr()), wsb, C.c_void_p(streams[i].cuda_stream))
        if rc != 0:
            raise RuntimeError(lib.fmx_last_error().decode())
    for i in (0, 1, 0, 1):
        launch(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for r in range(2 * reps):
        launch(r & 1)
    torch.cuda.synchronize()
    dt2 = (time.perf_counter() - t0) / (2 * reps)
    # one stream, same entry point (what the workspace form alone buys)
    t0 = time.perf_counter()
    for r in range(2 * reps):
        launch(0)
    torch.cuda.synchronize()
    dt1 = (time.perf_counter() - t0) / (2 * reps)
    ok = bool((pos[0][:total] == wl.d_pos[:total]).all()) and bool((pos[1][:total] == wl.d_pos[:total]).all())
    assert ok, "workspace-form locate differs from fmx_locate_batch_dev"
    return {"ms_per_batch": dt2 * 1e3, "hits_per_s": total / dt2, "one_stream_ws_ms_per_batch": dt1 * 1e3,
            "one_stream_ws_hits_per_s": total / dt1, "workspace_bytes": wsb,
            "note": "fmx_locate_batch_ws_dev, batches alternating between two streams with their own workspace "
                    "and output; positions identical to fmx_locate_batch_dev"}


def setup_3b(wl):
    """config 3b: 64 K patterns of length 8-12 -> counts of 2^6..2^14, wide [s, e).
    Returns the tensors and a closure that locates the whole batch once."""
    torch, lib, W = wl.torch, wl.lib, wl.W
    npat = 1 << 16
    z = W.splitmix64_torch(11, 0, npat, wl.dev)
    lens = 8 + W.umod_torch(z, 5)
    off = torch.zeros(npat + 1, dtype=torch.int64, device=wl.dev)
    off[1:] = torch.cumsum(lens, 0)
    src = W.umod_torch(W.splitmix64_torch(12, 0, npat, wl.dev), wl.n - 1 - 12)
    tot = int(off[-1].item())
    which = torch.repeat_interleave(torch.arange(npat, device=wl.dev), lens)
    within = torch.arange(tot, device=wl.dev) - off[which]
    pat = wl.text[src[which] + within].contiguous()
    s = torch.empty(npat, dtype=torch.int64, device=wl.dev)
    e = torch.empty(npat, dtype=torch.int64, device=wl.dev)
    rc = lib.fmx_count_batch_dev(wl.h, C.c_void_p(pat.data_ptr()), C.c_void_p(off.data_ptr()), npat, None,
                                 C.c_void_p(s.data_ptr()), C.c_void_p(e.data_ptr()), None, wl.sp)
    assert rc == 0
    hoff = torch.empty(npat + 1, dtype=torch.int64, device=wl.dev)
    lib.fmx_offsets_dev(wl.h, C.c_void_p(s.data_ptr()), C.c_void_p(e.data_ptr()), npat,
                        C.c_void_p(hoff.data_ptr()), wl.sp)
    total = int(hoff[-1].item())
    pos = torch.empty(max(total, 1), dtype=torch.int64, device=wl.dev)

    def lstep():
        rc = lib.fmx_locate_batch_dev(wl.h, C.c_void_p(s.data_ptr()), C.c_void_p(e.data_ptr()), npat,
                                      C.c_void_p(hoff.data_ptr()), total, C.c_void_p(pos.data_ptr()), wl.sp)
        assert rc == 0
    return npat, pat, off, s, e, total, pos, lstep


def locate_3b(out, wl, args, key):
    torch = wl.torch
    npat, pat, off, s, e, total, pos, lstep = setup_3b(wl)
    lstep()
    torch.cuda.synchronize()
    reps = 3
    t0 = time.perf_counter()
    for _ in range(reps):
        lstep()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    kms, lf_steps = wl.timed_kernel(lstep)
    kms = wl.series_kernel_ms(lstep, 3) or kms
    # every located position holds its pattern's first 8 symbols; positions of a pattern are distinct
    hp = torch.repeat_interleave(torch.arange(npat, device=wl.dev), e - s)
    ok = torch.ones(total, dtype=torch.bool, device=wl.dev)
    for j in range(8):
        ok &= wl.text[pos[:total] + j] == pat[off[hp] + j]
    assert bool(ok.all()), "3b: located position does not hold the pattern"
    cnts = (e - s)
    out["locate_3b"] = {"workload": "config 3b: %d substring patterns of length 8-12" % npat,
                        "hits": total, "hits_per_s": total / dt, "ms_per_batch": dt * 1e3,
                        "walk_kernel_ms": round(kms, 4), "lf_steps": lf_steps,
                        # one record line per LF step + one sample per hit (what the census counts for
                        # config 3: requested_lines == lf_steps + hits).  NOT fabric requests: the hits of a
                        # pattern are adjacent rows, LF keeps rows of one symbol adjacent, so many of these
                        # lines are L1 / L2 hits and the rate may exceed the 55 G/s random-request ceiling
                        "requested_lines": lf_steps + total,
                        "requested_lines_per_s": (lf_steps + total) / (kms / 1e3),
                        "count_min": int(cnts.min().item()), "count_median": int(cnts.median().item()),
                        "count_max": int(cnts.max().item())}
    # fabric traffic of this launch (told from the config-3 launches of the same kernel by its grid)
    # request widths by construction (what the census counts for config 3): one record per LF step, one sample per hit
    # (walk records: max(phase, 1) records per hit, phase = position mod 2^level -- one more than the LF steps for the
    # hits that sit on a sampled position)
    nrec = lf_steps + (int(((pos[:total] & ((1 << wl.level) - 1)) == 0).sum().item()) if wl.index.walk_records() else 0)
    widths = {"requested_lines": nrec + total, "requested_records": nrec, "requested_probes": total,
              "distinct_lines": None}
    if wl.index.walk_records():
        # tickets of adjacent rows are walked a lane per hit: a record is read as lane-wise 16-byte pieces -- the row's
        # own, the pieces in front of it (3 on average), the counter's -- so every request is a probe: about 5 per record
        # visit, one per sample; plus 24 bytes of s / e / off per pattern a slice's expansion looks at
        widths = {"requested_lines": 5 * nrec + total, "requested_records": 0, "requested_probes": 5 * nrec + total,
                  "distinct_lines": None}
    out["locate_3b"]["requested_lines"] = widths["requested_lines"]
    out["locate_3b"]["requested_lines_per_s"] = widths["requested_lines"] / (kms / 1e3)
    out["locate_3b"]["bound"] = ("vector-ALU issue, not memory: the hits of a pattern are adjacent rows, and LF keeps rows of one symbol "
                                 "adjacent -- their records (and, in text order, their samples: consecutive entries) come from "
                                 "the caches (L2 hit 0.71), few requests reach the fabric (roofline.fabric_requests, frac ~0.1); "
                                 "VALU busy 0.80 at 517 vector instructions per hit (profiles/r05/kernel_pmc_dna*.json).  Round "
                                 "5: one launch (fmx_locate_f3u_kernel expands its slices itself), the walk chosen per 64-hit "
                                 "ticket, a record's pieces requested at once, the lane walk in rounds of one record visit with "
                                 "the unfinished walks compacted in LDS")
    out["locate_3b"]["roofline"] = make_roofline(dna_walk_kernel_long(wl), kms, 1, lf_steps * wl.Lbits * 64 + total * 64,
                                                 total * 8 + npat * 24, widths, stored_traffic(key, "locate_3b"),
                                                 table_bytes=wl.locate_table_bytes())

