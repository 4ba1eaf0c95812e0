"""The other objects of bench.py's default run: opt-in accelerators, the host-pointer entry point, the 64-bit
engine at n = 2^32 + 2^20, and config 4 (RLFMIndex)."""
import ctypes as C
import time

from .common import PRETOUCH, Workload, event_time_ms
from .cpu import cpu_baseline
from .locate import locate_leg
from .roofline import make_roofline, price_traffic, run_census, stored_traffic

def accel_legs(out, wl, args, rflat):
    """the config-2 patterns on the PLAIN index (FMX_FLAG_PLAIN: the reference's loop step for step, one rank per interval
    end and symbol -- `plain` / `value_plain`, the headline of rounds 1-5) and on each count accelerator alone (pair index,
    k-mer start table).  (s, e) asserted identical to the headline index's on all patterns -- config 2 and, when given,
    config 2b (early-exit pairs); each leg gets the roofline object of the headline (census of its own launch, counters
    of its own kernel from the live PMC passes)."""
    torch, F, lib = wl.torch, wl.F, wl.lib
    legs = []
    if wl.dna and wl.accelerated():
        legs.append(("plain", dict(plain=True), "FMX_FLAG_PLAIN: no accelerators", "fmx_count_f3_kernel<1,false,false>"))
    if wl.dna:
        legs.append(("pair_index", dict(plain=True, pair_index=True), "FMX_FLAG_PLAIN | FMX_FLAG_PAIR_INDEX",
                     "fmx_count_pair_kernel<false>"))
    legs.append(("kmer_table", dict(plain=True, kmer_table=True) if wl.dna else dict(kmer_table=True),
                 "FMX_FLAG_PLAIN | FMX_FLAG_KMER_TABLE" if wl.dna else "opt-in FMX_FLAG_KMER_TABLE",
                 "fmx_count_f3_kernel<1,false,true>" if wl.dna else "fmx_count_ep_kernel<..., true>"))
    npat, m = wl.npat, wl.m
    stream_bytes = npat * m + (npat + 1) * 8 + 2 * npat * 8
    for leg_name, leg_kw, leg_note, kname in legs:
        try:
            pidx = (F.RLFMIndex if wl.rlfm else F.FMIndex).from_device_text(wl.text.data_ptr(), wl.n, wl.maxc,
                                                                            device=wl.local, **leg_kw)
            if leg_kw.get("kmer_table") and pidx.kmer_k() == 0:
                out[leg_name] = {"skipped": "FMX_FLAG_KMER_TABLE is ignored for this kind / alphabet"}
                pidx.close()
                continue
            ps = torch.empty(npat, dtype=torch.int64, device=wl.dev)
            pe = torch.empty(npat, dtype=torch.int64, device=wl.dev)

            def pstep(p, use=lib):
                rc = use.fmx_count_batch_dev(pidx.handle(), C.c_void_p(p.data_ptr()), C.c_void_p(wl.off.data_ptr()),
                                             npat, None, C.c_void_p(ps.data_ptr()), C.c_void_p(pe.data_ptr()),
                                             None, wl.sp)
                assert rc == 0
            for _ in range(args.warmup):
                pstep(wl.pat)
            torch.cuda.synchronize()
            pms = event_time_ms(torch, wl.stream, lambda: pstep(wl.pat), args.steps)
            assert bool((ps == wl.d_s).all()) and bool((pe == wl.d_e).all()), leg_name + " != the headline index"
            cen = None
            if not args.no_census:
                cen = run_census(wl, lambda cl: pstep(wl.pat, cl), npat * m * 3 + (1 << 20))
            out[leg_name] = {"value": npat * m / (pms / 1e3), "unit": "pattern-chars/s", "ms_per_step": pms,
                             "index_bytes": pidx.heap_size(), "kmer_k": pidx.kmer_k(),
                             "pair_index": pidx.has_pair_index(),
                             "build_ms": round(float(lib.fmx_build_ms(pidx.handle())), 1),
                             "note": leg_note + "; (s,e) identical to the headline index's",
                             "roofline": make_roofline(kname, pms, npat * m, wl.ref_bytes_per_char(), stream_bytes, cen,
                                                       None)}
            if leg_name == "plain":
                out["value_plain"] = out[leg_name]["value"]
                out[leg_name]["roofline"]["table_bytes"] = (wl.n // 256 + 1) * 128
            if rflat is not None:
                # config 2b patterns (uniform random, mostly absent) through the same index
                wl.count(pat=rflat)
                for _ in range(args.warmup):
                    pstep(rflat)
                torch.cuda.synchronize()
                rms2 = event_time_ms(torch, wl.stream, lambda: pstep(rflat), args.steps)
                assert bool((ps == wl.d_s).all()) and bool((pe == wl.d_e).all()), leg_name + " != the headline index (2b)"
                out[leg_name]["early_exit_ms_per_step"] = rms2
                out[leg_name]["early_exit_offered_chars_per_s"] = npat * m / (rms2 / 1e3)
                wl.count()                            # restore the config-2 (s, e)
                torch.cuda.synchronize()
            pidx.close()
        except Exception as ex:  # noqa: BLE001 -- never lose the headline line to an optional leg
            out[leg_name] = {"error": repr(ex)}



def d2h_leg(out, wl, args):
    """value_incl_d2h: host patterns in, (s, e, count) out in pinned, reused host arrays."""
    torch, lib = wl.torch, wl.lib
    import numpy as np
    npat, m = wl.npat, wl.m
    hp = torch.empty(npat * m, dtype=torch.uint8, pin_memory=True)
    hp.copy_(wl.pat)
    ho = torch.empty(npat + 1, dtype=torch.int64, pin_memory=True)
    ho.copy_(wl.off)
    hs = torch.empty(npat, dtype=torch.int64, pin_memory=True)
    he = torch.empty(npat, dtype=torch.int64, pin_memory=True)
    hc = torch.empty(npat, dtype=torch.int64, pin_memory=True)
    torch.cuda.synchronize()

    def call():
        rc = lib.fmx_count_batch(wl.h, C.c_void_p(hp.data_ptr()), C.c_void_p(ho.data_ptr()), npat, None,
                                 C.c_void_p(hs.data_ptr()), C.c_void_p(he.data_ptr()), C.c_void_p(hc.data_ptr()))
        if rc != 0:
            raise RuntimeError(lib.fmx_last_error().decode())
    for _ in range(3):
        call()
    reps = max(5, args.steps // 4)
    t0 = time.perf_counter()
    for _ in range(reps):
        call()
    dt = (time.perf_counter() - t0) / reps
    assert bool((hs.to(wl.dev) == wl.d_s).all()) and bool((he.to(wl.dev) == wl.d_e).all())
    # the same call on ordinary (pageable) numpy arrays, also reused across calls
    pp, po = hp.numpy().copy(), ho.numpy().copy()
    ps, pe, pc = (np.zeros(npat, dtype=np.int64) for _ in range(3))

    def call_pageable():
        rc = lib.fmx_count_batch(wl.h, pp.ctypes.data_as(C.c_void_p), po.ctypes.data_as(C.c_void_p), npat, None,
                                 ps.ctypes.data_as(C.c_void_p), pe.ctypes.data_as(C.c_void_p),
                                 pc.ctypes.data_as(C.c_void_p))
        if rc != 0:
            raise RuntimeError(lib.fmx_last_error().decode())
    for _ in range(3):
        call_pageable()
    t0 = time.perf_counter()
    for _ in range(reps):
        call_pageable()
    dtp = (time.perf_counter() - t0) / reps
    assert (ps == hs.numpy()).all() and (pe == he.numpy()).all()
    out["value_incl_d2h"] = npat * m / min(dt, dtp)
    out["incl_d2h"] = {"pinned_ms_per_call": dt * 1e3, "pinned_value": npat * m / dt,
                       "pageable_ms_per_call": dtp * 1e3, "pageable_value": npat * m / dtp,
                       "value_is": "pinned" if dt <= dtp else "pageable",
                       "bytes_in": npat * m + (npat + 1) * 8, "bytes_out": 3 * npat * 8,
                       "note": "fmx_count_batch (host pointers): upload, count, download, synchronise per call; "
                               "caller-owned arrays reused across calls.  Page-locked arrays: 4-chunk pipeline over "
                               "three streams (DMA upload, search, download by copy kernels); pageable arrays: the "
                               "runtime's pin-copy-unpin copies in two chunks.  value_incl_d2h is the better of the "
                               "two; never the headline value"}
    del hp, ho, hs, he, hc


def wide_leg(out, args, dev):
    """`usize` rows (fm_index.rs:86-95): a DNA FMIndexWithLocate over n = 2^32 + 2^20 symbols on the wide engine
    (fmx_wide.hip) -- built here, 2^20 length-32 substring patterns counted, 2^20 hits located; every count >= 1 and every
    located position holds its pattern (checked on the device).  tests/test_gpu_beyond_4g.py is the parity test."""
    import torch
    import fm_index_amd as F
    from fm_index_amd import workload as W
    from fm_index_amd import _lib as L
    torch.cuda.empty_cache()
    free, _total = torch.cuda.mem_get_info()
    need = 200 << 30
    if free < need:
        out["wide"] = {"skipped": "needs ~190 GB of free HBM for the build, %.0f GB free" % (free / 2 ** 30)}
        return
    lib = L.lib()
    n, level, npat, m = (1 << 32) + (1 << 20), 2, 1 << 20, 32
    t0 = time.perf_counter()
    text = W.dna_text_torch(n, 17, dev)
    torch.cuda.synchronize()
    textgen_s = time.perf_counter() - t0
    t0 = time.perf_counter()
    index = F.FMIndexWithLocate.from_device_text(text.data_ptr(), n, 4, level=level)
    build_wall_s = time.perf_counter() - t0
    h = index.handle()
    assert index.is_wide() and index.len() == n
    stream = torch.cuda.current_stream()
    sp = C.c_void_p(stream.cuda_stream)

    def patterns(seed, mm):
        src = W.umod_torch(W.splitmix64_torch(seed, 0, npat, dev), n - 1 - mm)
        pat = text[src[:, None] + torch.arange(mm, dtype=torch.int64, device=dev)[None, :]].reshape(-1).contiguous()
        off = (torch.arange(npat + 1, dtype=torch.int64, device=dev) * mm).contiguous()
        return src, pat, off
    s, e, c = (torch.empty(npat, dtype=torch.int64, device=dev) for _ in range(3))

    def count(pat, off):
        rc = lib.fmx_count_batch_dev(h, C.c_void_p(pat.data_ptr()), C.c_void_p(off.data_ptr()), npat, None,
                                     C.c_void_p(s.data_ptr()), C.c_void_p(e.data_ptr()), C.c_void_p(c.data_ptr()), sp)
        assert rc == 0, lib.fmx_last_error().decode()
    _src, pat, off = patterns(3, m)
    for _ in range(args.warmup):
        count(pat, off)
    torch.cuda.synchronize()
    count_ms = event_time_ms(torch, stream, lambda: count(pat, off), args.steps)
    assert lib.fmx_stream_status(h) == 0 and bool((c >= 1).all())
    rows_beyond = int((e > (1 << 32)).sum().item())
    # locate: length-22 substrings (about one hit each at this n), rows expanded and walked
    src2, pat2, off2 = patterns(5, 22)
    count(pat2, off2)
    d_off = torch.empty(npat + 1, dtype=torch.int64, device=dev)
    assert lib.fmx_offsets_dev(h, C.c_void_p(s.data_ptr()), C.c_void_p(e.data_ptr()), npat, C.c_void_p(d_off.data_ptr()), sp) == 0
    total = int(d_off[-1].item())
    pos = torch.empty(total, dtype=torch.int64, device=dev)

    def locate():
        rc = lib.fmx_locate_batch_dev(h, C.c_void_p(s.data_ptr()), C.c_void_p(e.data_ptr()), npat,
                                      C.c_void_p(d_off.data_ptr()), total, C.c_void_p(pos.data_ptr()), sp)
        assert rc == 0, lib.fmx_last_error().decode()
    for _ in range(args.warmup):
        locate()
    torch.cuda.synchronize()
    locate_ms = event_time_ms(torch, stream, locate, args.steps)
    assert lib.fmx_stream_status(h) == 0
    hit = torch.repeat_interleave(torch.arange(npat, device=dev), c)
    ok = torch.ones(total, dtype=torch.bool, device=dev)
    for j in range(22):
        ok &= text[pos + j] == pat2.view(npat, 22)[hit, j]
    assert bool(ok.all()), "a located position does not hold its pattern"
    found = torch.zeros(npat, dtype=torch.bool, device=dev)
    found[hit[pos == src2[hit]]] = True
    assert bool(found.all()), "a pattern's source position is not among its hits"
    out["wide"] = {
        "workload": "FMIndexWithLocate, n=2^32+2^20 sigma=4 DNA text: %d x len-%d substring patterns counted, %d hits located "
                    "(level %d)" % (npat, m, total, level),
        "text_len": n, "engine": "64-bit rows (fmx_wide.hip)", "value": npat * m / (count_ms / 1e3),
        "unit": "pattern-chars/s", "ms_per_step": count_ms, "intervals_with_e_beyond_2^32": rows_beyond,
        "locate": {"hits": total, "ms_per_batch": locate_ms, "hits_per_s": total / (locate_ms / 1e3),
                   "positions_beyond_2^32": int((pos >= (1 << 32)).sum().item()),
                   "checked": "every located position holds its pattern; every source position is among the hits"},
        "index_bytes": index.heap_size(), "build_ms": round(float(lib.fmx_build_ms(h)), 1),
        "build_wall_s": round(build_wall_s, 2), "textgen_s": round(textgen_s, 2),
        "walk_records": index.walk_records(),
        "pretouch": dict(PRETOUCH) if PRETOUCH else None,
        "note": "round 5: the builder takes its temporaries' slab (32 B per symbol) and the index's own arrays BEFORE its first "
                "kernel touches memory, so that no hipMalloc follows a first touch (on memory nobody has used since boot "
                "this driver charges such a call ~28 ms per GiB touched: 4-5 s of this build in rounds 3-4, hidden by a "
                "pre-touch child process in round 4 -- now opt-in, `--pretouch`)"}
    index.close()
    del text, pat, pat2, pos
    torch.cuda.empty_cache()


def rlfm_leg(out, args, dev, local):
    """config 4: RLFMIndex (src/rlfmi.rs) over the 1 GB sigma=255 byte text, 2^20 length-16 patterns."""
    import torch
    wr = Workload("bytes-rlfm", args, dev, local, 0, 1)
    npat, m = wr.npat, wr.m
    for _ in range(args.warmup):
        wr.count()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ms = event_time_ms(torch, wr.stream, wr.count, args.steps)
    dt = (time.perf_counter() - t0) / args.steps
    kms, steps_exec = wr.timed_kernel(wr.count)
    assert steps_exec == npat * m, (steps_exec, npat * m)
    assert bool((wr.d_c >= 1).all())
    assert wr.lib.fmx_stream_status(wr.h) == 0
    cen = None
    if not args.no_census:
        cen = run_census(wr, lambda cl: wr.count(lib=cl), npat * m * 10 + (1 << 20))
    key = "bytes-rlfm:%d:%d:%d" % (npat, m, args.log2n)
    stream_bytes = npat * m + (npat + 1) * 8 + 3 * npat * 8
    o = {"value": npat * m / (ms / 1e3), "unit": "pattern-chars/s", "ms_per_step": ms,
         "wall_ms_per_step": dt * 1e3,
         "config": {"workload": wr.describe(1), "text_len": wr.n, "patterns": npat, "pattern_len": m,
                    "index_bytes": wr.index.heap_size(), "runs": int(wr.lib.fmx_num_runs(wr.h)),
                    "build_ms": round(wr.build_ms, 1), "textgen_s": round(wr.textgen_s, 2)},
         "roofline": make_roofline("fmx_count_ep_kernel", ms, npat * m, wr.ref_bytes_per_char(), stream_bytes,
                                   cen, stored_traffic(key, "count"))}
    out["rlfm"] = o
    # the count-only type of BASELINE config 4 (`RLFMIndex`, frontend.rs:213-231): what the index costs without samples
    try:
        co = wr.F.RLFMIndex.from_device_text(wr.text.data_ptr(), wr.n, wr.maxc, device=local)
        o["config"]["count_only_index_bytes"] = co.heap_size()
        co.close()
    except Exception as ex:  # noqa: BLE001
        o["config"]["count_only_index_bytes"] = repr(ex)
    # the default RLFM index of 2^24+ symbols carries the k-mer start table (round 6); the plain one next to it
    o["config"]["kmer_k"] = wr.index.kmer_k()
    if wr.index.kmer_k() and not args.no_accel:
        try:
            pl = wr.F.RLFMIndex.from_device_text(wr.text.data_ptr(), wr.n, wr.maxc, device=local, plain=True)
            ps = torch.empty(npat, dtype=torch.int64, device=wr.dev)
            pe = torch.empty(npat, dtype=torch.int64, device=wr.dev)

            def pstep():
                rc = wr.lib.fmx_count_batch_dev(pl.handle(), C.c_void_p(wr.pat.data_ptr()), C.c_void_p(wr.off.data_ptr()), npat,
                                                None, C.c_void_p(ps.data_ptr()), C.c_void_p(pe.data_ptr()), None, wr.sp)
                assert rc == 0
            for _ in range(args.warmup):
                pstep()
            torch.cuda.synchronize()
            pms = event_time_ms(torch, wr.stream, pstep, args.steps)
            assert bool((ps == wr.d_s).all()) and bool((pe == wr.d_e).all()), "plain RLFM index != default RLFM index"
            o["plain"] = {"value": npat * m / (pms / 1e3), "ms_per_step": pms, "index_bytes": pl.heap_size(),
                          "note": "FMX_FLAG_PLAIN: no k-mer start table; (s, e) identical on all patterns"}
            pl.close()
        except Exception as ex:  # noqa: BLE001
            o["plain"] = {"error": repr(ex)}
    o["config"]["run_table"] = bool(wr.index.walk_records())
    o["config"]["space_policy"] = ("run table (4 B per run) only when r <= n/4 or FMX_FLAG_RUN_TABLE: this text has %.2f runs "
                                   "per row" % (o["config"]["runs"] / float(wr.n)))
    if wr.level is not None:
        locate_leg(out, wr, args, 1, 0, None, False, key, dest=o)
        if not wr.index.walk_records() and not args.no_accel:
            # the same text with FMX_FLAG_RUN_TABLE (opt-in on a text with about one run per row): the round-4 locate path
            try:
                w2 = Workload("bytes-rlfm", args, dev, local, 0, 1, index_kw={"run_table": True}, text=wr.text)
                try:
                    locate_leg(out, w2, args, 1, 0, None, False, key, dest=o, legname="locate_run_table")
                    o["locate_run_table"]["index_bytes"] = w2.index.heap_size()
                    assert bool((w2.d_pos[:w2.total_hits] == wr.d_pos[:wr.total_hits]).all()), "run table locates differently"
                finally:
                    w2.close()
            except Exception as ex:  # noqa: BLE001
                o["locate_run_table"] = {"error": repr(ex)}
    if not args.no_cpu_baseline:
        wr.count()
        torch.cuda.synchronize()
        o["cpu_baseline"] = cpu_baseline(wr, args, "rlfm")
    return wr



def ic_ab_leg(out, args, dev, local, pmc31):
    """Infinity-Cache A/B of the headline kernel: the SAME kernel, batch shape and pattern length on
    an index FOUR times the 256 MiB Infinity Cache (n = 2^31: 1 GiB of count records) next to the headline's n = 2^30
    (512 MiB of records, of which up to half can sit in that cache).  The fabric counters cannot tell a cache hit from an
    HBM read; this leg bounds the cache's part from outside: at n = 2^31 at most a quarter of the fetched bytes can be
    cache hits (`hbm_frac_min`), and the ratio of the two rates per fabric request is what the cache buys the headline."""
    import argparse
    import torch
    a31 = argparse.Namespace(**vars(args))
    a31.log2n = args.log2n + 1
    w = Workload("dna", a31, dev, local, 0, 1, with_locate=False)
    try:
        npat, m = w.npat, w.m
        for _ in range(args.warmup):
            w.count()
        torch.cuda.synchronize()
        ms = event_time_ms(torch, w.stream, w.count, args.steps)
        _kms, steps_exec = w.timed_kernel(w.count)
        assert steps_exec == npat * m and bool((w.d_c >= 1).all()) and w.lib.fmx_stream_status(w.h) == 0
        cen = None
        if not args.no_census:
            cen = run_census(w, lambda cl: w.count(lib=cl), npat * m * 3 + (1 << 20))
        stream_bytes = npat * m + (npat + 1) * 8 + 3 * npat * 8
        roof = make_roofline("fmx_count_f3_kernel<1,false,false>", ms, npat * m, w.ref_bytes_per_char(), stream_bytes, cen,
                             None, table_bytes=w.count_table_bytes())
        if pmc31 and isinstance(pmc31[0], dict):
            price_traffic(roof, pmc31[0].get("dna_count"))
        o = {"workload": "the headline kernel on n=2^%d (count records %d MiB = %.1f x the Infinity Cache), %d x len-%d "
                         "substring patterns" % (a31.log2n, w.count_table_bytes() >> 20,
                                                 w.count_table_bytes() / float(256 << 20), npat, m),
             "value": npat * m / (ms / 1e3), "unit": "pattern-chars/s", "ms_per_step": ms,
             "build_ms": round(w.build_ms, 1), "roofline": roof}
        h = (out.get("plain") or {}).get("roofline") or out.get("roofline") or {}     # the same kernel at n = 2^30
        if h.get("avg_kernel_ms"):
            o["ms_vs_headline"] = round(ms / h["avg_kernel_ms"], 4)
        if h.get("fabric_requests") and roof.get("fabric_requests"):
            o["fabric_requests_vs_headline"] = round(roof["fabric_requests"] / h["fabric_requests"], 4)
        out["count_n31"] = o
    finally:
        w.close()
        del w
        torch.cuda.empty_cache()
