"""BASELINE config 5 through the C ABI from ONE host caller (include/fmx.h: fmx_replicate, fmx_count_batch_multi,
fmx_count_batch_multi_resident), and the SURVEY 8(d) protocol number: patterns resident in HBM, results landed in host
memory ("exclude index build and H2D upload from both sides; include D2H of results on the GPU side";
benches/count.rs:29-37 times results the caller can read)."""
import ctypes as C
import time

from .common import counts_sha256, golden_counts_sha, ranges_sha256
from .dist import CONFIG5_PATTERNS, CONFIG5_SEED


def _time_calls(fn, reps, warm=2):
    for _ in range(warm):
        fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    return (time.perf_counter() - t0) / reps


def results_on_host_leg(out, wl, args):
    """config 2's batch (2^20 x len-32, resident in HBM) through fmx_count_batch_multi_resident on ONE handle: every call
    ends with (s, e, count) -- or the counts alone -- readable in page-locked host arrays.  `value_results_on_host` is the
    (s, e, count) figure; never the headline `value` (whose results stay in HBM)."""
    torch, lib = wl.torch, wl.lib
    npat, m = wl.npat, wl.m
    hs, he, hc = (torch.zeros(npat, dtype=torch.int64).pin_memory() for _ in range(3))
    handles = (C.c_void_p * 1)(wl.h.value)
    d_pat = (C.c_void_p * 1)(wl.pat.data_ptr())
    d_off = (C.c_void_p * 1)(wl.off.data_ptr())
    torch.cuda.synchronize()

    def call(full):
        rc = lib.fmx_count_batch_multi_resident(handles, 1, d_pat, d_off, npat, None,
                                                C.c_void_p(hs.data_ptr()) if full else None,
                                                C.c_void_p(he.data_ptr()) if full else None, C.c_void_p(hc.data_ptr()))
        if rc != 0:
            raise RuntimeError(lib.fmx_last_error().decode())
    reps = max(10, args.steps // 2)
    dt_full = _time_calls(lambda: call(True), reps)
    assert bool((hs.to(wl.dev) == wl.d_s).all()) and bool((he.to(wl.dev) == wl.d_e).all()) and \
        bool((hc.to(wl.dev) == wl.d_c).all()), "results_on_host: (s, e, count) differ from the device-resident results"
    hc.zero_()
    dt_cnt = _time_calls(lambda: call(False), reps)
    assert bool((hc.to(wl.dev) == wl.d_c).all())
    # pageable result arrays: device scratch + the runtime's copies
    import numpy as np
    ps, pe, pc = (np.zeros(npat, dtype=np.int64) for _ in range(3))

    def call_pageable():
        rc = lib.fmx_count_batch_multi_resident(handles, 1, d_pat, d_off, npat, None, ps.ctypes.data_as(C.c_void_p),
                                                pe.ctypes.data_as(C.c_void_p), pc.ctypes.data_as(C.c_void_p))
        if rc != 0:
            raise RuntimeError(lib.fmx_last_error().decode())
    dt_pg = _time_calls(call_pageable, max(5, reps // 2))
    assert (pc == hc.numpy()).all() and (ps == hs.numpy()).all()
    out["value_results_on_host"] = npat * m / dt_full
    out["results_on_host"] = {
        "protocol": "SURVEY 8(d): patterns resident in HBM, one synchronous call per batch, results readable in host "
                    "memory when it returns (fmx_count_batch_multi_resident, 1 handle); page-locked result arrays are "
                    "written by the search kernel itself over the host link",
        "s_e_count_ms_per_call": dt_full * 1e3, "s_e_count_value": npat * m / dt_full, "bytes_out_s_e_count": 24 * npat,
        "count_only_ms_per_call": dt_cnt * 1e3, "count_only_value": npat * m / dt_cnt, "bytes_out_count_only": 8 * npat,
        "pageable_s_e_count_ms_per_call": dt_pg * 1e3, "pageable_s_e_count_value": npat * m / dt_pg,
        "vs_value": round(npat * m / dt_full / out["value"], 4)}
    del hs, he, hc


def config5_cabi_leg(out, wl, args, dev):
    """config 5's 8 388 608-pattern set through fmx_count_batch_multi from one host thread: page-locked host patterns in,
    (s, e, count) in place in page-locked host arrays.  G = every device this process sees (the one-GPU box: 1), then --
    the sharding itself, on whatever hardware -- three replicas made by fmx_replicate on device 0 (ragged shards).  Every
    variant must hash to tests/golden/config5_counts.json (the CPU oracle over all patterns)."""
    import torch
    import fm_index_amd as F
    from fm_index_amd import workload as W
    lib, n, m = wl.lib, wl.n, wl.m
    T = CONFIG5_PATTERNS if args.log2n >= 30 else max(args.npat * 8, 1 << 15)
    hp = torch.empty(T * m, dtype=torch.uint8).pin_memory()
    ar = torch.arange(m, dtype=torch.int64, device=dev)[None, :]
    chunk = 1 << 20
    for lo in range(0, T, chunk):
        k = min(chunk, T - lo)
        src = W.umod_torch(W.splitmix64_torch(CONFIG5_SEED, lo, k, dev), n - 1 - m)
        hp[lo * m:(lo + k) * m].copy_(wl.text[src[:, None] + ar].reshape(-1))
    del src
    ho = (torch.arange(T + 1, dtype=torch.int64) * m).pin_memory()
    hs, he, hc = (torch.zeros(T, dtype=torch.int64).pin_memory() for _ in range(3))
    torch.cuda.synchronize()
    gold = golden_counts_sha(wl, args, total=T, seed=CONFIG5_SEED)
    o = {"workload": "config 5 through the C ABI: %d x len-%d substring patterns (seed %d), page-locked host arrays in and "
                     "out, one fmx_count_batch_multi call per step from one host thread" % (T, m, CONFIG5_SEED),
         "total_patterns": T, "unit": "pattern-chars/s"}

    def run(handles, g, reps):
        def call():
            rc = lib.fmx_count_batch_multi(handles, g, C.c_void_p(hp.data_ptr()), C.c_void_p(ho.data_ptr()), T, None,
                                           C.c_void_p(hs.data_ptr()), C.c_void_p(he.data_ptr()), C.c_void_p(hc.data_ptr()))
            if rc != 0:
                raise RuntimeError(lib.fmx_last_error().decode())
        for a in (hs, he, hc):
            a.zero_()
        dt = _time_calls(call, reps, warm=2)
        sha, rsha = counts_sha256(hc.numpy()), ranges_sha256(hs.numpy(), he.numpy())
        r = {"replicas": g, "ms_per_step": dt * 1e3, "value": T * m / dt, "counts_sha256": sha, "ranges_sha256": rsha}
        if gold is not None:
            r["matches_golden"] = gold[0] == sha and gold[1] in (None, rsha)
            assert r["matches_golden"], "config5 through the C ABI differs from tests/golden/config5_counts.json"
        return r

    reps = max(4, args.steps // 5)
    ndev = torch.cuda.device_count()
    others = []
    try:
        # G = the devices of this process: replica r on device r (device `dev.index` holds the workload's own handle)
        devs = [d for d in range(ndev) if d != dev.index]
        for d in devs:
            h = C.c_void_p()
            if lib.fmx_replicate(wl.h, d, C.byref(h)) != 0:
                raise RuntimeError(lib.fmx_last_error().decode())
            others.append(h)
        g = 1 + len(others)
        handles = (C.c_void_p * g)(wl.h.value, *[h.value for h in others])
        o["g%d" % g] = run(handles, g, reps)
        o["g%d" % g]["devices"] = [dev.index] + devs
        if g > 1:                                          # the one-device point of the same curve
            o["g1"] = run((C.c_void_p * 1)(wl.h.value), 1, reps)
            o["g1"]["devices"] = [dev.index]
    finally:
        for h in others:
            lib.fmx_free(h)
    # the sharding on ONE device: three replicas of the index on this GPU, ragged shards (T / 3)
    reps3 = []
    try:
        for _ in range(2):
            h = C.c_void_p()
            t0 = time.perf_counter()
            if lib.fmx_replicate(wl.h, dev.index, C.byref(h)) != 0:
                raise RuntimeError(lib.fmx_last_error().decode())
            torch.cuda.synchronize()
            o.setdefault("replicate_ms", []).append(round((time.perf_counter() - t0) * 1e3, 1))
            reps3.append(h)
        handles = (C.c_void_p * 3)(wl.h.value, reps3[0].value, reps3[1].value)
        o["three_replicas_one_device"] = run(handles, 3, max(3, reps // 2))
    finally:
        for h in reps3:
            lib.fmx_free(h)
    o["replica_bytes"] = wl.index.heap_size()
    o["value"] = o["g1"]["value"]
    o["matches_golden"] = all(v.get("matches_golden", True) for v in o.values() if isinstance(v, dict))
    out["config5_cabi"] = o
    del hp, ho, hs, he, hc
