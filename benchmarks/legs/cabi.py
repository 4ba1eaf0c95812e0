"""BASELINE config 5 through the C ABI from ONE host caller (include/fmx.h: fmx_replicate, fmx_count_batch_multi,
fmx_count_batch_multi_resident), and the SURVEY 8(d) protocol number: patterns resident in HBM, results landed in host
memory ("exclude index build and H2D upload from both sides; include D2H of results on the GPU side";
benches/count.rs:29-37 times results the caller can read)."""
import ctypes as C
import json
import os
import subprocess
import sys
import time

from .common import counts_sha256, golden_counts_sha, ranges_sha256
from .dist import CONFIG5_PATTERNS, CONFIG5_SEED


def _time_calls(fn, reps, warm=2, stats=None):
    """seconds per synchronous call: the MEDIAN of `reps` individually timed calls (a host-side hiccup -- one 49 ms stall
    among ten 0.3 ms calls was seen on one box -- must not become the figure); `stats` (a dict) receives mean and max"""
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    ts.sort()
    if stats is not None:
        stats.update({"calls": reps, "mean_ms": sum(ts) / reps * 1e3, "max_ms": ts[-1] * 1e3, "min_ms": ts[0] * 1e3})
    return ts[reps // 2]


def results_on_host_leg(out, wl, args):
    """config 2's batch (2^20 x len-32, resident in HBM) through fmx_count_batch_multi_resident on ONE handle: every call
    ends with the counts -- or (s, e, count) -- readable in page-locked host arrays.  `value_results_on_host` is the
    counts figure (`search(p).count()`), `value_ranges_on_host` the (s, e, count) one; never the headline `value` (whose
    results stay in HBM)."""
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
    reps = max(20, args.steps // 2)
    st_full, st_cnt = {}, {}
    dt_full = _time_calls(lambda: call(True), reps, stats=st_full)
    assert bool((hs.to(wl.dev) == wl.d_s).all()) and bool((he.to(wl.dev) == wl.d_e).all()) and \
        bool((hc.to(wl.dev) == wl.d_c).all()), "results_on_host: (s, e, count) differ from the device-resident results"
    hc.zero_()
    dt_cnt = _time_calls(lambda: call(False), reps, stats=st_cnt)
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
    # BASELINE's count metric is `search(p).count()` (benches/count.rs:29-37): the COUNTS landed in host memory are the
    # protocol number; (s, e, count) -- 24 bytes per pattern over a ~56 GB/s host link -- is reported next to it
    out["value_results_on_host"] = npat * m / dt_cnt
    out["value_ranges_on_host"] = npat * m / dt_full
    out["results_on_host"] = {
        "protocol": "SURVEY 8(d): patterns resident in HBM, one synchronous call per batch, results readable in host "
                    "memory when it returns (fmx_count_batch_multi_resident, 1 handle); page-locked result arrays are "
                    "written by the search kernel itself over the host link",
        "s_e_count_ms_per_call": dt_full * 1e3, "s_e_count_value": npat * m / dt_full, "bytes_out_s_e_count": 24 * npat,
        "count_only_ms_per_call": dt_cnt * 1e3, "count_only_value": npat * m / dt_cnt, "bytes_out_count_only": 8 * npat,
        "pageable_s_e_count_ms_per_call": dt_pg * 1e3, "pageable_s_e_count_value": npat * m / dt_pg,
        "timing": "median of individually timed synchronous calls", "s_e_count_calls": st_full, "count_only_calls": st_cnt,
        "value_results_on_host_is": "count_only", "vs_value": round(npat * m / dt_cnt / out["value"], 4),
        "s_e_count_vs_value": round(npat * m / dt_full / out["value"], 4)}
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

    def run_resident(handles, g, devices, reps):
        """the same set with every shard's patterns RESIDENT on its replica's device (uploads excluded), results
        into the page-locked host arrays: fmx_count_batch_multi_resident"""
        keep, d_pat, d_off = [], (C.c_void_p * g)(), (C.c_void_p * g)()
        for r in range(g):
            a = (T * r + g - 1) // g
            b = (T * (r + 1) + g - 1) // g
            dv = torch.device("cuda", devices[r])
            fp = hp[a * m:b * m].to(dv)
            fo = (torch.arange(b - a + 1, dtype=torch.int64, device=dv) * m).contiguous()
            keep += [fp, fo]
            d_pat[r], d_off[r] = fp.data_ptr(), fo.data_ptr()
        for d in set(devices):
            torch.cuda.synchronize(d)

        def call():                               # config 5 gathers the COUNTS: 8 bytes per pattern
            rc = lib.fmx_count_batch_multi_resident(handles, g, d_pat, d_off, T, None, None, None, C.c_void_p(hc.data_ptr()))
            if rc != 0:
                raise RuntimeError(lib.fmx_last_error().decode())
        hc.zero_()
        dt = _time_calls(call, reps, warm=2)
        sha = counts_sha256(hc.numpy())
        r = {"replicas": g, "ms_per_step": dt * 1e3, "value": T * m / dt, "counts_sha256": sha,
             "patterns": "resident on the replicas' devices", "results": "counts in a page-locked host array"}
        if gold is not None:
            r["matches_golden"] = gold[0] == sha
            assert r["matches_golden"], "config5 (resident) through the C ABI differs from tests/golden/config5_counts.json"
        del keep
        return r

    reps = max(4, args.steps // 5)
    # G = 1 in this process
    o["g1"] = run((C.c_void_p * 1)(wl.h.value), 1, reps)
    o["g1"]["devices"] = [dev.index]
    o["g1_resident"] = run_resident((C.c_void_p * 1)(wl.h.value), 1, [dev.index], reps)
    # G = 2, 4, 8 ... replicas on DISTINCT devices, when the process sees more than one: in a CHILD process with a
    # timeout (`bench.py --cabi-child`) -- no box this code was developed on has a second GPU, and neither a peer copy
    # that hangs nor a worker thread that faults may take the headline line with it
    ndev = torch.cuda.device_count()
    if ndev > 1 or os.environ.get("FMX_BENCH_CABI_CHILD"):
        o["multi_device"] = run_cabi_child(args, max(ndev, 1))
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
        o["three_replicas_one_device_resident"] = run_resident(handles, 3, [dev.index] * 3, max(3, reps // 2))
    finally:
        for h in reps3:
            lib.fmx_free(h)
    o["replica_bytes"] = wl.index.heap_size()
    o["value"] = o["g1"]["value"]
    o["matches_golden"] = all(v.get("matches_golden", True) for v in o.values() if isinstance(v, dict))
    out["config5_cabi"] = o
    del hp, ho, hs, he, hc


def single_call_leg(out, wl, args):
    """One-at-a-time callers: the latency of ONE pattern through the host-pointer
    entry point the trait shim's `search_range` uses, the per-call time of small batches, and the batch size from which a
    call beats the one-thread CPU port of the reference (`benches/count.rs:20-52` times one `search(p).count()` at a
    time) -- on the bench's 1 GiB index (the CPU walks DRAM there) and on a 50 000-symbol text that sits in the CPU's
    caches (the reference's own benchmark size).  Every query still runs on the GPU: no CPU fallback."""
    import numpy as np
    import fm_index_amd as F
    from fm_index_amd import workload as W
    from oracle import fm_oracle as O
    lib = wl.lib
    m = wl.m
    sizes = (1, 4, 16, 64, 256, 1024, 4096, 16384)

    def gpu_sweep(handle, flat, off):
        res = {}
        s, e, c = (np.zeros(max(sizes), np.uint64) for _ in range(3))
        for b in sizes:
            o = np.ascontiguousarray(off[:b + 1])
            f = np.ascontiguousarray(flat[:int(o[-1])])

            def call():
                rc = lib.fmx_count_batch(handle, f.ctypes.data_as(C.c_void_p), o.ctypes.data_as(C.c_void_p), b, None,
                                         s.ctypes.data_as(C.c_void_p), e.ctypes.data_as(C.c_void_p),
                                         c.ctypes.data_as(C.c_void_p))
                if rc != 0:
                    raise RuntimeError(lib.fmx_last_error().decode())
            res[b] = _time_calls(call, 200 if b <= 1024 else 50, warm=5) * 1e6
        return res

    def break_even(gpu_us, cpu_us_per_pattern):
        for b in sizes:
            if gpu_us[b] <= b * cpu_us_per_pattern:
                return b
        return None

    o = {"what": "fmx_count_batch (host pointers, pageable arrays, synchronous) per call, microseconds, by batch size; "
                 "break_even_batch = smallest measured batch whose call takes no longer than the one-thread CPU port needs "
                 "for the same patterns"}
    # (a) the bench's index: n = 2^30
    flat = wl.pat[:max(sizes) * m].cpu().numpy()
    off = np.arange(max(sizes) + 1, dtype=np.uint64) * np.uint64(m)
    big = gpu_sweep(wl.h, flat, off)
    cb = out.get("cpu_baseline") or {}
    cpu_big = m / cb["single_thread_value"] * 1e6 if cb.get("single_thread_value") else None
    o["n_2^%d" % args.log2n] = {"gpu_us_per_call": {str(k): round(v, 1) for k, v in big.items()},
                                "one_pattern_us": round(big[1], 1),
                                "cpu_1_thread_us_per_pattern": round(cpu_big, 2) if cpu_big else None,
                                "break_even_batch": break_even(big, cpu_big) if cpu_big else None}
    # (b) the reference's benchmark size: 50 000 symbols, cache-resident on the CPU
    n2 = 50000
    t2 = W.dna_text_np(n2, 11)
    g2 = F.FMIndex(F.Text.with_max_character(t2, 4), device=wl.local)
    o2 = O.OracleIndex(t2, 4)
    f2, off2, _ = W.substring_patterns_np(t2, max(sizes), m, 12)
    small = gpu_sweep(g2.handle(), f2, off2)
    o2.count_batch(f2, off2, nthreads=1)
    t0 = time.perf_counter()
    for _ in range(5):
        so, eo = o2.count_batch(f2, off2, nthreads=1)
    cpu_small = (time.perf_counter() - t0) / 5 / max(sizes) * 1e6
    b = g2.search_many(flat=f2, off=off2)
    assert (b.s == so).all() and (b.e == eo).all()
    o["n_50000"] = {"gpu_us_per_call": {str(k): round(v, 1) for k, v in small.items()}, "one_pattern_us": round(small[1], 1),
                    "cpu_1_thread_us_per_pattern": round(cpu_small, 2), "break_even_batch": break_even(small, cpu_small)}
    g2.close()
    o2.close()
    # the one-element trait calls of the shim (one kernel launch each)
    row = int(wl.d_s[0].item())
    for name, fn in (("fmx_lf_map2", lambda: lib.fmx_lf_map2(wl.h, 2, row)), ("fmx_get_sa", lambda: lib.fmx_get_sa(wl.h, row))):
        if name == "fmx_get_sa" and wl.level is None:
            continue
        o[name + "_us"] = round(_time_calls(fn, 200, warm=5) * 1e6, 1)
    out["single_call"] = o


def run_cabi_child(args, ndev, timeout=240):
    """`bench.py --cabi-child` in its own session: returns its JSON object, or {"error": ...}"""
    root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--cabi-child", "--log2n", str(args.log2n), "--plen", str(args.plen),
           "--steps", str(max(4, args.steps // 5))]
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    try:
        p = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, start_new_session=True)
        try:
            outb, errb = p.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            try:
                os.killpg(p.pid, 9)
            except OSError:
                pass
            p.communicate()
            return {"error": "bench.py --cabi-child timed out after %d s (process group killed)" % timeout, "devices": ndev}
        lines = [ln for ln in outb.decode(errors="replace").splitlines() if ln.startswith("{")]
        if p.returncode != 0 or not lines:
            return {"error": "bench.py --cabi-child failed (rc %s): %s" % (p.returncode, errb.decode(errors="replace")[-400:]),
                    "devices": ndev}
        return json.loads(lines[-1])
    except OSError as ex:
        return {"error": repr(ex), "devices": ndev}


def cabi_child(args):
    """The program behind `bench.py --cabi-child`: config 5 through the C ABI over the devices of ONE process.  Builds the
    default DNA index on device 0, replicates it onto every other visible device (fmx_replicate: device-to-device
    copies), and runs the 8 388 608-pattern set through fmx_count_batch_multi (page-locked host patterns in, (s, e, count)
    in place) and fmx_count_batch_multi_resident (patterns resident per device, counts to the host) at G = 1, 2, 4, 8, ...
    replicas on G distinct devices.  Every point's hashes are compared with tests/golden/config5_counts.json.  Prints ONE
    JSON object."""
    import argparse
    import torch
    import fm_index_amd as F
    from fm_index_amd import _lib as L
    from fm_index_amd import workload as W
    from .common import golden_key
    lib = L.lib()
    ndev = torch.cuda.device_count()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    n, m = 1 << args.log2n, args.plen
    T = CONFIG5_PATTERNS if args.log2n >= 30 else 1 << 17
    t0 = time.perf_counter()
    text = W.dna_text_torch(n, 1, dev)
    index = F.FMIndex.from_device_text(text.data_ptr(), n, 4, device=0)
    out = {"devices_visible": ndev, "text_len": n, "total_patterns": T, "unit": "pattern-chars/s",
           "index_bytes": index.heap_size(), "accelerated": bool(index.has_pair_index() and index.kmer_k()),
           "build_s": round(time.perf_counter() - t0, 2)}
    hp = torch.empty(T * m, dtype=torch.uint8).pin_memory()
    ar = torch.arange(m, dtype=torch.int64, device=dev)[None, :]
    for lo in range(0, T, 1 << 20):
        k = min(1 << 20, T - lo)
        src = W.umod_torch(W.splitmix64_torch(CONFIG5_SEED, lo, k, dev), n - 1 - m)
        hp[lo * m:(lo + k) * m].copy_(text[src[:, None] + ar].reshape(-1))
    ho = (torch.arange(T + 1, dtype=torch.int64) * m).pin_memory()
    hs, he, hc = (torch.zeros(T, dtype=torch.int64).pin_memory() for _ in range(3))
    gold = None
    try:
        with open(os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests", "golden",
                               "config5_counts.json")) as f:
            ent = json.load(f)["entries"].get(golden_key("dna", args.log2n, CONFIG5_SEED, T, m))
        gold = (ent["counts_sha256"], ent.get("ranges_sha256")) if ent else None
    except (OSError, ValueError):
        pass
    replicas = [index]
    t0 = time.perf_counter()
    for d in range(1, ndev):
        replicas.append(index.replicate(d))
    for d in range(ndev):
        torch.cuda.synchronize(d)
    out["replicate_s"] = round(time.perf_counter() - t0, 3)
    reps = max(3, args.steps)
    points = {}
    gs = sorted({g for g in (1, 2, 4, 8, 16, ndev) if 1 <= g <= ndev})
    for g in gs:
        handles = (C.c_void_p * g)(*[replicas[r].handle().value for r in range(g)])

        def call_host():
            rc = lib.fmx_count_batch_multi(handles, g, C.c_void_p(hp.data_ptr()), C.c_void_p(ho.data_ptr()), T, None,
                                           C.c_void_p(hs.data_ptr()), C.c_void_p(he.data_ptr()), C.c_void_p(hc.data_ptr()))
            if rc != 0:
                raise RuntimeError(lib.fmx_last_error().decode())
        for a_ in (hs, he, hc):
            a_.zero_()
        dt = _time_calls(call_host, reps, warm=2)
        sha, rsha = counts_sha256(hc.numpy()), ranges_sha256(hs.numpy(), he.numpy())
        pt = {"host_patterns": {"ms_per_step": dt * 1e3, "value": T * m / dt, "counts_sha256": sha[:16],
                                "matches_golden": (gold[0] == sha and gold[1] in (None, rsha)) if gold else None}}
        keep, d_pat, d_off = [], (C.c_void_p * g)(), (C.c_void_p * g)()
        for r in range(g):
            a = (T * r + g - 1) // g
            b = (T * (r + 1) + g - 1) // g
            dv = torch.device("cuda", r)
            fp = hp[a * m:b * m].to(dv)
            fo = (torch.arange(b - a + 1, dtype=torch.int64, device=dv) * m).contiguous()
            keep += [fp, fo]
            d_pat[r], d_off[r] = fp.data_ptr(), fo.data_ptr()
        for r in range(g):
            torch.cuda.synchronize(r)

        def call_resident():
            rc = lib.fmx_count_batch_multi_resident(handles, g, d_pat, d_off, T, None, None, None, C.c_void_p(hc.data_ptr()))
            if rc != 0:
                raise RuntimeError(lib.fmx_last_error().decode())
        hc.zero_()
        dt = _time_calls(call_resident, reps, warm=2)
        sha = counts_sha256(hc.numpy())
        pt["resident_patterns"] = {"ms_per_step": dt * 1e3, "value": T * m / dt, "counts_sha256": sha[:16],
                                   "matches_golden": (gold[0] == sha) if gold else None}
        del keep
        points["g%d" % g] = pt
    out["points"] = points
    out["matches_golden"] = all(v[k]["matches_golden"] in (True, None) for v in points.values() for k in v)
    for r in replicas[1:]:
        r.close()
    index.close()
    print(json.dumps(out))
