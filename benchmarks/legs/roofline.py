"""The roofline objects of bench.py: line census (libfmx_census.so), live rocprofv3 --pmc passes over a child of
bench.py, and the pricing of the counters."""
import csv
import ctypes as C
import glob
import json
import os
import shutil
import subprocess
import sys
import tempfile

from .common import GATHER_CEILING_GLINES, HBM_PEAK_GBS, LINE, ROOT, Workload

IC_BYTES = 256 << 20         # MI355X_MICROARCH.md: Infinity Cache (L3), die-level, memory side of the fabric

# --------------------------------------------------------------------------------------------
# line census (libfmx_census.so: same kernels, every index-line load logged) -- outside timing
# --------------------------------------------------------------------------------------------
_CENSUS = {}


def census_lib():
    if "lib" not in _CENSUS:
        from fm_index_amd import _lib as L
        path = os.path.join(os.path.dirname(L.LIB_PATH), "libfmx_census.so")
        lib = None
        if os.path.exists(path):
            try:
                lib = C.CDLL(path)
                for name, res, argt in L.SYMBOLS:
                    fn = getattr(lib, name)
                    fn.restype, fn.argtypes = res, argt
                lib.fmx_census_begin.restype = C.c_int
                lib.fmx_census_begin.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p]
                lib.fmx_census_end.restype = C.c_int
            except (OSError, AttributeError):
                lib = None
        _CENSUS["lib"] = lib
    return _CENSUS["lib"]


def run_census(wl, launch, cap_entries):
    """lines REQUESTED by one launch and the DISTINCT lines among them (128-byte granules)."""
    lib = census_lib()
    if lib is None:
        return None
    torch = wl.torch
    log = torch.empty(cap_entries, dtype=torch.int64, device=wl.dev)
    cnt = torch.zeros(1, dtype=torch.int64, device=wl.dev)
    torch.cuda.synchronize()
    if lib.fmx_census_begin(C.c_void_p(log.data_ptr()), cap_entries, C.c_void_p(cnt.data_ptr())) != 0:
        return None
    try:
        launch(lib)
        torch.cuda.synchronize()
    except (RuntimeError, AssertionError) as ex:
        # e.g. a census library built from other sources than libfmx.so refuses the handle (FMX_LAYOUT): no census,
        # the line goes on without the request counts
        lib.fmx_census_end()
        return {"requested_lines": None, "distinct_lines": None, "note": "census launch failed: %r" % (ex,)}
    finally:
        lib.fmx_census_end()
    requested = int(cnt.item())
    if requested > cap_entries:
        return {"requested_lines": requested, "distinct_lines": None, "note": "log capacity exceeded"}
    ent = log[:requested]
    narrow = int((ent < 0).sum().item())          # bit 63: a lane-wise probe of <= 16 bytes (fmx_device.h)
    distinct = None
    if requested < (1 << 31):                     # (torch.unique sorts through a 32-bit-indexed primitive)
        try:
            distinct = int(torch.unique(ent & 0x7FFFFFFFFFFFFFFF).numel())
        except RuntimeError:
            distinct = None
    del log, ent
    return {"requested_lines": requested, "distinct_lines": distinct, "requested_records": requested - narrow,
            "requested_probes": narrow}


# --------------------------------------------------------------------------------------------
# roofline object
# --------------------------------------------------------------------------------------------
def csrc_hash():
    from fm_index_amd import _lib as L
    return L.csrc_hash()


def stored_traffic(workload_key, leg):
    """fabric bytes per launch from profiles/traffic.json -- only when it was measured on THIS
    source tree (csrc hash), otherwise None."""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            t = json.load(f)
    except (OSError, ValueError):
        return None
    ent = t.get(workload_key, {})
    if ent.get("csrc_hash") != csrc_hash():
        return None
    return ent.get(leg)


def set_miss_lines(r, traffic_bytes, stream_bytes, fetch_kb_raw=None, requests=None):
    """requests that left the L2 per second against the rate of dependent random requests the memory
    system sustains.  `requests`: the counted read requests of the launch (TCC_EA0_RDREQ); the streamed part
    (patterns, offsets: whole 128-byte lines) is taken out.  Without the counted figure: raw FETCH_SIZE / 64 B
    (FETCH_SIZE tallies 64 B per fabric request whatever its size, profiles/microbench/gather_fetch_calibration_r02.txt),
    or (traffic - streamed bytes) / 128 B."""
    t_s = r["avg_kernel_ms"] / 1e3
    r["stream_bytes"] = stream_bytes
    if requests is not None:
        req = max(requests - stream_bytes / float(LINE), 0.0)
    elif fetch_kb_raw:
        req = max(fetch_kb_raw * 1024.0 - stream_bytes / 2.0, 0.0) / 64.0     # streamed lines are 128-B requests too
    else:
        req = max(traffic_bytes - stream_bytes, 0) / LINE
    r["fabric_requests"] = int(req)
    r["fabric_requests_per_s"] = req / t_s
    r["frac_of_gather_ceiling"] = round(req / t_s / (GATHER_CEILING_GLINES * 1e9), 4)


def fabric_read_bytes(ent):
    """read bytes that left the L2s in one launch, from the COUNTED request widths: 32 B x TCC_EA0_RDREQ_32B + 64 B x
    _64B + 128 B x _128B (requests of no counted width -- none on these kernels -- are priced at 64 B, what FETCH_SIZE
    assumes for every request).  None when the launch's widths were not collected."""
    w = ent.get("rdreq") if ent else None
    if not w or w.get("all") is None:
        return None
    r32, r64, r128 = w.get("32B") or 0.0, w.get("64B") or 0.0, w.get("128B") or 0.0
    other = max(w["all"] - r32 - r64 - r128, 0.0)
    return 32.0 * r32 + 64.0 * (r64 + other) + 128.0 * r128


def price_traffic(roof, ent):
    """FABRIC bytes of one launch from its counters -> roof["traffic"], "achieved", "frac".  ONE basis for every
    roofline object (round 6): the read requests of the launch counted BY WIDTH
    (`ent["rdreq"]`: TCC_EA0_RDREQ and its _32B / _64B / _128B parts, one rocprofv3 pass) priced at their widths, plus
    WRITE_SIZE:
        traffic = 32 B x RDREQ_32B + 64 B x RDREQ_64B + 128 B x RDREQ_128B + WRITE_SIZE.
    (Rounds 2-5 priced FETCH_SIZE -- 64 B per request whatever its width on gfx950 -- with the census's share of
    128-byte record requests, which undercounts: a lane-wise 16-byte probe still fetches its 128-byte line, and the
    counters say every fabric read of these kernels is 128 bytes wide.)  An entry without widths (profiles/traffic.json
    of an earlier round: FETCH_SIZE only) is priced by the guide's rule, 2 x FETCH_SIZE + WRITE_SIZE, and the basis
    says so."""
    if not roof or not ent or not (ent.get("fetch_kb_raw") or ent.get("rdreq")):
        return
    t_s = roof["avg_kernel_ms"] / 1e3
    write = (ent.get("write_kb") or 0.0) * 1024.0
    rd = fabric_read_bytes(ent)
    requests = None
    if rd is not None:
        w = ent["rdreq"]
        requests = float(w["all"])
        tb = int(rd + write)
        roof["read_request_widths"] = {k: int(w.get(k) or 0) for k in ("32B", "64B", "128B")}
        roof["read_requests"] = int(requests)
        roof["share_of_128B_requests"] = round((w.get("128B") or 0.0) / requests, 4) if requests else None
        roof["basis"] = ("FABRIC bytes (what left the L2s, Infinity-Cache hits INCLUDED -- not all of it reached HBM, see "
                         "hbm_frac_min): read requests counted by width (32 B x TCC_EA0_RDREQ_32B + 64 B x _64B + 128 B x "
                         "_128B) + WRITE_SIZE, over the kernel time")
        fetch_kb = requests * 64.0 / 1024.0            # what FETCH_SIZE would have reported (= RDREQ x 64 B)
    else:
        fetch_kb = ent["fetch_kb_raw"]
        tb = int(2.0 * fetch_kb * 1024.0 + write)
        roof["basis"] = ("FABRIC bytes by the guide's rule for gfx950: 2 x FETCH_SIZE + WRITE_SIZE (FETCH_SIZE tallies 64 B "
                         "per 128-byte request; no request widths were collected for this entry)")
    roof["traffic"] = tb
    roof["traffic_source"] = ent.get("source")
    if ent.get("kernel"):
        roof["traffic_kernel"] = ent["kernel"]
    roof["fetch_kb_raw"], roof["write_kb"] = round(fetch_kb, 1), ent.get("write_kb")
    roof["achieved"] = round(tb / t_s / 1e9, 1)
    roof["frac"] = round(roof["achieved"] / HBM_PEAK_GBS, 4)
    set_miss_lines(roof, tb, roof.get("stream_bytes", 0), fetch_kb, requests)
    if roof.get("min_bytes"):
        roof["traffic_over_min_bytes"] = round(tb / roof["min_bytes"], 3)
    ic_split(roof)


def ic_split(roof):
    """`frac` prices FABRIC bytes: requests that left an L2, whether the 256 MiB Infinity Cache or HBM served them
    (MI355X_MICROARCH.md: "Infinity-Cache hits appear to be counted, not excluded"; no TCC counter tells the two apart --
    the cache sits on the memory side of the fabric).  What can be said from this side: the reads that reach the fabric
    are the ones the L2s do not hold, spread over `table_bytes` of index (the rows of the late steps / of the walks are
    uniformly random), of which at most IC_BYTES are resident -- so at most IC_BYTES / table_bytes of the fetched bytes
    are Infinity-Cache hits, and at least frac x (1 - that share) of the peak really crossed the HBM interface.
    hbm_frac_min is that lower bound; the same kernel on an index four times the cache (`count_n31`) is the A/B."""
    tb_ = roof.get("table_bytes")
    if not tb_ or roof.get("frac") is None:
        return
    share = min(1.0, IC_BYTES / float(tb_))
    roof["ic_hit_share_max"] = round(share, 4)
    roof["hbm_frac_min"] = round(roof["frac"] * (1.0 - share), 4)


def make_roofline(kernel, avg_kernel_ms, units, ref_bytes_per_unit, stream_bytes, census, traffic, table_bytes=None):
    """HBM roofline of one kernel.  `achieved` / `frac` use what the memory system really moved
    when it was measured (PMC passes, priced by price_traffic) -- real bytes of this layout, below the peak;
    the SURVEY 8d figure (the reference layout's 64-byte blocks) is kept as `algorithmic_ref_bytes` for
    information only."""
    t_s = avg_kernel_ms / 1e3
    r = {"bound": "hbm", "kernel": kernel, "peak": HBM_PEAK_GBS, "unit": "GB/s",
         "avg_kernel_ms": round(avg_kernel_ms, 4), "stream_bytes": stream_bytes}
    if table_bytes:
        r["table_bytes"] = int(table_bytes)    # the index arrays this kernel's random requests spread over
    if census and census.get("requested_lines") is not None:
        r["requested_lines"] = census["requested_lines"]
        r["requested_records"] = census.get("requested_records")
        r["requested_probes"] = census.get("requested_probes")
        nrec = census.get("requested_records")
        nprb = census.get("requested_probes") or 0
        r["requested_bytes"] = (nrec * LINE + nprb * 16 if nrec is not None else census["requested_lines"] * LINE) \
            + stream_bytes
        r["requested_lines_per_s"] = census["requested_lines"] / t_s     # L2 hits included
        if census.get("distinct_lines") is not None:
            r["min_bytes"] = census["distinct_lines"] * LINE + stream_bytes
    r["traffic"] = None
    r["achieved"] = None
    r["frac"] = None
    # the census counts L2 hits too, so requested bytes are not fabric traffic: no number is
    # better than one that can exceed the peak
    r["basis"] = "no PMC traffic for this build (rocprofv3 unavailable and no profiles/traffic.json entry " \
                 "measured on these sources)"
    price_traffic(r, traffic)
    r["algorithmic_ref_bytes"] = units * ref_bytes_per_unit
    r["algorithmic_ref_bytes_per_unit"] = ref_bytes_per_unit
    return r


# --------------------------------------------------------------------------------------------
# live PMC passes: rocprofv3 runs a child of this script; separate --pmc passes, no trace domains
# --------------------------------------------------------------------------------------------
WALK_KERNEL = "fmx_locate_f3u_kernel"    # the default DNA index (text order + walk records): ONE launch per batch (round 5)
LANE_WALK_KERNEL = "fmx_locate_walk_lane_kernel"   # (round 4, measurement builds: batches of 64+ hits per pattern)
PMC_LEGS = {   # leg -> substrings identifying its dominant kernel in the counter CSV
    "dna_count": ["fmx_count_f3_kernel<1, false, false>", "fmx_count_f3_kernel"],
    "dna_count_pair": ["fmx_count_pair_kernel<false>"],          # opt-in accelerators (accel_legs)
    "dna_count_kmer": ["fmx_count_f3_kernel<1, false, true>"],
    "dna_count_both": ["fmx_count_pair_kernel<true>"],
    "dna_locate": [WALK_KERNEL],
    "dna_locate_3b": [WALK_KERNEL, LANE_WALK_KERNEL],     # the same kernel on a larger grid (told apart by the grid)
    "rlfm_count": ["fmx_count_ep_kernel", "fmx_count_kernel"],
    "rlfm_locate": ["fmx_locate_ep_kernel", "fmx_locate_kernel"],
    "rlfm_locate_lane": ["fmx_locate_rl_rounds_kernel", "fmx_locate_rl_lane_kernel"],   # config 4b: 2+ hits per pattern (--workload rep-rlfm)
}


def pmc_aggregate(rows, counter):
    """{kernel: [launches, total]} of one counter from rocprofv3 counter_collection rows.  Launches of the DNA
    walk kernel are keyed by grid as well (it runs two shapes under one name: config 3 with 2^20 hits, config 3b
    with 2.9e8 hits on twice the blocks), and only a kernel's HEAVY launches count: a kernel may also run once
    on a small side batch (the count that prepares config 3b), which must not dilute the per-launch mean."""
    agg = {}
    for row in rows:
        if row.get("Counter_Name") != counter:
            continue
        kn = row.get("Kernel_Name", "?")
        if WALK_KERNEL in kn:
            kn = "%s @grid %s" % (kn, row.get("Grid_Size", "?"))
        agg.setdefault(kn, []).append(float(row.get("Counter_Value", 0) or 0))
    out = {}
    for kn, vals in agg.items():
        heavy = [v for v in vals if v >= 0.9 * max(vals)]
        out[kn] = [len(heavy), sum(heavy)]
    return out


def pmc_grid_of(kn):
    try:
        return int(kn.rsplit("@grid ", 1)[1])
    except (IndexError, ValueError):
        return 0


def pmc_per_dispatch(agg, subs, which="largest"):
    """(kernel name, counter value per launch) of the leg whose kernel matches the first of `subs` present.
    which: "largest" = the instantiation that moved the most; "grid_min" / "grid_max" = the walk kernel's
    launches on its smallest / largest grid ("grid_max" only when two shapes were launched)."""
    for sub in subs:
        cands = [(kn, nd, tot) for kn, (nd, tot) in agg.items() if sub in kn]
        if not cands:
            continue
        if which == "largest" or all(pmc_grid_of(c[0]) == 0 for c in cands):     # (a kernel that is not keyed by grid)
            best = max(cands, key=lambda c: c[2])
        else:
            grids = sorted({pmc_grid_of(c[0]) for c in cands})
            if which == "grid_max" and len(grids) < 2:
                return None, None
            g = grids[0] if which == "grid_min" else grids[-1]
            best = max((c for c in cands if pmc_grid_of(c[0]) == g), key=lambda c: c[2])
        return best[0], best[2] / best[1]
    return None, None


PMC_WHICH = {"dna_locate": "grid_min", "dna_locate_3b": "grid_max"}


def pmc_child(args):
    """the program rocprofv3 profiles: builds the two indexes and runs each leg's kernel a few times."""
    import torch
    local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    reps = 3
    if args.workload != "dna":       # profiles/run_rocprof.sh: another workload's count + locate kernels on their own
        wo = Workload(args.workload, args, dev, local, 0, 1)
        for _ in range(reps):
            wo.count()
        if wo.level is not None:
            wo.prepare_locate()
            for _ in range(reps):
                wo.locate()
        torch.cuda.synchronize()
        wo.close()
        return
    wl = Workload("dna", args, dev, local, 0, 1)
    for _ in range(reps):
        wl.count()
    if wl.level is not None:
        wl.prepare_locate()
        for _ in range(reps):
            wl.locate()
        if not args.no_3b:           # config 3b: the same walk kernel on a 2.9e8-hit batch (larger grid)
            from .locate import setup_3b      # (locate.py imports this module)
            lstep3b = setup_3b(wl)[-1]
            for _ in range(2):
                lstep3b()
            del lstep3b
    if not args.no_accel:            # the opt-in count accelerators on the same patterns
        import fm_index_amd as F
        for kw in (dict(plain=True), dict(plain=True, pair_index=True), dict(plain=True, kmer_table=True)):
            pidx = F.FMIndex.from_device_text(wl.text.data_ptr(), wl.n, wl.maxc, device=local, **kw)
            for _ in range(reps):
                rc = wl.lib.fmx_count_batch_dev(pidx.handle(), C.c_void_p(wl.pat.data_ptr()), C.c_void_p(wl.off.data_ptr()),
                                                wl.npat, None, C.c_void_p(wl.d_s.data_ptr()), C.c_void_p(wl.d_e.data_ptr()),
                                                None, wl.sp)
                assert rc == 0
            torch.cuda.synchronize()
            pidx.close()
    torch.cuda.synchronize()
    wl.close()
    del wl
    torch.cuda.empty_cache()
    if not args.no_rlfm:
        wr = Workload("bytes-rlfm", args, dev, local, 0, 1)
        for _ in range(reps):
            wr.count()
        if wr.level is not None:
            wr.prepare_locate()
            for _ in range(reps):
                wr.locate()
        torch.cuda.synchronize()
        wr.close()


def run_pmc_passes(args, npat=None, count_only=False, seed7=True):
    """returns {leg: {"bytes", "fetch_kb_raw", "write_kb", "kernel", "source"}} or {} when rocprofv3 is
    missing / fails.  FETCH_SIZE and WRITE_SIZE in separate passes (TCC slots), kernel-trace only."""
    exe = shutil.which("rocprofv3")
    if not exe:
        return {}, "rocprofv3 not found"
    out = {}
    raw = {}
    work = tempfile.mkdtemp(prefix="fmx_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    child = [sys.executable, os.path.join(ROOT, "bench.py"), "--pmc-child", "--log2n", str(args.log2n),
             "--npat", str(npat or args.npat), "--plen", str(args.plen), "--level", str(args.level)]
    if args.no_rlfm or count_only:
        child.append("--no-rlfm")
    if args.no_locate or count_only:
        child.append("--no-locate")
    if args.no_3b or count_only:
        child.append("--no-3b")
    if args.no_accel or count_only:
        child.append("--no-accel")
    if count_only and seed7:
        child += ["--pattern-seed", "7"]
    try:
        # two passes (TCC has 4 counter slots: MI355X_MICROARCH.md "rocprofv3 PMC slots"): the read requests that left the
        # L2s with their widths, and WRITE_SIZE.  FETCH_SIZE itself is TCC_EA0_RDREQ x 64 B on gfx950 (the guide's HBM
        # section) -- the first pass holds it, with what it cannot say: how wide the requests were
        for counter, names in (("RDREQ", RDREQ_COUNTERS), ("WRITE_SIZE", ["WRITE_SIZE"])):
            d = os.path.join(work, counter)
            cmd = [exe, "--pmc"] + names + ["--kernel-trace", "-d", d, "--output-format", "csv", "--"] + child
            # its own session: on a timeout the WHOLE group goes (rocprofv3 and the `bench.py --pmc-child` under
            # it, which holds a 2^30 index) and is waited for before the timed run starts
            proc = subprocess.Popen(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                                    start_new_session=True)
            try:
                _, perr = proc.communicate(timeout=420)
            except subprocess.TimeoutExpired:
                try:
                    os.killpg(proc.pid, 9)
                except OSError:
                    pass
                proc.communicate()
                return {}, "rocprofv3 --pmc %s pass timed out after 420 s (process group killed)" % counter
            if proc.returncode != 0:
                return {}, "rocprofv3 --pmc %s failed (rc %d): %s" % (counter, proc.returncode,
                                                                     perr.decode(errors="replace")[-300:])
            rows = []
            for f in glob.glob(os.path.join(d, "**", "*counter_collection*.csv"), recursive=True):
                with open(f, newline="") as fh:
                    rows.extend(csv.DictReader(fh))
            for nm in names:
                raw[nm] = pmc_aggregate(rows, nm)
    except OSError as ex:
        return {}, "rocprofv3 pass did not start: %r" % (ex,)
    finally:
        shutil.rmtree(work, ignore_errors=True)

    def per_dispatch(counter, subs, which="largest"):
        return pmc_per_dispatch(raw.get(counter, {}), subs, which)
    for leg, subs in PMC_LEGS.items():
        which = PMC_WHICH.get(leg, "largest")
        ent = pmc_entry(raw, subs, which)
        if ent is not None:
            out[leg] = ent
    # calibration in our own access pattern: k_mwm_pieces<3> reads the n-byte BWT exactly once
    c = pmc_entry(raw, ["k_mwm_pieces<3"], "largest")
    cal = None
    if c:
        rd = fabric_read_bytes(c)
        cal = {"kernel": "k_mwm_pieces<3>", "fetch_kb_raw": c["fetch_kb_raw"], "read_request_widths": c["rdreq"],
               "expected_bytes": 1 << args.log2n,
               "bytes_per_reported_byte": round((1 << args.log2n) / (c["fetch_kb_raw"] * 1024), 3),     # FETCH_SIZE's factor
               "bytes_per_width_priced_byte": round((1 << args.log2n) / rd, 3) if rd else None}       # ~1.0: widths are exact
    return out, cal


RDREQ_COUNTERS = ["TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum", "TCC_EA0_RDREQ_64B_sum", "TCC_EA0_RDREQ_128B_sum"]


def pmc_entry(raw, subs, which):
    """the counters of one leg's dominant kernel, per launch: {"rdreq": {"all", "32B", "64B", "128B"}, "fetch_kb_raw" (=
    RDREQ x 64 B: what FETCH_SIZE reports on gfx950), "write_kb", "kernel", "source"}; None when the kernel is not in
    the trace.  `raw`: {counter name: pmc_aggregate(...)}."""
    kn, allreq = pmc_per_dispatch(raw.get("TCC_EA0_RDREQ_sum", {}), subs, which)
    if kn is None or allreq is None:
        return None
    w = {"all": round(allreq, 1)}
    for tag in ("32B", "64B", "128B"):
        _, v = pmc_per_dispatch(raw.get("TCC_EA0_RDREQ_%s_sum" % tag, {}), subs, which)
        w[tag] = round(v or 0.0, 1)
    _, write_kb = pmc_per_dispatch(raw.get("WRITE_SIZE", {}), subs, which)
    return {"rdreq": w, "fetch_kb_raw": round(allreq * 64.0 / 1024.0, 1), "write_kb": round(write_kb or 0.0, 1),
            "kernel": kn.split("(")[0].replace("void ", ""), "source": "live rocprofv3 --pmc passes of this run"}



def apply_pmc(out, pmc, cal):
    redo = price_traffic
    if isinstance(cal, str):
        out["pmc"] = {"status": cal}
    elif pmc:
        out["pmc"] = {"status": "ok", "calibration": cal,
                      "note": "separate rocprofv3 --pmc passes (kernel-trace only) over `bench.py --pmc-child`: "
                              "TCC_EA0_RDREQ with its _32B / _64B / _128B parts (FETCH_SIZE = RDREQ x 64 B on gfx950: "
                              "the widths say what the requests really moved), and WRITE_SIZE; see roofline.basis"}
    else:
        out["pmc"] = {"status": "no counters collected"}
    accelerated = "fmx_count_pair_kernel<true>" in ((out.get("roofline") or {}).get("kernel") or "")
    redo(out.get("roofline"), pmc.get("dna_count_both" if accelerated else "dna_count"))
    for leg, key in (("plain", "dna_count"), ("pair_index", "dna_count_pair"), ("kmer_table", "dna_count_kmer")):
        redo((out.get(leg) or {}).get("roofline"), pmc.get(key))
    redo(out.get("locate", {}).get("roofline"), pmc.get("dna_locate"))
    redo(out.get("locate_3b", {}).get("roofline"), pmc.get("dna_locate_3b"))
    redo(out.get("rlfm", {}).get("roofline"), pmc.get("rlfm_count"))
    redo(out.get("rlfm", {}).get("locate", {}).get("roofline"), pmc.get("rlfm_locate"))



def two_stream_roofline(leg):
    """the walk kernel's roofline figures at the rate of two batches in flight: same bytes and requests per launch
    as `roofline` (one launch at a time), over the per-batch time of the two-stream run -- with launches that
    overlap, a launch's share of the wall clock is its duration"""
    if not leg or "two_streams" not in leg or "ms_per_batch" not in leg["two_streams"]:
        return
    r, two = leg.get("roofline") or {}, leg["two_streams"]
    if not r.get("traffic"):
        return
    t_s = two["ms_per_batch"] / 1e3
    two["roofline"] = {"traffic": r["traffic"], "achieved": round(r["traffic"] / t_s / 1e9, 1),
                       "frac": round(r["traffic"] / t_s / 1e9 / HBM_PEAK_GBS, 4),
                       "frac_of_gather_ceiling": round(r.get("fabric_requests", 0) / t_s / (GATHER_CEILING_GLINES * 1e9), 4),
                       "basis": "bytes and fabric requests per launch of `roofline`, over the per-batch time of two "
                                "batches in flight (includes the row expansion)"}

