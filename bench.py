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

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured streaming)
GATHER_CEILING_GLINES = 55.0  # profiles/microbench/gather_r02.txt: dependent random lines the memory
#                               system sustains (16..128-byte requests alike, 128 MiB..2 GiB tables)
LINE = 128


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
    ap.add_argument("--no-pretouch", action="store_true",
                    help="do not have a child process write the device's free memory once before the run (see pretouch_device)")
    ap.add_argument("--no-d2h", action="store_true", help="skip value_incl_d2h")
    ap.add_argument("--no-census", action="store_true", help="skip the requested / distinct line census")
    ap.add_argument("--no-pmc", action="store_true",
                    help="do not run the rocprofv3 --pmc child passes that measure roofline.traffic live")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
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
    ap.add_argument("--dump-counts", default=None, help=argparse.SUPPRESS)   # tests: gathered counts -> .npy
    ap.add_argument("--pattern-seed", type=int, default=None, help=argparse.SUPPRESS)   # tests: same global set at any N
    return ap.parse_args(argv)


# --------------------------------------------------------------------------------------------
# workloads
# --------------------------------------------------------------------------------------------
class Workload:
    """text + index + pattern batch of one BASELINE config, resident in HBM."""

    def __init__(self, name, args, dev, local, rank, world, rlfm=None, with_locate=True, npat=None, plen=None):
        import torch
        import fm_index_amd as F
        from fm_index_amd import workload as W
        from fm_index_amd import _lib as L
        self.torch, self.F, self.W = torch, F, W
        self.lib = L.lib()
        self.name, self.dev, self.local = name, dev, local
        self.n = 1 << args.log2n
        self.dna = name == "dna"
        self.rlfm = name.endswith("rlfm") if rlfm is None else rlfm
        self.maxc = 4 if self.dna else 255
        self.Lbits = 3 if self.dna else 8
        self.m = plen if plen is not None else (args.plen if self.dna else (16 if args.plen == 32 else args.plen))
        self.npat = npat if npat is not None else args.npat
        # patterns are a function of (seed, GLOBAL pattern index) alone, so any sharding of the same global set
        # searches the same patterns: weak scaling = world x npat patterns, rank r owns [r npat, (r+1) npat);
        # strong scaling (--total-patterns T) = T patterns, rank r owns sharding.shard_range(T, r, world)
        self.total_patterns = self.npat * world
        self.pat_lo = rank * self.npat
        self.strong = bool(getattr(args, "total_patterns", None)) and npat is None
        if self.strong:
            from fm_index_amd import sharding
            self.total_patterns = args.total_patterns
            self.pat_lo, hi = sharding.shard_range(self.total_patterns, rank, world)
            self.npat = hi - self.pat_lo
            self.shard_sizes = [sharding.shard_range(self.total_patterns, r, world)[1] -
                                sharding.shard_range(self.total_patterns, r, world)[0] for r in range(world)]
        else:
            self.shard_sizes = [self.npat] * world
        self.npat_pad = max(self.shard_sizes)        # every rank's slot in the gathered buffer
        t0 = time.time()
        if self.dna:
            self.text = W.dna_text_torch(self.n, 1, dev)
        elif name.startswith("rep"):
            self.text = W.repetitive_text_torch(self.n, 5, dev, base_len=1 << 20, mut_per_1024=args.mut_per_1024)
        else:
            self.text = W.byte_text_torch(self.n, 4, dev)
        torch.cuda.synchronize()
        self.textgen_s = time.time() - t0
        self.level = args.level if (with_locate and not args.no_locate) else None
        if self.rlfm:
            cls = F.RLFMIndexWithLocate if self.level is not None else F.RLFMIndex
        else:
            cls = F.FMIndexWithLocate if self.level is not None else F.FMIndex
        self.index = cls.from_device_text(self.text.data_ptr(), self.n, self.maxc, level=self.level, device=local)
        self.h = self.index.handle()
        self.build_ms = self.lib.fmx_build_ms(self.h)
        # global pattern set = world * npat substrings of the text; this rank owns a contiguous shard
        seed = (3 if (world == 1 and not self.strong) else 7) if self.dna else 6
        if args.pattern_seed is not None:
            seed = args.pattern_seed
        self.pattern_seed = seed
        z = W.splitmix64_torch(seed, self.pat_lo, self.npat, dev)
        self.src_pos = W.umod_torch(z, self.n - 1 - self.m)
        idx2d = self.src_pos[:, None] + torch.arange(self.m, dtype=torch.int64, device=dev)[None, :]
        self.pat = self.text[idx2d].reshape(-1).contiguous()
        del idx2d
        self.off = (torch.arange(self.npat + 1, dtype=torch.int64, device=dev) * self.m).contiguous()
        self.d_s = torch.empty(self.npat, dtype=torch.int64, device=dev)
        self.d_e = torch.empty(self.npat, dtype=torch.int64, device=dev)
        self.d_c = torch.empty(self.npat, dtype=torch.int64, device=dev)
        self.stream = torch.cuda.current_stream()
        self.sp = C.c_void_p(self.stream.cuda_stream)

    # SURVEY 8d reference figure: 2 endpoints x L levels x 64 B (FM); 2 x (2L+4) probes x 64 B (RLFM)
    def ref_bytes_per_char(self):
        return 2 * (2 * self.Lbits + 4) * 64 if self.rlfm else 2 * self.Lbits * 64

    def count(self, out_cnt=None, lib=None, pat=None):
        lib = lib or self.lib
        oc = self.d_c if out_cnt is None else out_cnt
        p = self.pat if pat is None else pat
        rc = lib.fmx_count_batch_dev(self.h, C.c_void_p(p.data_ptr()), C.c_void_p(self.off.data_ptr()), self.npat,
                                     None, C.c_void_p(self.d_s.data_ptr()), C.c_void_p(self.d_e.data_ptr()),
                                     C.c_void_p(oc.data_ptr()), self.sp)
        if rc != 0:
            raise RuntimeError(lib.fmx_last_error().decode())

    def prepare_locate(self):
        torch = self.torch
        self.d_off = torch.empty(self.npat + 1, dtype=torch.int64, device=self.dev)
        self.lib.fmx_offsets_dev(self.h, C.c_void_p(self.d_s.data_ptr()), C.c_void_p(self.d_e.data_ptr()), self.npat,
                                 C.c_void_p(self.d_off.data_ptr()), self.sp)
        self.total_hits = int(self.d_off[-1].item())
        self.d_pos = torch.empty(max(self.total_hits, 1), dtype=torch.int64, device=self.dev)

    def locate(self, lib=None, out=None):
        lib = lib or self.lib
        dst = self.d_pos if out is None else out
        rc = lib.fmx_locate_batch_dev(self.h, C.c_void_p(self.d_s.data_ptr()), C.c_void_p(self.d_e.data_ptr()),
                                      self.npat, C.c_void_p(self.d_off.data_ptr()), self.total_hits,
                                      C.c_void_p(dst.data_ptr()), self.sp)
        if rc != 0:
            raise RuntimeError(lib.fmx_last_error().decode())

    def timed_kernel(self, fn):
        """one launch with the library's own HIP events around the kernel (launch stream)."""
        self.lib.fmx_set_timing(self.h, 1)
        fn()
        self.torch.cuda.synchronize()
        ms = self.lib.fmx_last_kernel_ms(self.h)
        steps = int(self.lib.fmx_last_steps(self.h))
        self.lib.fmx_set_timing(self.h, 0)
        return ms, steps

    def series_kernel_ms(self, fn, reps, handle=None):
        """mean duration of the dominant kernel over `reps` launches issued BACK TO BACK (fmx_set_timing(h, 2): a pair
        of HIP events around the kernel of every launch, on the launch stream, no synchronisation in between) -- the
        launch duration of the timed region, where timed_kernel() measures a launch that starts on an idle device"""
        h = handle or self.h
        reps = min(int(reps), 64)
        self.lib.fmx_set_timing(h, 2)
        for _ in range(reps):
            fn()
        self.torch.cuda.synchronize()
        ms = float(self.lib.fmx_series_kernel_ms(h))
        self.lib.fmx_set_timing(h, 0)
        return ms if ms > 0 else None

    def describe(self, world):
        if self.dna:
            if self.strong:
                return ("config5: FMIndex count, n=2^%d sigma=4 DNA text (L=3), %d x len-%d substring patterns (seed %d) in "
                        "contiguous shards over %d GPU(s), index replicated, counts all-gathered every step"
                        % (self.n.bit_length() - 1, self.total_patterns, self.m, self.pattern_seed, world))
            w = "config2: FMIndex count, n=2^%d sigma=4 DNA text (L=3), %d x len-%d substring patterns per GPU"
        elif self.name.startswith("rep"):
            w = "config4b (" + self.name + "): n=2^%d repetitive byte text (L=8), %d x len-%d substring patterns per GPU"
        else:
            w = "config4 (" + ("RLFMIndex" if self.rlfm else "FMIndex") + \
                "): n=2^%d sigma=255 byte text (L=8), %d x len-%d substring patterns per GPU"
        return w % (self.n.bit_length() - 1, self.npat, self.m)

    def close(self):
        if getattr(self, "_oracle", None) is not None:
            self._oracle[0].close()
            self._oracle = None
        self.index.close()


def event_time_ms(torch, stream, fn, steps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(steps):
        fn()
    e1.record(stream)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps


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
    """HBM-side bytes per launch from profiles/traffic.json -- only when it was measured on THIS
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


def set_miss_lines(r, traffic_bytes, stream_bytes, fetch_kb_raw=None):
    """requests that left the L2 per second against the rate of dependent random requests the memory
    system sustains.  FETCH_SIZE tallies 64 B per fabric request whatever its size (calibrated:
    profiles/microbench/gather_fetch_calibration_r02.txt -- 16-, 32-, 64- and 128-byte random requests
    all report 64 B), so requests = raw FETCH_SIZE / 64 B; without the raw counter, (traffic -
    streamed bytes) / 128 B."""
    t_s = r["avg_kernel_ms"] / 1e3
    r["stream_bytes"] = stream_bytes
    if fetch_kb_raw:
        req = max(fetch_kb_raw * 1024.0 - stream_bytes / 2.0, 0.0) / 64.0     # streamed lines are 128-B requests too
    else:
        req = max(traffic_bytes - stream_bytes, 0) / LINE
    r["fabric_requests"] = int(req)
    r["fabric_requests_per_s"] = req / t_s
    r["frac_of_gather_ceiling"] = round(req / t_s / (GATHER_CEILING_GLINES * 1e9), 4)


def price_traffic(roof, ent):
    """HBM-side bytes of one launch from its counters (`ent`: fetch_kb_raw, write_kb, source) -> roof["traffic"],
    "achieved", "frac".  gfx950's FETCH_SIZE reports 64 B per fabric request whatever its size.  A request
    for a whole 128-byte record therefore moved twice what the counter says; a lane-wise probe (<= 16 bytes:
    B / B' pieces, select blocks, positions, phase pieces, samples) moved the 64 B it reports.  The census
    (same kernels, every request logged with its width) gives the share of record requests among the
    requests of this launch, and
        traffic = FETCH_SIZE x (1 + record share) + WRITE_SIZE.
    For the all-record kernels (DNA count) that is the guide's 2 x FETCH_SIZE; for the run-length kernels,
    half of whose requests are 16-byte probes, 2 x FETCH_SIZE overstated the bytes -- it is
    kept as `traffic_upper`.  Without a census the upper bound is all there is, and `frac` says so."""
    if not roof or not ent or not ent.get("fetch_kb_raw"):
        return
    t_s = roof["avg_kernel_ms"] / 1e3
    fetch, write = ent["fetch_kb_raw"] * 1024.0, (ent.get("write_kb") or 0.0) * 1024.0
    rec, prb = roof.get("requested_records"), roof.get("requested_probes")
    share = rec / (rec + prb) if rec is not None and prb is not None and rec + prb > 0 else None
    upper = int(2.0 * fetch + write)
    tb = int(fetch * (1.0 + share) + write) if share is not None else upper
    roof["traffic"] = tb
    roof["traffic_upper"] = upper
    roof["record_share_of_requests"] = round(share, 4) if share is not None else None
    roof["traffic_source"] = ent.get("source")
    if ent.get("kernel"):
        roof["traffic_kernel"] = ent["kernel"]
    roof["fetch_kb_raw"], roof["write_kb"] = ent["fetch_kb_raw"], ent.get("write_kb")
    roof["achieved"] = round(tb / t_s / 1e9, 1)
    roof["frac"] = round(roof["achieved"] / HBM_PEAK_GBS, 4)
    roof["frac_upper"] = round(upper / t_s / 1e9 / HBM_PEAK_GBS, 4)
    set_miss_lines(roof, tb, roof.get("stream_bytes", 0), ent["fetch_kb_raw"])
    roof["basis"] = ("HBM-side bytes of the PMC counters: FETCH_SIZE x (1 + share of 128-byte record requests, from "
                     "the census) + WRITE_SIZE, over the kernel time" if share is not None else
                     "UPPER BOUND: 2 x FETCH_SIZE + WRITE_SIZE (no census of request widths in this run)")
    if roof.get("min_bytes"):
        roof["traffic_over_min_bytes"] = round(tb / roof["min_bytes"], 3)


def make_roofline(kernel, avg_kernel_ms, units, ref_bytes_per_unit, stream_bytes, census, traffic):
    """HBM roofline of one kernel.  `achieved` / `frac` use what the memory system really moved
    when it was measured (PMC passes, priced by price_traffic) -- real bytes of this layout, below the peak;
    the SURVEY 8d figure (the reference layout's 64-byte blocks) is kept as `algorithmic_ref_bytes` for
    information only."""
    t_s = avg_kernel_ms / 1e3
    r = {"bound": "hbm", "kernel": kernel, "peak": HBM_PEAK_GBS, "unit": "GB/s",
         "avg_kernel_ms": round(avg_kernel_ms, 4), "stream_bytes": stream_bytes}
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
    # the census counts L2 hits too, so requested bytes are not HBM-side traffic: no number is
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
WALK_KERNEL = "fmx_locate_f3t_kernel"    # the default DNA index: text order + walk records (round 4)
LANE_WALK_KERNEL = "fmx_locate_walk_lane_kernel"   # the same index on batches of 64+ hits per pattern (config 3b)
PMC_LEGS = {   # leg -> substrings identifying its dominant kernel in the counter CSV
    "dna_count": ["fmx_count_f3_kernel<1, false, false>", "fmx_count_f3_kernel"],
    "dna_count_pair": ["fmx_count_pair_kernel<false>"],          # opt-in accelerators (accel_legs)
    "dna_count_kmer": ["fmx_count_f3_kernel<1, false, true>"],
    "dna_count_both": ["fmx_count_pair_kernel<true>"],
    "dna_locate": [WALK_KERNEL],
    "dna_locate_3b": [LANE_WALK_KERNEL, WALK_KERNEL],
    "rlfm_count": ["fmx_count_ep_kernel", "fmx_count_kernel"],
    "rlfm_locate": ["fmx_locate_ep_kernel", "fmx_locate_kernel"],
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
    wl = Workload("dna", args, dev, local, 0, 1)
    for _ in range(reps):
        wl.count()
    if wl.level is not None:
        wl.prepare_locate()
        for _ in range(reps):
            wl.locate()
        if not args.no_3b:           # config 3b: the same walk kernel on a 2.9e8-hit batch (larger grid)
            lstep3b = setup_3b(wl)[-1]
            for _ in range(2):
                lstep3b()
            del lstep3b
    if not args.no_accel:            # the opt-in count accelerators on the same patterns
        import fm_index_amd as F
        for kw in (dict(pair_index=True), dict(kmer_table=True), dict(auto=True)):
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


def run_pmc_passes(args, npat=None, count_only=False):
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
    if count_only:
        child += ["--pattern-seed", "7"]
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(work, counter)
            cmd = [exe, "--pmc", counter, "--kernel-trace", "-d", d, "--output-format", "csv", "--"] + child
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
            raw[counter] = pmc_aggregate(rows, counter)
    except OSError as ex:
        return {}, "rocprofv3 pass did not start: %r" % (ex,)
    finally:
        shutil.rmtree(work, ignore_errors=True)

    def per_dispatch(counter, subs, which="largest"):
        return pmc_per_dispatch(raw.get(counter, {}), subs, which)
    for leg, subs in PMC_LEGS.items():
        which = PMC_WHICH.get(leg, "largest")
        kn, fetch_kb = per_dispatch("FETCH_SIZE", subs, which)
        _, write_kb = per_dispatch("WRITE_SIZE", subs, which)
        if kn is None or fetch_kb is None:
            continue
        # gfx950: FETCH_SIZE tallies 128-byte requests at 64 B -> x2 (MI355X_MICROARCH.md, HBM section;
        # re-calibrated below on a kernel with a known byte count); WRITE_SIZE reads exactly
        out[leg] = {"fetch_kb_raw": round(fetch_kb, 1), "write_kb": round(write_kb or 0.0, 1),
                    "kernel": kn.split("(")[0].replace("void ", ""),
                    "source": "live rocprofv3 --pmc passes of this run"}
    # calibration in our own access pattern: k_mwm_pieces<3> reads the n-byte BWT exactly once
    kn, kb = per_dispatch("FETCH_SIZE", ["k_mwm_pieces<3"])
    cal = None
    if kb:
        cal = {"kernel": "k_mwm_pieces<3>", "fetch_kb_raw": round(kb, 1), "expected_bytes": 1 << args.log2n,
               "bytes_per_reported_byte": round((1 << args.log2n) / (kb * 1024), 3)}
    return out, cal


# --------------------------------------------------------------------------------------------
# CPU baseline: the oracle (port of the reference algorithm) on this box's cores
# --------------------------------------------------------------------------------------------
def host_cpu():
    """what the CPU column really ran on: the cores this process may use (affinity mask, capped by the
    cgroup CPU quota -- os.cpu_count() sees neither), the CPU model, sockets and threads per core"""
    info = {"os_cpu_count": os.cpu_count()}
    try:
        info["affinity"] = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        info["affinity"] = os.cpu_count() or 1
    quota = None
    try:                                    # cgroup v2
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(per)
    except (OSError, ValueError):
        try:                                # cgroup v1
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except (OSError, ValueError):
            pass
    info["cgroup_cpu_quota"] = quota
    cores = info["affinity"]
    if quota is not None:
        cores = max(1, min(cores, int(quota)))
    info["effective_cpus"] = cores
    model, phys, siblings, cpu_cores = None, set(), None, None
    try:
        for ln in open("/proc/cpuinfo"):
            k, _, v = ln.partition(":")
            k, v = k.strip(), v.strip()
            if k == "model name" and model is None:
                model = v
            elif k == "physical id":
                phys.add(v)
            elif k == "siblings" and siblings is None:
                siblings = int(v)
            elif k == "cpu cores" and cpu_cores is None:
                cpu_cores = int(v)
    except (OSError, ValueError):
        pass
    info["cpu_model"] = model
    info["sockets"] = len(phys) or None
    info["threads_per_core"] = (siblings // cpu_cores) if siblings and cpu_cores else None
    info["physical_cores"] = (len(phys) * cpu_cores) if phys and cpu_cores else None
    return info


def cpu_baseline(wl, args, kind):
    """the oracle (CPU port of the reference algorithm) on this box's cores: the headline number on the CPUs this
    process may really use (affinity mask capped by the cgroup CPU quota), a thread sweep through and beyond that
    number with the parallel efficiency, and the single-thread rate (the reference itself is single-threaded).
    The threads of a batch pin themselves one per CPU, spread evenly over the allowed CPUs
    (oracle/fm_oracle.c: orc_set_thread_spread); the oracle's bit planes are first touched by the static thread
    decomposition that fills them, so their pages are spread over the sockets like the threads that probe them
    at random."""
    import numpy as np
    oi, t_ob = wl_oracle(wl, kind)
    host = host_cpu()
    cores = host["effective_cpus"]
    oi.set_thread_spread(True)
    m = wl.m
    pat_h = wl.pat.cpu().numpy()
    s_h = wl.d_s.cpu().numpy().view(np.uint64)
    e_h = wl.d_e.cpu().numpy().view(np.uint64)

    def cpu_run(k, threads):
        offk = np.arange(k + 1, dtype=np.uint64) * np.uint64(m)
        t = time.perf_counter()
        so, eo = oi.count_batch(pat_h[:k * m], offk, nthreads=threads)
        return time.perf_counter() - t, so, eo
    budget = args.cpu_seconds
    k0 = min(1 << 14, wl.npat)
    t_probe, so, eo = cpu_run(k0, cores)
    k = int(min(wl.npat, max(k0, k0 * budget / 5 / max(t_probe, 1e-6))))
    t_all, so, eo = cpu_run(k, cores)
    assert (so == s_h[:k]).all() and (eo == e_h[:k]).all(), "GPU != oracle on the CPU sample"
    times = [t_all]
    while sum(times) < budget * 0.5 and len(times) < 25:
        times.append(cpu_run(k, cores)[0])
    t_all = sorted(times)[len(times) // 2]
    value = k * m / t_all
    # one thread, then the sweep: each point ~ budget / 12 seconds of work at the rate of the point before
    k1 = max(1024, int(k0 * (budget / 12) / max(t_probe * cores, 1e-6)))
    k1 = min(k1, wl.npat)
    t_one, _, _ = cpu_run(k1, 1)
    one = k1 * m / t_one
    sweep = [{"threads": 1, "value": one, "scaling_vs_1t": 1.0, "parallel_efficiency": 1.0}]
    rate = one
    # through the effective CPU count and beyond it, up to every CPU the affinity mask shows: where the curve
    # flattens is what this box gives this process, whatever os.cpu_count() says
    phys = host.get("physical_cores") or cores
    limit = host["affinity"]
    eff_cores = min(cores, phys)
    for th in sorted({t for t in (2, 4, 8, 16, 32, 64, 128, phys, cores, limit) if 1 < t <= limit}):
        kk = int(min(wl.npat, max(2048, rate * min(th / sweep[-1]["threads"], 2.0) * (budget / 16) / m)))
        dt, _, _ = cpu_run(kk, th)
        rate = kk * m / dt
        sweep.append({"threads": th, "value": rate, "scaling_vs_1t": round(rate / one, 2),
                      "parallel_efficiency": round(rate / one / min(th, eff_cores), 3)})
    best = max(sweep, key=lambda p: p["value"])
    team = oi.team_size(cores)
    oi.set_thread_spread(False)
    return {"value": max(value, best["value"]), "unit": "pattern-chars/s", "cores": cores, "threads_used": team or cores,
            "kind": "port", "cpu_model": host["cpu_model"], "sockets": host["sockets"],
            "physical_cores": host["physical_cores"], "threads_per_core": host["threads_per_core"],
            "host": {k_: host[k_] for k_ in ("os_cpu_count", "affinity", "cgroup_cpu_quota", "effective_cpus")},
            "placement": "one thread per CPU, spread evenly over the allowed CPUs (sched_setaffinity per batch)",
            "sample": "first %d of the %d patterns (same text, same %s built from the index's exported "
                      "BWT), median of %d runs of %.2f s on %d threads; GPU (s,e) bit-identical on the sample"
                      % (k, wl.npat, "RLFM structure" if kind == "rlfm" else "wavelet matrix", len(times), t_all, cores),
            "all_threads_value": value, "best_threads": best["threads"],
            "single_thread_value": one, "scaling_vs_1t": round(max(value, best["value"]) / one, 2),
            "parallel_efficiency": round(max(value, best["value"]) / one / eff_cores, 3),
            "parallel_efficiency_note": "best rate / (single-thread rate x effective CPUs = min(affinity, cgroup quota, "
                                        "physical cores))",
            "thread_sweep": sweep, "oracle_build_s": round(t_ob, 1)}


# --------------------------------------------------------------------------------------------
# config 5: one hash for "multi-GPU output identical to 1-GPU output"
# --------------------------------------------------------------------------------------------
def counts_sha256(counts):
    """sha256 over the per-pattern counts of the whole global pattern set, input order, as little-endian int64"""
    import hashlib
    import numpy as np
    a = np.ascontiguousarray(np.asarray(counts).astype("<i8", copy=False))
    return hashlib.sha256(a.tobytes()).hexdigest()


def ranges_sha256(s, e):
    """sha256 over the (s, e) pairs of the whole global pattern set, input order: [s_0, e_0, s_1, e_1, ...] as
    little-endian int64 -- the search ranges themselves (wrapper.rs:126-129), not only their widths"""
    import hashlib
    import numpy as np
    a = np.empty((len(s), 2), dtype="<i8")
    a[:, 0] = np.asarray(s).astype("<i8", copy=False)
    a[:, 1] = np.asarray(e).astype("<i8", copy=False)
    return hashlib.sha256(a.tobytes()).hexdigest()


def golden_key(workload, log2n, seed, total, m):
    return "%s:n=2^%d:seed=%d:patterns=%d:len=%d" % (workload, log2n, seed, total, m)


def golden_counts_sha(wl, args, total=None, seed=None):
    """the committed hash of this global pattern set's counts (tests/golden/config5_counts.json: computed by the CPU
    oracle over ALL patterns, tests/golden/make_config5_golden.py), or None when this set has no entry"""
    try:
        with open(os.path.join(ROOT, "tests", "golden", "config5_counts.json")) as f:
            g = json.load(f)
    except (OSError, ValueError):
        return None
    ent = g.get("entries", {}).get(golden_key(wl.name, args.log2n, wl.pattern_seed if seed is None else seed,
                                               wl.total_patterns if total is None else total, wl.m))
    return (ent["counts_sha256"], ent.get("ranges_sha256")) if ent else None


def golden_locate(wl, args):
    """{"level", "hits", "positions_sha256"} of this global pattern set in tests/golden/config5_counts.json (the ORDERED
    positions of every pattern, from the CPU oracle), or None"""
    try:
        with open(os.path.join(ROOT, "tests", "golden", "config5_counts.json")) as f:
            g = json.load(f)
    except (OSError, ValueError):
        return None
    ent = g.get("entries", {}).get(golden_key(wl.name, args.log2n, wl.pattern_seed, wl.total_patterns, wl.m))
    loc = (ent or {}).get("locate")
    return loc if loc and loc.get("level") == wl.level else None


def positions_sha256(pos):
    import hashlib
    import numpy as np
    return hashlib.sha256(np.ascontiguousarray(np.asarray(pos).astype("<i8", copy=False)).tobytes()).hexdigest()


def wl_oracle(wl, kind):
    """the CPU oracle of this workload's index (built once per workload from the index's exported BWT / C array:
    test infrastructure, used only by the cpu_baseline leg and the oracle sample of config5_g1) -> (index, build s)"""
    if getattr(wl, "_oracle", None) is None:
        from oracle import fm_oracle as O
        t0 = time.time()
        bwt = wl.index.export_bwt()
        cs = wl.index.export_cs()
        wl._oracle = (O.OracleIndex.from_bwt(bwt, cs, wl.maxc, native=True, kind=kind), time.time() - t0)
        del bwt
    return wl._oracle


# --------------------------------------------------------------------------------------------
def main():
    args = parse_args()
    if args.gpus > 1 and "RANK" not in os.environ:
        # one process per GPU, started before anything in THIS process touches the GPU
        from fm_index_amd import launcher
        sys.exit(launcher.spawn_ranks(args.gpus, [os.path.abspath(__file__)] + sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and not args.pmc_child:
        sys.exit("bench.py: --gpus %d but WORLD_SIZE=%d (start it as `python bench.py --gpus N`, or under "
                 "torch.distributed.run with --nproc-per-node equal to --gpus)" % (args.gpus, world))
    if args.pmc_child:
        pmc_child(args)
        return
    # HBM-side traffic of this build, measured now: rocprofv3 --pmc passes over a child of this
    # script.  Started BEFORE this process touches the GPU (no torch import yet): the children are
    # then spawned from a process that holds no device state.
    pmc = None
    rank = int(os.environ.get("RANK", "0"))
    if (world == 1 and not args.total_patterns and args.workload == "dna" and not args.no_wide and args.log2n >= 30
            and not args.no_pretouch):
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
    run(args, world, pmc, pmc_seconds=time.perf_counter() - t_pmc)


PRETOUCH = {}


def pretouch_device(device):
    """Memory no process has used since the box booted is handed out on a slow path by this driver: a hipMalloc that
    follows the first touch of such pages costs ~28 ms per GiB touched (benchmarks/gpu/alloc_probe2.hip,
    profiles/r04/alloc_probe2.txt: 0.3 ms for the first 34 GiB buffer, 965 ms for each further one), which is what the
    `wide` leg's builder -- 137 GB of scratch in a handful of buffers -- met on the driver's fresh box (build_ms 2 980 in
    round 3, 5 416 in round 4 against 570-850 on a box whose memory an earlier process had used).  A child process
    that allocates what is free, writes it once and exits puts the box into the state of a machine that has been up
    for a while; its cost is reported (`wide.pretouch`), the builder's work is unchanged.  --no-pretouch skips it."""
    code = (
        "import ctypes as C, time, sys\n"
        "h = C.CDLL('libamdhip64.so')\n"
        "t0 = time.time()\n"
        "assert h.hipSetDevice(%d) == 0\n"
        "fr, tot = C.c_size_t(), C.c_size_t()\n"
        "assert h.hipMemGetInfo(C.byref(fr), C.byref(tot)) == 0\n"
        "n = max(fr.value - (4 << 30), 0)\n"
        "p = C.c_void_p()\n"
        "assert h.hipMalloc(C.byref(p), C.c_size_t(n)) == 0\n"
        "assert h.hipMemset(p, 0, C.c_size_t(n)) == 0\n"
        "assert h.hipDeviceSynchronize() == 0\n"
        "print('%%.1f %%.2f' %% (n / 2.0 ** 30, time.time() - t0))\n" % device)
    t0 = time.perf_counter()
    try:
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
        gib, secs = (r.stdout.split() + ["0", "0"])[:2] if r.returncode == 0 else ("0", "0")
        PRETOUCH.update({"gib": float(gib), "child_seconds": float(secs), "seconds": round(time.perf_counter() - t0, 2),
                         "returncode": r.returncode, "error": r.stderr[-300:] if r.returncode else None})
    except Exception as ex:  # noqa: BLE001 -- an optional preparation step
        PRETOUCH.update({"gib": 0.0, "error": repr(ex)})


def run(args, world, pmc=None, pmc_seconds=0.0):
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
    kname = {"dna": "fmx_count_f3_kernel<1,false,false>"}.get(
        args.workload, "fmx_count_ep_kernel" if wl.rlfm else "fmx_count_kernel<FMX_KIND_FM>")
    roofline = make_roofline(kname, avg_kernel_ms, chars_per_step_rank, wl.ref_bytes_per_char(), stream_bytes,
                             cen, stored_traffic(key, "count"))

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
                   "textgen_s": round(wl.textgen_s, 2)},
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
    # ---- opt-in accelerators: same patterns, results asserted identical to the plain index ----
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

    # ---- HBM-side traffic measured by the counter passes at the start of this run ----
    if pmc is not None and rank == 0:
        apply_pmc(out, pmc[0], pmc[1])
    two_stream_roofline(out.get("locate"))
    two_stream_roofline((out.get("rlfm") or {}).get("locate"))

    if use_dist:             # every rank is done before the line is printed; nothing follows it on stdout
        dist.barrier()
        dist.destroy_process_group()
    flush_c_stdio()
    if rank == 0:
        out["leg_seconds"] = leg_seconds
        print(json.dumps(out))
        sys.stdout.flush()


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


def flush_c_stdio():
    try:
        C.CDLL(None).fflush(None)
    except Exception:  # noqa: BLE001
        pass


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


def apply_pmc(out, pmc, cal):
    redo = price_traffic
    if isinstance(cal, str):
        out["pmc"] = {"status": cal}
    elif pmc:
        out["pmc"] = {"status": "ok", "calibration": cal,
                      "note": "separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (kernel-trace only) over "
                              "`bench.py --pmc-child`; gfx950: FETCH_SIZE reports 64 B per request -> x (1 + share of "
                              "128-byte record requests), see roofline.basis"}
    else:
        out["pmc"] = {"status": "no counters collected"}
    redo(out.get("roofline"), pmc.get("dna_count"))
    for leg, key in (("pair_index", "dna_count_pair"), ("kmer_table", "dna_count_kmer"), ("kmer_table+pair_index", "dna_count_both")):
        redo((out.get(leg) or {}).get("roofline"), pmc.get(key))
    redo(out.get("locate", {}).get("roofline"), pmc.get("dna_locate"))
    redo(out.get("locate_3b", {}).get("roofline"), pmc.get("dna_locate_3b"))
    redo(out.get("rlfm", {}).get("roofline"), pmc.get("rlfm_count"))
    redo(out.get("rlfm", {}).get("locate", {}).get("roofline"), pmc.get("rlfm_locate"))


def accel_legs(out, wl, args, rflat):
    """the opt-in count accelerators on the config-2 patterns: pair index, k-mer start table, and both -- the last one
    built with FMX_FLAG_AUTO (the builder adds both when the index qualifies and the device has room) and reported as
    `value_auto`.  (s, e) asserted identical to the plain index on all patterns; each leg gets the roofline object of
    the headline (census of its own launch, counters of its own kernel from the live PMC passes)."""
    torch, F, lib = wl.torch, wl.F, wl.lib
    legs = []
    if wl.dna:
        legs.append(("pair_index", dict(pair_index=True), "opt-in FMX_FLAG_PAIR_INDEX", "fmx_count_pair_kernel<false>"))
    legs.append(("kmer_table", dict(kmer_table=True), "opt-in FMX_FLAG_KMER_TABLE",
                 "fmx_count_f3_kernel<1,false,true>" if wl.dna else "fmx_count_ep_kernel<..., true>"))
    if wl.dna:
        legs.append(("kmer_table+pair_index", dict(auto=True),
                     "FMX_FLAG_AUTO: the builder added FMX_FLAG_KMER_TABLE | FMX_FLAG_PAIR_INDEX (DNA-like FM index, "
                     "n >= 2^24, four times the index free on the device)", "fmx_count_pair_kernel<true>"))
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
            if leg_kw.get("auto") and not (pidx.kmer_k() and pidx.has_pair_index()):
                out[leg_name] = {"skipped": "FMX_FLAG_AUTO left the index plain (n < 2^24, or not enough free HBM)"}
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
            assert bool((ps == wl.d_s).all()) and bool((pe == wl.d_e).all()), leg_name + " != plain index"
            cen = None
            if not args.no_census:
                cen = run_census(wl, lambda cl: pstep(wl.pat, cl), npat * m * 3 + (1 << 20))
            out[leg_name] = {"value": npat * m / (pms / 1e3), "unit": "pattern-chars/s", "ms_per_step": pms,
                             "index_bytes": pidx.heap_size(), "kmer_k": pidx.kmer_k(),
                             "pair_index": pidx.has_pair_index(),
                             "build_ms": round(float(lib.fmx_build_ms(pidx.handle())), 1),
                             "note": leg_note + "; (s,e) identical to the plain-index run",
                             "roofline": make_roofline(kname, pms, npat * m, wl.ref_bytes_per_char(), stream_bytes, cen,
                                                       None)}
            if leg_kw.get("auto"):
                out["value_auto"] = out[leg_name]["value"]
            if rflat is not None:
                # config 2b patterns (uniform random, mostly absent) through the same index
                wl.count(pat=rflat)
                for _ in range(args.warmup):
                    pstep(rflat)
                torch.cuda.synchronize()
                rms2 = event_time_ms(torch, wl.stream, lambda: pstep(rflat), args.steps)
                assert bool((ps == wl.d_s).all()) and bool((pe == wl.d_e).all()), leg_name + " != plain index (2b)"
                out[leg_name]["early_exit_ms_per_step"] = rms2
                out[leg_name]["early_exit_offered_chars_per_s"] = npat * m / (rms2 / 1e3)
                wl.count()                            # restore the config-2 (s, e)
                torch.cuda.synchronize()
            pidx.close()
        except Exception as ex:  # noqa: BLE001 -- never lose the headline line to an optional leg
            out[leg_name] = {"error": repr(ex)}


def dna_walk_kernel(wl):
    return "fmx_locate_f3t_kernel<4>" if wl.index.walk_records() else "fmx_locate_f3p_kernel<4>"


def dna_walk_kernel_long(wl):
    """batches that average 64+ hits per pattern on an index with walk records: a lane per walk on consecutive hits"""
    return LANE_WALK_KERNEL if wl.index.walk_records() else "fmx_locate_f3p_kernel<4>"


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
    kname = dna_walk_kernel(wl) if wl.dna else ("fmx_locate_ep_kernel" if wl.rlfm else "fmx_locate_kernel<FMX_KIND_FM>")
    roof = make_roofline(kname, kavg_ms, 1, ref_bytes, total_hits * 4 + total_hits * 8, cen,
                         stored_traffic(key, "locate"))
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
    # HBM-side traffic of this launch (told from the config-3 launches of the same kernel by its grid)
    # request widths by construction (what the census counts for config 3): one record per LF step, one sample per hit
    # (walk records: max(phase, 1) records per hit, phase = position mod 2^level -- one more than the LF steps for the
    # hits that sit on a sampled position)
    nrec = lf_steps + (int(((pos[:total] & ((1 << wl.level) - 1)) == 0).sum().item()) if wl.index.walk_records() else 0)
    widths = {"requested_lines": nrec + total, "requested_records": nrec, "requested_probes": total,
              "distinct_lines": None}
    if wl.index.walk_records():
        # the lane-per-walk kernel reads a record as lane-wise 16-byte pieces -- the row's own, the pieces in front of it
        # (3 on average), the counter's -- so every request is a probe: about 5 per record visit, one per sample
        widths = {"requested_lines": 5 * nrec + total, "requested_records": 0, "requested_probes": 5 * nrec + total,
                  "distinct_lines": None}
    out["locate_3b"]["requested_lines"] = widths["requested_lines"]
    out["locate_3b"]["requested_lines_per_s"] = widths["requested_lines"] / (kms / 1e3)
    out["locate_3b"]["bound"] = ("not HBM: the hits of a pattern are adjacent rows, and LF keeps rows of one symbol adjacent -- "
                                 "their records (and, in text order, their samples: consecutive entries) come from the "
                                 "caches, so few requests reach the fabric (roofline.fabric_requests).  The group-cooperative "
                                 "walk was bound by vector-instruction issue here (9.7 ms: ~12 wave instructions per walk "
                                 "step, 8 walks per instruction); since round 4 batches of 64+ hits per pattern take "
                                 "fmx_locate_walk_lane_kernel: a lane decodes its row's record alone, 64 walks per instruction")
    out["locate_3b"]["roofline"] = make_roofline(dna_walk_kernel_long(wl), kms, 1, lf_steps * wl.Lbits * 64 + total * 64,
                                                 total * 4 + total * 8, widths, stored_traffic(key, "locate_3b"))


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
        "note": "build_ms includes the driver's hipMalloc of ~137 GB of scratch in five buffers: 0.6-0.9 s on memory some "
                "process has used before; on memory nobody has touched since boot every hipMalloc that follows a first "
                "touch costs ~28 ms per GiB touched (3-5 s here), and so does re-allocating what this process has freed "
                "(DESIGN.md section 4.3; `pretouch` = the child process that wrote the free memory once before this run)"}
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
    if wr.level is not None:
        locate_leg(out, wr, args, 1, 0, None, False, key, dest=o)
    if not args.no_cpu_baseline:
        wr.count()
        torch.cuda.synchronize()
        o["cpu_baseline"] = cpu_baseline(wr, args, "rlfm")
    return wr


if __name__ == "__main__":
    main()
