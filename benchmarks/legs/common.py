"""What every leg of bench.py shares: the workload object (text + index + pattern batch of one BASELINE config,
resident in HBM), event timing, the hashes of config 5 and their committed golden values."""
import ctypes as C
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured streaming)
GATHER_CEILING_GLINES = 55.0  # profiles/microbench/gather_r02.txt: dependent random lines the memory
#                               system sustains (16..128-byte requests alike, 128 MiB..2 GiB tables)
LINE = 128


class Workload:
    """text + index + pattern batch of one BASELINE config, resident in HBM."""

    def __init__(self, name, args, dev, local, rank, world, rlfm=None, with_locate=True, npat=None, plen=None,
                 index_kw=None, text=None):
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
        if text is not None:
            self.text = text                         # another workload's text (same name, same n)
        elif self.dna:
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
        self.index = cls.from_device_text(self.text.data_ptr(), self.n, self.maxc, level=self.level, device=local,
                                          **(index_kw or {}))
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

    def accelerated(self):
        """does the index run the accelerated count kernel (pair index + k-mer start table: the default DNA index of
        2^24+ symbols since round 6)?"""
        return bool(self.dna and self.index.has_pair_index() and self.index.kmer_k())

    def count_table_bytes(self):
        """bytes of index the count kernel's random record reads spread over: DNA = one 128-byte
        fmt-3 record per 256 rows -- on the accelerated index one fmt-4 pair record per 128 rows (the plain records serve
        a pattern's odd last step only) + the k-mer table; other kinds: not modelled (None)"""
        if not self.dna:
            return None
        if self.accelerated():
            return (self.n // 128 + 1) * 128 + (8 << (2 * self.index.kmer_k()))
        return (self.n // 256 + 1) * 128

    def locate_table_bytes(self):
        """... and the walk kernel's: 112-row walk records + the u32 samples (DNA index with walk records)"""
        if not (self.dna and self.level is not None and self.index.walk_records()):
            return None
        return (self.n // 112 + 1) * 128 + 4 * (((self.n - 1) >> self.level) + 1)

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

