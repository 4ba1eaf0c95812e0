#!/usr/bin/env python3
"""The reference's own criterion workloads (benches/count.rs, benches/locate.rs) on the GPU path.

Text = 50 000 random '0'/'1' bytes (P('0') = prob) + \\0 with max_character b'1' (L = 6:
benches/common.rs:5-15); patterns = all 256 binary strings of length 8 (common.rs:18-27);
one criterion "iteration" = all 256 patterns.  Prints one JSON line per (bench, index, param)
with the time of ONE batch of 256 patterns (HIP-event kernel time and host wall time) next to
the number the reference publishes for the same row (CHANGES.md:45-88, unknown CPU).
These shapes are launch-latency bound on a GPU (2048 pattern symbols per batch); they are the
like-for-like row, not the headline.
"""
import ctypes as C
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

PUBLISHED_US = {  # CHANGES.md v0.2.0 table: default build / target-cpu=native
    ("count", "FMIndex", 0.5): (116.1, 86.0), ("count", "FMIndex", 0.05): (118.9, 75.6),
    ("count", "FMIndex", 0.005): (67.6, 47.1), ("count", "RLFMIndex", 0.5): (494.0, 252.8),
    ("count", "RLFMIndex", 0.05): (426.4, 203.8), ("count", "RLFMIndex", 0.005): (232.5, 115.3),
    ("locate", "FMIndex", 1): (3200.0, 2700.0), ("locate", "FMIndex", 2): (8300.0, 7100.0),
    ("locate", "FMIndex", 3): (18200.0, 15600.0), ("locate", "RLFMIndex", 1): (8900.0, 5200.0),
    ("locate", "RLFMIndex", 2): (24900.0, 14100.0), ("locate", "RLFMIndex", 3): (57700.0, 31900.0),
    # benches/construction.rs (CHANGES.md:42-49 / 69-76): time of one ::new(&text), default / native
    ("construction", "FMIndex", 1000): (45.4, 44.2), ("construction", "FMIndex", 10000): (647.8, 635.5),
    ("construction", "FMIndex", 100000): (7800.0, 7100.0), ("construction", "FMIndex", 1000000): (101800.0, 76700.0),
    ("construction", "RLFMIndex", 1000): (52.7, 48.2), ("construction", "RLFMIndex", 10000): (686.2, 680.5),
    ("construction", "RLFMIndex", 100000): (8000.0, 7700.0), ("construction", "RLFMIndex", 1000000): (101800.0, 93700.0),
}


def main():
    import numpy as np
    import torch
    import fm_index_amd as F
    from fm_index_amd import _lib as L
    from fm_index_amd import workload as W
    lib = L.lib()
    dev = torch.device("cuda", 0)
    pats = [format(k, "08b").encode() for k in range(256)]
    flat, off = F.pack_patterns(pats)
    d_pat = torch.from_numpy(flat).to(dev)
    d_off = torch.from_numpy(off.astype(np.int64)).to(dev)
    d_s = torch.empty(256, dtype=torch.int64, device=dev)
    d_e = torch.empty(256, dtype=torch.int64, device=dev)
    d_o = torch.empty(257, dtype=torch.int64, device=dev)
    reps = 200

    def text(prob, seed=0, n=50000):
        r = W.splitmix64_np(seed, 0, n).astype(np.float64) / 2.0 ** 64
        t = np.where(r < prob, ord("0"), ord("1")).astype(np.uint8)
        return np.concatenate([t, np.zeros(1, dtype=np.uint8)])

    # benches/construction.rs: one ::new(&text) from a HOST text (H2D copy included), warm process
    F.FMIndex(F.Text.with_max_character(text(0.5, n=1000), ord("1"))).close()
    for n in (1000, 10000, 100000, 1000000):
        tx = F.Text.with_max_character(text(0.5, n=n), ord("1"))
        for name, cls in (("FMIndex", F.FMIndex), ("RLFMIndex", F.RLFMIndex)):
            cls(tx).close()
            k = 20
            t0 = time.perf_counter()
            for _ in range(k):
                cls(tx).close()
            us = (time.perf_counter() - t0) / k * 1e6
            print(json.dumps({"bench": "construction", "index": name, "n": n, "gpu_us_per_build": round(us, 1),
                              "reference_published_us": PUBLISHED_US[("construction", name, n)]}))

    def time_batches(fn):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / reps, (time.perf_counter() - t0) * 1e6 / reps

    for prob in (0.5, 0.05, 0.005):
        for name, cls in (("FMIndex", F.FMIndex), ("RLFMIndex", F.RLFMIndex)):
            idx = cls(F.Text.with_max_character(text(prob), ord("1")))
            h = idx.handle()

            def count():
                assert lib.fmx_count_batch_dev(h, C.c_void_p(d_pat.data_ptr()), C.c_void_p(d_off.data_ptr()),
                                               256, None, C.c_void_p(d_s.data_ptr()),
                                               C.c_void_p(d_e.data_ptr()), None, None) == 0
            dev_us, wall_us = time_batches(count)
            pub = PUBLISHED_US[("count", name, prob)]
            print(json.dumps({"bench": "count", "index": name, "prob": prob, "gpu_us_per_256": round(dev_us, 2),
                              "gpu_wall_us_per_256": round(wall_us, 2), "reference_published_us": pub,
                              "pattern_chars_per_s": round(2048 / (dev_us * 1e-6))}))
    for level in (1, 2, 3):
        for name, cls in (("FMIndex", F.FMIndexWithLocate), ("RLFMIndex", F.RLFMIndexWithLocate)):
            idx = cls(F.Text.with_max_character(text(0.5), ord("1")), level)
            h = idx.handle()
            b = idx.search_many(flat=flat, off=off)
            d_s.copy_(torch.from_numpy(b.s.astype(np.int64)))
            d_e.copy_(torch.from_numpy(b.e.astype(np.int64)))
            total = int(b.counts.sum())
            assert total == 50000 - 7
            d_p = torch.empty(total, dtype=torch.int64, device=dev)

            def locate():   # search + offsets + locate, like the reference's timed closure
                lib.fmx_count_batch_dev(h, C.c_void_p(d_pat.data_ptr()), C.c_void_p(d_off.data_ptr()), 256,
                                        None, C.c_void_p(d_s.data_ptr()), C.c_void_p(d_e.data_ptr()), None, None)
                lib.fmx_offsets_dev(h, C.c_void_p(d_s.data_ptr()), C.c_void_p(d_e.data_ptr()), 256,
                                    C.c_void_p(d_o.data_ptr()), None)
                assert lib.fmx_locate_batch_dev(h, C.c_void_p(d_s.data_ptr()), C.c_void_p(d_e.data_ptr()),
                                                256, C.c_void_p(d_o.data_ptr()), total,
                                                C.c_void_p(d_p.data_ptr()), None) == 0
            dev_us, wall_us = time_batches(locate)
            pub = PUBLISHED_US[("locate", name, level)]
            print(json.dumps({"bench": "locate", "index": name, "level": level, "hits": total,
                              "gpu_us_per_256": round(dev_us, 2), "gpu_wall_us_per_256": round(wall_us, 2),
                              "reference_published_us": pub, "hits_per_s": round(total / (dev_us * 1e-6))}))


if __name__ == "__main__":
    main()
