#!/usr/bin/env python3
"""config 3 (2^20 length-32 substring patterns, ~2^20 hits) and config 3b (65 536 patterns of length 8-12, 2.9e8 hits)
on three indexes over the same n = 2^30 DNA text at level 2: row order (the reference's sampling), text order without
walk records (round 3's phase-probe walk), text order with walk records.  Positions asserted identical.  One JSON
line per (index, shape)."""
import ctypes as C
import json
import sys
import os
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
import fm_index_amd as F
from fm_index_amd import _lib as L


def main():
    log2n = int(os.environ.get("LOG2N", "30"))
    args = bench.parse_args(["--log2n", str(log2n)])
    dev = torch.device("cuda", 0)
    wl = bench.Workload("dna", args, dev, 0, 0, 1)
    lib = wl.lib
    wl.count()
    wl.prepare_locate()
    wl.locate()
    torch.cuda.synchronize()
    ref = wl.d_pos[:wl.total_hits].clone()
    npat3, pat3, off3, s3, e3, total3, pos3, lstep3 = bench.setup_3b(wl)
    lstep3()
    torch.cuda.synchronize()
    ref3 = pos3[:total3].clone()
    variants = [("row", dict(sampling="row")), ("text_no_walk", dict(sampling="text", walk_records=False)),
                ("text_walk", dict(sampling="text"))]
    only = os.environ.get("ONLY")
    for name, kw in variants:
        if only and name not in only.split(","):
            continue
        ix = F.FMIndexWithLocate.from_device_text(wl.text.data_ptr(), wl.n, 4, level=args.level, device=0, **kw)
        h = ix.handle()
        out = torch.empty(max(wl.total_hits, total3), dtype=torch.int64, device=dev)
        shapes = {"config3": (wl.d_s, wl.d_e, wl.d_off, wl.npat, wl.total_hits, ref),
                  "config3b": (s3, e3, None, npat3, total3, ref3)}
        if os.environ.get("SHAPES"):
            shapes = {k: v for k, v in shapes.items() if k in os.environ["SHAPES"].split(",")}
        for shape, (s, e, off, npat, total, want) in shapes.items():
            if off is None:
                off = torch.empty(npat + 1, dtype=torch.int64, device=dev)
                lib.fmx_offsets_dev(h, C.c_void_p(s.data_ptr()), C.c_void_p(e.data_ptr()), npat, C.c_void_p(off.data_ptr()), wl.sp)

            def step():
                rc = lib.fmx_locate_batch_dev(h, C.c_void_p(s.data_ptr()), C.c_void_p(e.data_ptr()), npat,
                                              C.c_void_p(off.data_ptr()), total, C.c_void_p(out.data_ptr()), wl.sp)
                assert rc == 0, lib.fmx_last_error().decode()
            for _ in range(3):
                step()
            torch.cuda.synchronize()
            reps = 20 if shape == "config3" else 3
            t0 = time.perf_counter()
            for _ in range(reps):
                step()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / reps
            lib.fmx_set_timing(h, 1)
            step()
            torch.cuda.synchronize()
            kms, steps = lib.fmx_last_kernel_ms(h), int(lib.fmx_last_steps(h))
            lib.fmx_set_timing(h, 0)
            assert bool((out[:total] == want).all()), (name, shape)
            print(json.dumps({"index": name, "shape": shape, "hits": total, "ms_per_batch": round(dt * 1e3, 4),
                              "hits_per_s": total / dt, "walk_kernel_ms": round(kms, 4), "lf_steps": steps,
                              "index_bytes": ix.heap_size(), "walk_records": ix.walk_records(),
                              "build_ms": round(float(lib.fmx_build_ms(h)), 1)}), flush=True)
        ix.close()


if __name__ == "__main__":
    main()
