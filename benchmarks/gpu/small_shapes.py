"""Where do the endpoint-per-lane kernels pay?  RLFM count / locate and two-level FM locate on small
and mid-size texts and batches, shipped dispatch vs the round-1 kernels (FMX_VARIANT=0, measurement
build).  One JSON line per cell: kernel-event microseconds per batch.
    FMX_LIB=.../libfmx_measure.so [FMX_VARIANT=0] python benchmarks/gpu/small_shapes.py"""
import ctypes as C, json, os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import fm_index_amd as F
from fm_index_amd import workload as W, _lib as L
lib = L.lib()
dev = torch.device("cuda", 0)
tag = "v" + os.environ.get("FMX_VARIANT", "1") + ("-" + os.environ["SAMPLING"] if os.environ.get("SAMPLING") else "")
only = sys.argv[1:]            # e.g. `dna`: only that index
for log2n in (16, 20, 24, 27):
    n = 1 << log2n
    for name, cls in (("rlfm", F.RLFMIndexWithLocate), ("fm2", F.FMIndexWithLocate), ("dna", F.FMIndexWithLocate)):
        if only and name not in only:
            continue
        text = W.dna_text_torch(n, 1, dev) if name == "dna" else W.byte_text_torch(n, 4, dev)
        kw = {"sampling": os.environ["SAMPLING"]} if os.environ.get("SAMPLING") else {}     # "row" / "text": override the default
        idx = cls.from_device_text(text.data_ptr(), n, 4 if name == "dna" else 255, level=2, device=0, **kw)
        h = idx.handle()
        for log2p in (8, 12, 16, 20):
            npat, m = 1 << log2p, 8 if log2n <= 20 else 12
            pat, off, _ = W.substring_patterns_torch(text, npat, m, 3)
            s = torch.empty(npat, dtype=torch.int64, device=dev); e = torch.empty_like(s); o = torch.empty(npat + 1, dtype=torch.int64, device=dev)
            def count():
                assert lib.fmx_count_batch_dev(h, C.c_void_p(pat.data_ptr()), C.c_void_p(off.data_ptr()), npat, None, C.c_void_p(s.data_ptr()), C.c_void_p(e.data_ptr()), None, None) == 0
            count(); torch.cuda.synchronize()
            lib.fmx_offsets_dev(h, C.c_void_p(s.data_ptr()), C.c_void_p(e.data_ptr()), npat, C.c_void_p(o.data_ptr()), None)
            total = int(o[-1].item())
            p = torch.empty(max(total, 1), dtype=torch.int64, device=dev)
            def locate():
                assert lib.fmx_locate_batch_dev(h, C.c_void_p(s.data_ptr()), C.c_void_p(e.data_ptr()), npat, C.c_void_p(o.data_ptr()), total, C.c_void_p(p.data_ptr()), None) == 0
            def timeit(fn, reps=30):
                fn(); torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(reps): fn()
                e1.record(); torch.cuda.synchronize()
                return e0.elapsed_time(e1) * 1e3 / reps
            print(json.dumps({"tag": tag, "index": name, "log2n": log2n, "log2npat": log2p, "hits": total,
                              "count_us": round(timeit(count), 1), "locate_us": round(timeit(locate), 1)}))
        idx.close()
