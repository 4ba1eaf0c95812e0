#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-pointer entry point fmx_count_batch (config 2 workload):
patterns + offsets copied in, (s, e, count) copied out, every call.  Never the headline value
(bench.py times the device-resident path); DESIGN.md section 6 quotes this number."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import numpy as np
    import torch
    import fm_index_amd as F
    from fm_index_amd import workload as W
    n, npat, m = 1 << 30, 1 << 20, 32
    dev = torch.device("cuda", 0)
    text = W.dna_text_torch(n, 1, dev)
    index = F.FMIndex.from_device_text(text.data_ptr(), n, 4)
    pat, off, _ = W.substring_patterns_torch(text, npat, m, 3)
    flat = pat.cpu().numpy()
    offs = off.cpu().numpy().astype(np.uint64)
    index.search_many(flat=flat, off=offs)
    reps = 10
    t0 = time.perf_counter()
    for _ in range(reps):
        b = index.search_many(flat=flat, off=offs)
    dt = (time.perf_counter() - t0) / reps
    assert (b.counts >= 1).all()
    print(json.dumps({"entry_point": "fmx_count_batch (host pointers, pageable memory)",
                      "ms_per_call": round(dt * 1e3, 3), "pattern_chars_per_s": round(npat * m / dt),
                      "bytes_in": int(flat.nbytes + offs.nbytes), "bytes_out": 3 * 8 * npat}))


if __name__ == "__main__":
    main()
