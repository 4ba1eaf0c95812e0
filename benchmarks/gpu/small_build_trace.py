#!/usr/bin/env python3
"""200 small builds (benches/construction.rs shape: binary text, n given) for an API / kernel trace:
   rocprofv3 --hip-trace --kernel-trace --stats -d DIR -- python3 benchmarks/gpu/small_build_trace.py 1000 fm"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402
import fm_index_amd as F  # noqa: E402
from fm_index_amd import workload as W  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
cls = F.RLFMIndex if len(sys.argv) > 2 and sys.argv[2] == "rlfm" else F.FMIndex
r = W.splitmix64_np(0, 0, n).astype(np.float64) / 2.0 ** 64
t = np.concatenate([np.where(r < 0.5, ord("0"), ord("1")).astype(np.uint8), np.zeros(1, dtype=np.uint8)])
tx = F.Text.with_max_character(t, ord("1"))
for _ in range(5):
    cls(tx).close()
t0 = time.perf_counter()
for _ in range(200):
    cls(tx).close()
print("n=%d %s: %.1f us per build" % (n, cls.__name__, (time.perf_counter() - t0) / 200 * 1e6))
