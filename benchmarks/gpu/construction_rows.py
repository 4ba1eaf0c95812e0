#!/usr/bin/env python3
"""the construction rows of benchmarks/criterion_shapes.py only (benches/construction.rs: one ::new(&text) from a
host text, warm process), several repetitions: what small builds cost"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402
import fm_index_amd as F  # noqa: E402
from fm_index_amd import workload as W  # noqa: E402


def text(prob, seed=0, n=50000):
    r = W.splitmix64_np(seed, 0, n).astype(np.float64) / 2.0 ** 64
    t = np.where(r < prob, ord("0"), ord("1")).astype(np.uint8)
    return np.concatenate([t, np.zeros(1, dtype=np.uint8)])


F.FMIndex(F.Text.with_max_character(text(0.5, n=1000), ord("1"))).close()
for n in (1000, 10000, 100000, 1000000):
    tx = F.Text.with_max_character(text(0.5, n=n), ord("1"))
    for name, cls in (("FMIndex", F.FMIndex), ("RLFMIndex", F.RLFMIndex)):
        cls(tx).close()
        best = 1e9
        for rep in range(3):
            k = 20
            t0 = time.perf_counter()
            for _ in range(k):
                cls(tx).close()
            best = min(best, (time.perf_counter() - t0) / k * 1e6)
        print(json.dumps({"bench": "construction", "index": name, "n": n, "gpu_us_per_build": round(best, 1)}))
