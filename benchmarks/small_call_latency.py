#!/usr/bin/env python3
"""Latency of the one-element entry points (the trait-method shims of INTEGRATION.md)."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import numpy as np
    import fm_index_amd as F
    from fm_index_amd import workload as W
    t = W.dna_text_np(1 << 20, 1)
    t0 = time.perf_counter()
    gi = F.FMIndexWithLocate(F.Text.with_max_character(t, 4), 2)
    build = time.perf_counter() - t0
    lib, h = gi._lib, gi._h
    out = {"build_ms_n_2^20": round(build * 1e3, 2)}
    for name, fn in (("fmx_lf_map2", lambda: lib.fmx_lf_map2(h, 2, 12345)),
                     ("fmx_get_sa", lambda: lib.fmx_get_sa(h, 12345)),
                     ("search(1 pattern).count()", lambda: gi.search(bytes([1, 2, 3, 4, 1, 2, 3, 4])).count())):
        fn()
        t0 = time.perf_counter()
        for _ in range(200):
            fn()
        out[name + "_us"] = round((time.perf_counter() - t0) / 200 * 1e6, 1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
