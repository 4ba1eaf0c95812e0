#!/usr/bin/env python3
"""Writes tests/golden/config5_counts.json: the sha256 of the per-pattern counts of BASELINE config 5's global
pattern set (n = 2^30 sigma=4 DNA text seed 1; 8 388 608 length-32 substring patterns, seed 7),
computed by the CPU ORACLE over ALL patterns -- not by the HIP path.  Runs on a GPU box (the 2^30 suffix array is
built by the GPU builder and exported as the BWT the oracle is built from; tests/test_gpu_fullsize.py checks that
suffix array against the text on the host); ~1 minute.  bench.py compares its gathered counts with this hash at
every G (`counts_sha256_matches_golden`).

    python tests/golden/make_config5_golden.py            # n = 2^30, the committed entry
    python tests/golden/make_config5_golden.py --log2n 20 --total 65536    # a small extra entry
"""
import argparse
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log2n", type=int, default=30)
    ap.add_argument("--total", type=int, default=8 << 20)
    ap.add_argument("--plen", type=int, default=32)
    ap.add_argument("--seed", type=int, default=7)
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--locate-level", type=int, default=2,
                    help="also hash the ORDERED locate positions of every pattern (iter_matches().map(locate), "
                         "wrapper.rs:203-242) at this sampling level; -1 = counts only")
    a = ap.parse_args()
    import torch
    import bench
    import fm_index_amd as F
    from fm_index_amd import workload as W
    from oracle import fm_oracle as O
    dev = torch.device("cuda", 0)
    n, T, m = 1 << a.log2n, a.total, a.plen
    text = W.dna_text_torch(n, 1, dev)
    lvl = a.locate_level if a.locate_level >= 0 else None
    index = (F.FMIndexWithLocate.from_device_text(text.data_ptr(), n, 4, level=lvl, device=0) if lvl is not None else
             F.FMIndex.from_device_text(text.data_ptr(), n, 4, device=0))
    t0 = time.time()
    # (export_sa_samples yields the reference's samples SA[k << level] whatever the index samples inside)
    oi = O.OracleIndex.from_bwt(index.export_bwt(), index.export_cs(), 4, native=True, kind="fm",
                                samples=index.export_sa_samples() if lvl is not None else None, level=lvl)
    threads = a.threads or bench.host_cpu()["effective_cpus"]
    h, hr, hp = hashlib.sha256(), hashlib.sha256(), hashlib.sha256()
    hits = 0
    total_count = 0
    ar = torch.arange(m, dtype=torch.int64, device=dev)[None, :]
    chunk = 1 << 20
    for lo in range(0, T, chunk):
        k = min(chunk, T - lo)
        src = W.umod_torch(W.splitmix64_torch(a.seed, lo, k, dev), n - 1 - m)
        ph = text[src[:, None] + ar].reshape(-1).cpu().numpy()
        so, eo = oi.count_batch(ph, np.arange(k + 1, dtype=np.uint64) * np.uint64(m), nthreads=threads)
        c = (eo - so).astype("<i8")
        assert (c >= 1).all()
        total_count += int(c.sum())
        h.update(c.tobytes())
        se = np.empty((k, 2), dtype="<i8")                 # bench.ranges_sha256: [s_0, e_0, s_1, e_1, ...]
        se[:, 0], se[:, 1] = so.astype("<i8"), eo.astype("<i8")
        hr.update(se.tobytes())
        if lvl is not None:                                # the ordered positions, pattern after pattern
            _, pos = oi.locate_batch(so, eo, nthreads=threads)
            hp.update(pos.astype("<i8").tobytes())
            hits += len(pos)
        print("patterns %d..%d done (%.0f s)" % (lo, lo + k, time.time() - t0), file=sys.stderr)
    path = os.path.join(ROOT, "tests", "golden", "config5_counts.json")
    try:
        g = json.load(open(path))
    except (OSError, ValueError):
        g = {"what": "counts_sha256 = sha256 over the int64 little-endian per-pattern counts (input order), ranges_sha256 = the "
                     "same over the (s, e) pairs [s_0, e_0, s_1, e_1, ...], of a global pattern set: "
                     "substrings text[p : p + len] with p = splitmix64(seed, k) mod (n - 1 - len) of the sigma=4 DNA text "
                     "(seed 1) -- computed by the CPU oracle over ALL patterns (tests/golden/make_config5_golden.py)",
             "locate": "positions_sha256 = sha256 over the int64 little-endian text positions of every pattern's matches in "
                       "the reference's iteration order (ascending suffix-array row within a pattern, wrapper.rs:203-217), "
                       "patterns in input order, from the oracle's get_sa walk over the reference's own samples",
             "entries": {}}
    ent = {"counts_sha256": h.hexdigest(), "ranges_sha256": hr.hexdigest(), "counts_sum": total_count,
           "oracle_threads": threads}
    if lvl is not None:
        ent["locate"] = {"level": lvl, "hits": hits, "positions_sha256": hp.hexdigest()}
    ent["oracle_seconds"] = round(time.time() - t0, 1)
    g["entries"][bench.golden_key("dna", a.log2n, a.seed, T, m)] = ent
    with open(path, "w") as f:
        json.dump(g, f, indent=1, sort_keys=True)
        f.write("\n")
    print(json.dumps(g["entries"]))


if __name__ == "__main__":
    main()
