"""BASELINE.json configs[0] -- the CPU-runnable plumbing case: FMIndex over a 1 MB sigma=4
synthetic text, 10 000 uniform random length-20 patterns -- on the CPU oracle (no GPU):
counts against a brute-force scan, and the executed-step census SURVEY section 8d quotes."""
import numpy as np

from fm_index_amd import workload as W
from oracle import fm_oracle as O


def test_config1_oracle_counts_and_step_census():
    n = 1 << 20
    t = W.dna_text_np(n, 1)
    fm = O.OracleIndex(t, 4)
    flat, off = W.random_patterns_np(10000, 20, 4, 2)
    s, e, steps = fm.count_batch(flat, off, nthreads=8, want_steps=True)
    cnt = (e - s).astype(np.int64)
    # uniform random patterns die after ~log4(n)+1 steps (~11 of 20)
    assert 10.0 < steps.mean() < 13.0
    assert (cnt[steps < 20] == 0).all()
    tb = t.tobytes()
    for k in range(0, 10000, 40):                       # brute force on a 250-pattern sample
        p = flat[int(off[k]):int(off[k + 1])].tobytes()
        c, pos = 0, tb.find(p)
        while pos != -1:
            c += 1
            pos = tb.find(p, pos + 1)
        assert c == cnt[k]
    # substrings of the text always survive all 20 steps
    flat2, off2, _ = W.substring_patterns_np(t, 2000, 20, 3)
    s2, e2, st2 = fm.count_batch(flat2, off2, nthreads=8, want_steps=True)
    assert (st2 == 20).all() and ((e2 - s2) >= 1).all()
