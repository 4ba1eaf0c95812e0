import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
import fm_index_amd as F
from fm_index_amd import workload as W
from oracle import fm_oracle as O
for n, per in ((4097, 4), (50001, 1000), (200003, 7), (1 << 18, 1 << 16)):
    base = (W.splitmix64_np(5, 0, per) % np.uint64(4)).astype(np.uint8) + 1
    t = np.tile(base, n // per + 1)[:n].copy()
    t[-1] = 0
    gi = F.FMIndexWithLocate(F.Text.with_max_character(t, 4), 2, keep_sa=True, force_wide=True)
    oi = O.OracleIndex(t, 4, level=2)
    rows = np.arange(n, dtype=np.uint64)
    print(n, per, "verify", gi.verify_sa(), "sa equal", bool((gi.get_sa(rows) == oi.get_sa(rows)).all()))
    gi.close()
