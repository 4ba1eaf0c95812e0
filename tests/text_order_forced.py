"""Text-order sampling on ONE-level indexes (opt-in FMX_FLAG_TEXT_ORDER; the builder's default there is row
order): the text-order branch of the one-level DNA walk kernel (fmx_locate_f3q_kernel<Q, true>) and of the
generic walk at one level against the oracle, and FMX_FLAG_ROW_ORDER on the index kinds whose default is
text order.  Imported by tests/test_gpu_text_order.py; `python tests/text_order_forced.py` prints "OK <cases>"."""
import sys

import numpy as np

import fm_index_amd as F
from fm_index_amd import workload as W
from oracle import fm_oracle as O


def main():
    cases = 0
    for level in (1, 2, 3, 4):
        for kind, n in (("fm", 5000 + level), ("fm", (1 << 17) + 3 * level), ("multi", 7000 + level)):
            t = (W.splitmix64_np(40 + level, 0, n) % np.uint64(4)).astype(np.uint8) + 1
            t[-1] = 0
            if kind == "multi":
                t[np.arange(101, n - 1, 307)] = 0
            cls = F.FMIndexWithLocate if kind == "fm" else F.FMIndexMultiPiecesWithLocate
            gi = cls(F.Text.with_max_character(t, 4), level, sampling="text")
            assert gi.text_order()
            oi = O.OracleIndex(t, 4, level=level, kind=kind)
            rows = np.arange(n)
            want = oi.get_sa(rows).astype(np.uint64)
            lib = gi._lib
            lib.fmx_set_timing(gi.handle(), 1)
            _, pos = gi.locate_many(np.array([0], np.uint64), np.array([n], np.uint64))   # >= 2^16 hits: 4 walks per group
            steps = int(lib.fmx_last_steps(gi.handle()))
            assert (pos == want).all(), (kind, n, level)
            assert steps == int((want & np.uint64((1 << level) - 1)).sum()), (kind, n, level, steps)
            assert (gi.get_sa(rows[:3000]) == want[:3000]).all()
            # few hits: one walk per group
            _, pos = gi.locate_many(np.array([7, 900], np.uint64), np.array([19, 1000], np.uint64))
            assert (pos == np.concatenate([want[7:19], want[900:1000]])).all()
            assert (gi.export_sa_samples() == want[::1 << level]).all()
            cases += 1
    print("OK", cases)
    return cases


def row_order_forced():
    """FMX_FLAG_ROW_ORDER on a run-length index and on a two-level FM index (default: text order)"""
    for kind, cls, maxc in (("rlfm", F.RLFMIndexWithLocate, 255), ("fm", F.FMIndexWithLocate, 255)):
        n = 40000
        t = (W.splitmix64_np(77, 0, n) % np.uint64(200)).astype(np.uint8) + 1
        t[-1] = 0
        oi = O.OracleIndex(t, maxc, level=2, kind=kind)
        want = oi.get_sa(np.arange(n)).astype(np.uint64)
        for sampling, expect_text in (("row", False), (None, True), ("text", True)):
            gi = cls(F.Text(t), 2, sampling=sampling)
            assert gi.text_order() == expect_text, (kind, sampling)
            _, pos = gi.locate_many(np.array([0], np.uint64), np.array([n], np.uint64))
            assert (pos == want).all(), (kind, sampling)
            assert (gi.export_sa_samples() == want[::4]).all()
            gi.close()
    return True


if __name__ == "__main__":
    main()
    row_order_forced()
    sys.exit(0)
