#!/usr/bin/env python3
"""Differential soak test: random texts / alphabets / index kinds / levels / flags, random and
substring patterns, every result compared with the CPU oracle (bit-exact (s,e), ordered locate
sequences, every trait method on sampled rows).  Usage: python tests/fuzz_gpu_vs_oracle.py [seconds] [seed] [--long
[--iters=N]]  (--long: the long-interval batches of long_intervals() below)
(the oracle is the checker here, exactly as in tests/)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def long_intervals(budget, seed, max_iters):
    """`long` mode (round 5): the kernels that walk LONG intervals -- the one-launch DNA kernel with its
    per-ticket choice (fmx_locate_f3u_kernel: lane per hit / cooperative walk), the RLFM lane-per-walk kernel
    (fmx_locate_rl_rounds_kernel) next to the endpoint-per-lane one -- on batches that cross their dispatch thresholds
    (>= 64 hits per pattern; >= 2^16 DNA / 2^18 RLFM hits), in text and row order, skewed batches (singletons + a few huge
    intervals), and RLFM texts whose run-length mix reaches every branch of fmx_bits_lane_select: stored positions (sparse
    B), select blocks of every size (dense B), and the hint + record search (a long run inside a dense vector, so that
    the run starts before the row's 96-bit piece and its select block cannot hold it).  Ordered positions == oracle."""
    import numpy as np
    import fm_index_amd as F
    from oracle import fm_oracle as O
    rng = np.random.default_rng(seed)
    t_end = time.time() + budget
    it = 0
    stats = {"dna": 0, "rlfm": 0, "hits": 0, "text_order": 0, "run_table": 0, "skewed": 0, "row_order": 0}
    threads = min(16, os.cpu_count() or 1)
    while time.time() < t_end and it < max_iters:
        it += 1
        dna = rng.random() < 0.5
        n = int(rng.choice([66000, 131072, 200003, 300000]))
        if dna:
            alpha = int(rng.choice([2, 3, 4, 5]))
            maxc = int(rng.choice([alpha, 4 if alpha <= 4 else 5, 5]))
            level = int(rng.integers(1, 4))
            style = rng.choice(["random", "repetitive"])
        else:
            alpha = int(rng.choice([2, 4, 20, 255]))
            maxc = 255
            level = int(rng.integers(0, 5))
            style = rng.choice(["runs_short", "runs_long", "runs_mixed", "repetitive", "random"])
        if style == "random":
            t = rng.integers(1, alpha + 1, size=n)
        elif style == "repetitive":
            blk = rng.integers(1, alpha + 1, size=int(rng.choice([7, 64, 300, 2000])))
            t = np.tile(blk, n // len(blk) + 1)[:n]
            mut = rng.random(n) < float(rng.choice([0.0005, 0.004, 0.02]))
            t[mut] = rng.integers(1, alpha + 1, size=int(mut.sum()))
        else:
            # run lengths: short (dense B: select blocks), long (sparse B: stored positions), or short with a few
            # runs of 100..900 symbols (dense B whose select blocks cannot hold those runs)
            k = n
            if style == "runs_short":
                lens = rng.integers(1, int(rng.choice([2, 3, 5, 9])), size=k)
            elif style == "runs_long":
                lens = rng.integers(8, 60, size=k)
            else:
                lens = rng.integers(1, int(rng.choice([2, 4, 8])), size=k)
                big = rng.random(k) < 0.004
                lens[big] = rng.integers(100, 900, size=int(big.sum()))
            t = np.repeat(rng.integers(1, alpha + 1, size=k), lens)[:n]
        t = t.astype(np.uint8)
        t[n - 1] = 0
        sampling = [None, "row"][int(rng.random() < 0.3)]
        oi = O.OracleIndex(t, maxc, level=level, kind="fm" if dna else "rlfm")
        text = F.Text.with_max_character(t, maxc)
        if dna:
            gi = F.FMIndexWithLocate(text, level, sampling=sampling)
        else:
            gi = F.RLFMIndexWithLocate(text, level, sampling=sampling, run_table=bool(rng.random() < 0.85))
        stats["dna" if dna else "rlfm"] += 1
        stats["text_order"] += int(gi.text_order())
        stats["row_order"] += int(not gi.text_order())
        stats["run_table"] += int((not dna) and gi.walk_records())
        # the batch: intervals over the rows (any (s, e) with s <= e <= n is an argument of locate)
        shape = rng.choice(["long", "skewed", "threshold", "adjacent_patterns"])
        if shape == "long":                      # every pattern far above 64 hits
            k = int(rng.integers(40, 400))
            ln = rng.integers(200, 6000, size=k)
        elif shape == "skewed":                  # singletons and short intervals + a few huge ones
            k = int(rng.integers(2000, 30000))
            ln = rng.integers(0, 3, size=k)
            hv = rng.choice(k, int(rng.integers(1, 6)), replace=False)
            ln[hv] = rng.integers(30000, min(n, 250000), size=len(hv))
            stats["skewed"] += 1
        elif shape == "threshold":               # around 64 hits per pattern, around the 2^16 / 2^18 totals
            per = int(rng.choice([60, 63, 64, 65, 70, 128]))
            tot = int(rng.choice([1 << 16, 1 << 18])) + int(rng.integers(-3000, 3000))
            k = max(1, tot // per)
            ln = np.full(k, per)
            ln[rng.integers(0, k, size=max(1, k // 50))] += rng.integers(0, 5, size=max(1, k // 50))
        else:                                    # consecutive patterns cover consecutive rows (adjacent across patterns)
            k = int(rng.integers(500, 5000))
            ln = rng.integers(10, 200, size=k)
        s = rng.integers(0, n, size=k).astype(np.int64)
        if shape == "adjacent_patterns":
            s = (int(rng.integers(0, n // 2)) + np.concatenate([[0], np.cumsum(ln[:-1])])) % n
        e = np.minimum(s + ln, n)
        s, e = s.astype(np.uint64), e.astype(np.uint64)
        goff, gpos = gi.locate_many(s, e)
        ooff, opos = oi.locate_batch(s, e, nthreads=threads)
        assert (np.asarray(goff) == np.asarray(ooff)).all() and (np.asarray(gpos) == np.asarray(opos)).all(), \
            ("long", it, "dna" if dna else "rlfm", n, alpha, maxc, level, style, sampling, shape, seed)
        stats["hits"] += int(ooff[-1])
        gi.close()
        oi.close()
    print("fuzz (long intervals) ok: %d iterations in %.0f s, seed %d, %s" % (it, budget, seed, stats))


def main():
    import numpy as np
    import fm_index_amd as F
    from fm_index_amd import workload as W
    from oracle import fm_oracle as O
    argv = [a for a in sys.argv[1:] if not a.startswith("--")]
    budget = float(argv[0]) if len(argv) > 0 else 60.0
    seed = int(argv[1]) if len(argv) > 1 else 1
    if "--long" in sys.argv[1:]:
        iters = [int(a.split("=")[1]) for a in sys.argv[1:] if a.startswith("--iters=")]
        return long_intervals(budget, seed, iters[0] if iters else 1 << 30)
    rng = np.random.default_rng(seed)
    t_end = time.time() + budget
    it = 0
    stats = {"fm": 0, "rlfm": 0, "multi": 0, "pair": 0, "wide": 0}
    while time.time() < t_end:
        it += 1
        kind = rng.choice(["fm", "fm", "rlfm", "multi"])
        n = int(rng.choice([2, 3, 5, 17, 64, 255, 256, 257, 700, 1023, 1024, 1025, 3000, 9000, 20000, 66000]))
        wide = kind != "multi" and rng.random() < 0.2
        if wide:
            alpha = int(rng.choice([3, 40, 300, 5000]))
            maxc = int(rng.choice([alpha, alpha + 7, 2 * alpha, 65535]))
            dtype = np.uint16 if maxc <= 65535 and rng.random() < 0.5 else np.uint32
        else:
            alpha = int(rng.choice([1, 2, 3, 4, 5, 8, 16, 100, 255]))
            maxc = int(rng.choice([alpha, min(255, alpha + 1), 255]))
            dtype = np.uint8
        style = rng.choice(["random", "repetitive", "runs"])
        if style == "random":
            t = rng.integers(1, alpha + 1, size=n)
        elif style == "repetitive":
            blk = rng.integers(1, alpha + 1, size=max(1, int(rng.integers(1, 40))))
            t = np.tile(blk, n // len(blk) + 1)[:n]
            mut = rng.random(n) < 0.02
            t[mut] = rng.integers(1, alpha + 1, size=int(mut.sum()))
        else:
            t = np.repeat(rng.integers(1, alpha + 1, size=n), rng.integers(1, 9, size=n))[:n]
        t = t.astype(dtype)
        if kind == "multi" and n > 4:
            z = rng.random(n) < 0.05
            z[0] = False
            z[1:] &= ~z[:-1]          # no double zeros
            z[n - 2] = False
            t[z] = 0
        t[n - 1] = 0
        level = None if rng.random() < 0.2 else int(rng.integers(0, 6))
        pair = kind == "fm" and not wide and maxc <= 4 and rng.random() < 0.5
        kmer = not wide and rng.random() < 0.5      # the build ignores it where it would not pay
        t_or = t if dtype == np.uint8 else t.astype(np.uint32)
        try:
            oi = O.OracleIndex(t_or, maxc, level=level, kind=kind)
        except O.OracleError:
            continue
        text = F.Text.with_max_character(t, maxc)
        # which rows carry the samples (FMX_FLAG_TEXT_ORDER / FMX_FLAG_ROW_ORDER / the builder's choice), and for
        # eligible indexes sometimes the 64-bit engine (FMX_FLAG_FORCE_WIDE)
        sampling = [None, "text", "row"][int(rng.integers(0, 3))] if level is not None else None
        engine64 = n >= 2 and rng.random() < (0.4 if (maxc <= 7 and dtype == np.uint8) else 0.25)
        if engine64:                     # (the 64-bit engine: row order or its default -- walk records where eligible)
            pair, kmer, sampling = False, False, (sampling if sampling != "text" else None)
        # the derived locate structures (walk records / run table: FMX_FLAG_NO_WALK_RECORDS) on or off
        walk = bool(rng.random() < 0.7)
        if kind == "fm":
            gi = F.FMIndexWithLocate(text, level, pair_index=pair, kmer_table=kmer, sampling=sampling,
                                     force_wide=engine64, walk_records=walk) if level is not None else \
                F.FMIndex(text, pair_index=pair, kmer_table=kmer, force_wide=engine64)
            assert gi.is_wide() == engine64
        elif kind == "rlfm":
            if n < 2:
                continue
            # (FMX_FLAG_RUN_TABLE: most texts here have about one run per row, where the builder leaves the table out)
            gi = F.RLFMIndexWithLocate(text, level, kmer_table=kmer, sampling=sampling, walk_records=walk,
                                       force_wide=engine64, run_table=bool(rng.random() < 0.7)) \
                if level is not None else F.RLFMIndex(text, kmer_table=kmer, force_wide=engine64)
            assert gi.is_wide() == engine64
        else:
            gi = F.FMIndexMultiPiecesWithLocate(text, level, kmer_table=kmer, sampling=sampling, force_wide=engine64) \
                if level is not None else F.FMIndexMultiPieces(text, kmer_table=kmer, force_wide=engine64)
            assert gi.is_wide() == engine64
        stats["engine64"] = stats.get("engine64", 0) + int(engine64)
        stats["text_order"] = stats.get("text_order", 0) + int(gi.text_order())
        stats["walk_records"] = stats.get("walk_records", 0) + int(gi.walk_records())
        stats[kind] += 1
        stats["pair"] += int(pair)
        stats["kmer"] = stats.get("kmer", 0) + int(gi.kmer_k() > 0)
        stats["wide"] += int(wide)
        # patterns: random ragged + substrings (+ occasional zero symbol)
        npat = 200
        lens = rng.integers(0, 12, size=npat)
        pats = []
        for k in range(npat):
            m = int(lens[k])
            if rng.random() < 0.5 and n > 2:
                a = int(rng.integers(0, n - 1))
                p = t[a:min(n - 1, a + m)].copy()
            else:
                p = rng.integers(0 if rng.random() < 0.1 else 1, alpha + 1, size=m).astype(dtype)
            pats.append(p)
        flat, off = F.pack_patterns(pats, dtype)
        se = None
        if rng.random() < 0.3:
            a = rng.integers(0, n + 1, size=npat)
            b2 = rng.integers(0, n + 1, size=npat)
            se = np.stack([np.minimum(a, b2), np.maximum(a, b2)], axis=1).reshape(-1).astype(np.uint64)
        gb = gi.search_many(flat=flat, off=off, s0e0=se)
        os_, oe = oi.count_batch(flat.astype(np.uint32) if dtype != np.uint8 else flat, off, se)
        assert (gb.s == os_).all() and (gb.e == oe).all(), ("count", it, kind, n, maxc, level, pair, kmer, style)
        if level is not None:
            small = (oe - os_) < 2000
            goff, gpos = gi.locate_many(gb.s[small], gb.e[small])
            ooff, opos = oi.locate_batch(os_[small], oe[small])
            assert (goff == ooff).all() and (gpos == opos).all(), ("locate", it, kind, n, maxc, level)
        # round 6: the same batch over 2..4 replicas of the index through ONE call (fmx_replicate + fmx_*_batch_multi):
        # contiguous ragged shards, results in place -- must be the one-handle results
        if rng.random() < 0.3:
            g = int(rng.integers(2, 5))
            reps = F.Replicas.of(gi, [0] * (g - 1))
            mb = reps.search_many(flat=flat, off=off, s0e0=se)
            assert (mb.s == os_).all() and (mb.e == oe).all(), ("count_multi", it, kind, n, maxc, level, g)
            if level is not None:
                moff, mpos = reps.locate_many(os_[small], oe[small])
                assert (moff == ooff).all() and (mpos == opos).all(), ("locate_multi", it, kind, n, maxc, level, g)
            reps.close(keep_first=True)
            stats["replicas"] = stats.get("replicas", 0) + 1
        rows = rng.integers(0, n, size=min(n, 64)).astype(np.uint64)
        assert (gi.get_l(rows) == oi.get_l(rows)).all(), ("get_l", it, kind)
        assert (gi.lf_map(rows) == oi.lf_map(rows)).all(), ("lf_map", it, kind)
        assert (gi.get_f(rows) == oi.get_f(rows)).all(), ("get_f", it, kind)
        assert (gi.fl_map(rows) == oi.fl_map(rows)).all(), ("fl_map", it, kind)
        if level is not None:
            assert (gi.get_sa(rows) == oi.get_sa(rows)).all(), ("get_sa", it, kind)
        if kind == "multi":
            assert (gi.piece_id(rows) == oi.piece_id(rows)).all(), ("piece_id", it)
        cs = rng.integers(0, maxc + 1, size=64).astype(np.uint64)
        ii = rng.integers(0, n + 1, size=64).astype(np.uint64)
        assert (gi.lf_map2(cs, ii) == oi.lf_map2(cs, ii)).all(), ("lf_map2", it, kind, n, maxc)
        gi.close()
        oi.close()
    print("fuzz ok: %d iterations in %.0f s, seed %d, %s" % (it, budget, seed, stats))


if __name__ == "__main__":
    main()
