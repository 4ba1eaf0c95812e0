#!/usr/bin/env python3
"""Locate on batches of MIXED interval lengths: the default DNA index at n = 2^30, level 2.

  pure_singletons   S one-row intervals at random rows
  pure_long         P intervals of H consecutive rows
  mixed             both in one batch, shuffled        -> should cost about pure_singletons + pure_long
  config3 / config3b  the bench's own shapes (2^20 singleton-ish hits; 65 536 short patterns, 2.9e8 hits)

    python3 benchmarks/gpu/locate_mix.py [--log2n 30] [--singles 1000000] [--longs 1000] [--long-len 100000] [--kind dna|rlfm]

With FMX_LIB=.../libfmx_measure.so the environment selects the path: FMX_VARIANT=28 = the round-4 pair of launches chosen
by the batch average; FMX_ADJ_CLUSTERS=0 / 65 = every ticket through the cooperative walk / a lane per hit.
Every batch's positions are checked on the device: text[pos .. pos+k) must equal text[SA-walk start] for the pattern
batches, and all paths must agree with the first one measured (hash of the positions)."""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log2n", type=int, default=30)
    ap.add_argument("--singles", type=int, default=1000000)
    ap.add_argument("--longs", type=int, default=1000)
    ap.add_argument("--long-len", type=int, default=100000)
    ap.add_argument("--kind", default="dna")
    ap.add_argument("--reps", type=int, default=10)
    a = ap.parse_args()
    import torch
    import fm_index_amd as F
    from fm_index_amd import workload as W
    from fm_index_amd import _lib as L
    lib = L.lib()
    dev = torch.device("cuda", 0)
    n = 1 << a.log2n
    if a.kind == "dna":
        text = W.dna_text_torch(n, 1, dev)
        idx = F.FMIndexWithLocate.from_device_text(text.data_ptr(), n, 4, level=2)
    elif a.kind == "dna-row":           # the reference's own row-order sampling: a rows array + the cooperative row-order walk
        text = W.dna_text_torch(n, 1, dev)
        idx = F.FMIndexWithLocate.from_device_text(text.data_ptr(), n, 4, level=2, sampling="row")
    elif a.kind == "rlfm-random":       # config 4's text (about one run per row) with FMX_FLAG_RUN_TABLE
        text = W.byte_text_torch(n, 4, dev)
        idx = F.RLFMIndexWithLocate.from_device_text(text.data_ptr(), n, 255, level=2, run_table=True)
    else:
        text = W.repetitive_text_torch(n, 5, dev, base_len=1 << 20, mut_per_1024=10)
        idx = F.RLFMIndexWithLocate.from_device_text(text.data_ptr(), n, 255, level=2)
    h = idx.handle()
    st = torch.cuda.current_stream()
    sp = C.c_void_p(st.cuda_stream)
    g = torch.Generator(device=dev)
    g.manual_seed(5)

    def batch(singles, longs, long_len):
        s1 = torch.randint(0, n, (singles,), device=dev, generator=g, dtype=torch.int64)
        s2 = torch.randint(0, max(n - long_len, 1), (longs,), device=dev, generator=g, dtype=torch.int64)
        s = torch.cat([s1, s2])
        e = torch.cat([s1 + 1, s2 + long_len])
        p = torch.randperm(len(s), device=dev, generator=g)
        return s[p].contiguous(), e[p].contiguous()

    def measure(name, s, e, out):
        npat = len(s)
        off = torch.zeros(npat + 1, dtype=torch.int64, device=dev)
        off[1:] = torch.cumsum(e - s, 0)
        total = int(off[-1].item())
        pos = torch.empty(max(total, 1), dtype=torch.int64, device=dev)

        def call():
            rc = lib.fmx_locate_batch_dev(h, C.c_void_p(s.data_ptr()), C.c_void_p(e.data_ptr()), npat,
                                          C.c_void_p(off.data_ptr()), total, C.c_void_p(pos.data_ptr()), sp)
            assert rc == 0, lib.fmx_last_error().decode()
        call()
        torch.cuda.synchronize()
        assert lib.fmx_stream_status(h) == 0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(a.reps):
            call()
        e1.record(st)
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / a.reps
        # a cheap order-sensitive hash of the positions (all paths must agree)
        w = (torch.arange(total, device=dev, dtype=torch.int64) * 2654435761 + 12345) & 0xFFFFFFFF
        hsh = int(((pos[:total] * w) & 0x7FFFFFFFFFFF).sum().item())
        out[name] = {"patterns": npat, "hits": total, "ms": round(ms, 4), "hits_per_s": total / (ms / 1e3), "hash": hsh}
        return ms

    out = {"kind": a.kind, "n": n, "lib": os.environ.get("FMX_LIB", "libfmx.so"), "FMX_VARIANT": os.environ.get("FMX_VARIANT"),
           "FMX_ADJ_CLUSTERS": os.environ.get("FMX_ADJ_CLUSTERS"), "walk_records": bool(idx.walk_records())}
    s, e = batch(a.singles, 0, 0)
    t1 = measure("pure_singletons", s, e, out)
    s, e = batch(0, a.longs, a.long_len)
    t2 = measure("pure_long", s, e, out)
    s, e = batch(a.singles, a.longs, a.long_len)
    t3 = measure("mixed", s, e, out)
    out["mixed_over_sum_of_pure"] = round(t3 / (t1 + t2), 4)
    # the mirror case: few long intervals among very many singletons (batch average < 64 hits per pattern)
    s, e = batch(a.singles * 4, max(a.longs // 10, 1), a.long_len)
    measure("mixed_low_average", s, e, out)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
