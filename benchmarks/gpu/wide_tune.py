"""Times the wide engine's count / locate kernels on the config-2 / config-3 shapes (2^20 substring patterns of
length 32; 2^20 hits at level 2).  One process = one index.  The round-3 sweep (profiles/r03/wide_tune.jsonl) ran it
over temporary builds whose kernel shape came from the environment (FMXW_COUNT_VARIANT: patterns per group / one
load for both ends of a narrow interval; FMXW_WALKS: walks per group; FMXW_SB_SHIFT=31: the wide engine on a
2^30 text next to the 32-bit one); the shipped kernels are the winners and read no environment.

    python benchmarks/gpu/wide_tune.py LOG2N [extra]      n = 2^LOG2N + extra; n < 2^32 - 16 is built with force_wide
"""
import ctypes as C
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch  # noqa: E402
import fm_index_amd as F  # noqa: E402
from fm_index_amd import _lib as L  # noqa: E402
from fm_index_amd import workload as W  # noqa: E402


def main():
    log2n = int(sys.argv[1])
    extra = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    narrow = os.environ.get("NARROW") == "1"
    n = (1 << log2n) + extra
    dev = torch.device("cuda", 0)
    lib = L.lib()
    text = W.dna_text_torch(n, 17, dev)
    t0 = time.time()
    index = F.FMIndexWithLocate.from_device_text(text.data_ptr(), n, 4, level=2, force_wide=not narrow)
    build_s = time.time() - t0
    h = index.handle()
    kp, mp = 1 << 20, 32
    out = {"n": n, "wide": bool(index.is_wide()), "build_s": round(build_s, 2),
           "env": {k: v for k, v in os.environ.items() if k.startswith("FMXW_")}}
    for label, seed in (("count", 3), ("locate", 5)):
        m = mp if label == "count" else 22          # length-22 substrings of a 2^32 DNA text: ~1 hit each
        src = W.umod_torch(W.splitmix64_torch(seed, 0, kp, dev), n - 1 - m)
        pat = text[src[:, None] + torch.arange(m, dtype=torch.int64, device=dev)[None, :]].reshape(-1).contiguous()
        off = (torch.arange(kp + 1, dtype=torch.int64, device=dev) * m).contiguous()
        s, e = (torch.empty(kp, dtype=torch.int64, device=dev) for _ in range(2))

        def count():
            assert lib.fmx_count_batch_dev(h, C.c_void_p(pat.data_ptr()), C.c_void_p(off.data_ptr()), kp, None,
                                           C.c_void_p(s.data_ptr()), C.c_void_p(e.data_ptr()), None, None) == 0
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

        def timed(fn, reps=20):
            fn(); fn()
            torch.cuda.synchronize()
            ev0.record()
            for _ in range(reps):
                fn()
            ev1.record()
            torch.cuda.synchronize()
            return ev0.elapsed_time(ev1) / reps
        if label == "count":
            out["count_ms"] = round(timed(count), 4)
            out["count_chars_per_s"] = kp * mp / (out["count_ms"] / 1e3)
            out["count_checksum"] = int((s.sum() * 3 + e.sum()).item() & ((1 << 62) - 1))
        else:
            count()
            offh = torch.empty(kp + 1, dtype=torch.int64, device=dev)
            lib.fmx_offsets_dev(h, C.c_void_p(s.data_ptr()), C.c_void_p(e.data_ptr()), kp, C.c_void_p(offh.data_ptr()), None)
            tot = int(offh[-1].item())
            pos = torch.empty(tot, dtype=torch.int64, device=dev)

            def locate():
                assert lib.fmx_locate_batch_dev(h, C.c_void_p(s.data_ptr()), C.c_void_p(e.data_ptr()), kp,
                                                C.c_void_p(offh.data_ptr()), tot, C.c_void_p(pos.data_ptr()), None) == 0
            out["locate_hits"] = tot
            out["locate_ms"] = round(timed(locate), 4)
            out["locate_hits_per_s"] = tot / (out["locate_ms"] / 1e3)
            out["locate_checksum"] = int(pos.sum().item() & ((1 << 62) - 1))
    print(json.dumps(out), flush=True)


main()
