"""RLFMIndexWithLocate on the wide engine at n = 2^32 + 2^20 (or --log2n K: forced wide below 2^32), on the repetitive
byte text of config 4b (1 MiB random block repeated, 1 % point mutations): build time, count of 2^20 substring patterns
of 16 symbols, locate of the hits of the first --locate-patterns of them.  Prints one JSON line.
Kernel times are the library's own events (fmx_set_timing), the steps its device counter."""
import argparse
import ctypes as C
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

import fm_index_amd as F  # noqa: E402
from fm_index_amd import _lib as L  # noqa: E402
from fm_index_amd import workload as W  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log2n", type=int, default=0, help="0: n = 2^32 + 2^20; else n = 2^k with FMX_FLAG_FORCE_WIDE")
    ap.add_argument("--level", type=int, default=3)
    ap.add_argument("--npat", type=int, default=1 << 20)
    ap.add_argument("--m", type=int, default=16)
    ap.add_argument("--locate-patterns", type=int, default=1 << 17)
    ap.add_argument("--no-run-table", action="store_true")
    ap.add_argument("--narrow", action="store_true", help="the 32-bit engine (needs --log2n <= 31), for comparison")
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--sweep-ep-blocks", default="", help="measurement build: count kernel ms per FMXW_EP_BLOCKS value "
                    "(and 'group' = the group-per-pattern kernel)")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    lib = L.lib()
    n = (1 << 32) + (1 << 20) if a.log2n == 0 else 1 << a.log2n
    text = W.repetitive_text_torch(n, 17, dev, base_len=1 << 20, mut_per_1024=10)
    t0 = time.time()
    idx = F.RLFMIndexWithLocate.from_device_text(text.data_ptr(), n, 255, level=a.level,
                                                 force_wide=bool(a.log2n) and not a.narrow,
                                                 walk_records=not a.no_run_table)
    wall = time.time() - t0
    h = idx.handle()
    m, npat = a.m, a.npat
    src = W.umod_torch(W.splitmix64_torch(3, 0, npat, dev), n - 1 - m)
    pat = text[src[:, None] + torch.arange(m, dtype=torch.int64, device=dev)[None, :]].reshape(-1).contiguous()
    off = (torch.arange(npat + 1, dtype=torch.int64, device=dev) * m).contiguous()
    s, e = (torch.empty(npat, dtype=torch.int64, device=dev) for _ in range(2))

    def count():
        assert lib.fmx_count_batch_dev(h, C.c_void_p(pat.data_ptr()), C.c_void_p(off.data_ptr()), npat, None,
                                       C.c_void_p(s.data_ptr()), C.c_void_p(e.data_ptr()), None, None) == 0
    count()
    torch.cuda.synchronize()
    lib.fmx_set_timing(h, 1)
    kms = []
    for _ in range(a.reps):
        count()
        torch.cuda.synchronize()
        kms.append(float(lib.fmx_last_kernel_ms(h)))
    count_steps = int(lib.fmx_last_steps(h))
    sweep = {}
    for v in [x for x in a.sweep_ep_blocks.split(",") if x]:
        if v == "group":
            os.environ["FMXW_R_COUNT_GROUP"] = "1"
        else:
            os.environ["FMXW_EP_BLOCKS"] = v
        ms = []
        for _ in range(a.reps + 1):
            count()
            torch.cuda.synchronize()
            ms.append(float(lib.fmx_last_kernel_ms(h)))
        sweep[v] = round(min(ms[1:]), 4)
        os.environ.pop("FMXW_R_COUNT_GROUP", None)
        os.environ.pop("FMXW_EP_BLOCKS", None)
    assert bool(((e - s) >= 1).all()) and lib.fmx_stream_status(h) == 0
    k = min(a.locate_patterns, npat)
    offh = torch.empty(k + 1, dtype=torch.int64, device=dev)
    lib.fmx_offsets_dev(h, C.c_void_p(s.data_ptr()), C.c_void_p(e.data_ptr()), k, C.c_void_p(offh.data_ptr()), None)
    total = int(offh[-1].item())
    pos = torch.empty(total, dtype=torch.int64, device=dev)

    def locate():
        assert lib.fmx_locate_batch_dev(h, C.c_void_p(s.data_ptr()), C.c_void_p(e.data_ptr()), k,
                                        C.c_void_p(offh.data_ptr()), total, C.c_void_p(pos.data_ptr()), None) == 0
    lib.fmx_set_timing(h, 0)
    locate()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(a.reps):
        locate()
    ev1.record()
    torch.cuda.synchronize()
    call_ms = ev0.elapsed_time(ev1) / a.reps
    lib.fmx_set_timing(h, 1)
    lms = []
    for _ in range(a.reps):
        locate()
        torch.cuda.synchronize()
        lms.append(float(lib.fmx_last_kernel_ms(h)))
    loc_steps = int(lib.fmx_last_steps(h))
    # every position holds its pattern (first 2^22 hits)
    chk = min(total, 1 << 22)
    hit = torch.repeat_interleave(torch.arange(k, device=dev), (e - s)[:k])[:chk]
    ok = torch.ones(chk, dtype=torch.bool, device=dev)
    for j in range(m):
        ok &= text[pos[:chk] + j] == pat.view(npat, m)[hit, j]
    assert bool(ok.all()) and lib.fmx_stream_status(h) == 0
    cm, lm = min(kms), min(lms)
    print(json.dumps({
        "n": n, "wide": bool(idx.is_wide()), "runs": int(lib.fmx_num_runs(h)), "run_table": bool(idx.walk_records()),
        "level": a.level, "index_bytes": idx.heap_size(), "build_ms": round(float(lib.fmx_build_ms(h)), 1),
        "build_wall_s": round(wall, 2),
        "count": {"patterns": npat, "m": m, "kernel_ms": round(cm, 4), "pattern_chars_per_s": npat * m / (cm / 1e3),
                  "steps": count_steps, "sweep_ms": sweep},
        "locate": {"patterns": k, "hits": total, "call_ms": round(call_ms, 4), "walk_kernel_ms": round(lm, 4),
                   "hits_per_s": total / (call_ms / 1e3), "lf_steps": loc_steps,
                   "steps_per_hit": round(loc_steps / max(total, 1), 3)}}))
    idx.close()


if __name__ == "__main__":
    main()
