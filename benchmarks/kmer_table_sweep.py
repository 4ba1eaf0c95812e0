#!/usr/bin/env python3
"""Where the opt-in k-mer start table (FMX_FLAG_KMER_TABLE) pays: count kernel time with and
without it for several alphabets and both index kinds (substring patterns, device-resident
entry point, results asserted identical).  One JSON line per case."""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    import fm_index_amd as F
    from fm_index_amd import workload as W, _lib as L
    lib = L.lib()
    dev = torch.device("cuda", 0)
    log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 28
    n, npat = 1 << log2n, 1 << 20
    for sigma, m in ((4, 32), (7, 32), (12, 24), (20, 24), (60, 16), (255, 16)):
        text = ((W.splitmix64_torch(sigma, 0, n, dev) % sigma) + 1).to(torch.uint8)
        text[-1] = 0
        pos = (W.splitmix64_torch(7, 0, npat, dev) % (n - 1 - m)).to(torch.int64)
        pat = text[(pos[:, None] + torch.arange(m, device=dev)[None, :]).reshape(-1)].contiguous()
        off = (torch.arange(npat + 1, device=dev, dtype=torch.int64) * m).contiguous()
        for kind, cls in (("fm", F.FMIndex), ("rlfm", F.RLFMIndex)):
            res = {}
            for table in (False, True):
                idx = cls.from_device_text(text.data_ptr(), n, sigma, kmer_table=table)
                d_s = torch.empty(npat, dtype=torch.int64, device=dev)
                d_e = torch.empty(npat, dtype=torch.int64, device=dev)

                def run():
                    assert lib.fmx_count_batch_dev(idx.handle(), C.c_void_p(pat.data_ptr()),
                                                   C.c_void_p(off.data_ptr()), npat, None,
                                                   C.c_void_p(d_s.data_ptr()), C.c_void_p(d_e.data_ptr()),
                                                   None, None) == 0
                for _ in range(3):
                    run()
                torch.cuda.synchronize()
                lib.fmx_set_timing(idx.handle(), 1)
                ms = []
                for _ in range(7):
                    run()
                    torch.cuda.synchronize()
                    ms.append(lib.fmx_last_kernel_ms(idx.handle()))
                lib.fmx_set_timing(idx.handle(), 0)
                res[table] = (sorted(ms)[3], idx.kmer_k(), idx.heap_size(), d_s.clone(), d_e.clone())
                idx.close()
            assert bool((res[False][3] == res[True][3]).all()) and bool((res[False][4] == res[True][4]).all())
            print(json.dumps({"sigma": sigma, "kind": kind, "log2n": log2n, "patterns": npat, "pattern_len": m,
                              "kmer_k": res[True][1], "ms_plain": round(res[False][0], 4),
                              "ms_table": round(res[True][0], 4),
                              "speedup": round(res[False][0] / res[True][0], 3),
                              "table_bytes": res[True][2] - res[False][2]}), flush=True)


if __name__ == "__main__":
    main()
