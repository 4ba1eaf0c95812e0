"""Would processing the patterns of a batch in suffix-sorted order pay?  Backward search consumes a pattern
from its END, so patterns sorted by their reversed text follow the same (s, e) path for the symbols they
share and neighbouring groups of a wave ask for the same record lines.  This probe sorts the config-2
patterns on the device with torch (outside the timing), runs the shipped count kernel on the sorted
buffer and prints kernel ms against the unsorted run, plus what a device radix sort of the keys costs.
    python benchmarks/gpu/sorted_patterns_probe.py"""
import argparse, json, os, sys, time
sys.path.insert(0, os.getcwd())
import torch
import bench as B

args = B.parse_args(["--no-rlfm"])
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
wl = B.Workload("dna", args, dev, 0, 0, 1)
npat, m = wl.npat, wl.m
wl.count(); torch.cuda.synchronize()
base = [wl.timed_kernel(wl.count)[0] for _ in range(5)]
ref_cnt = wl.d_c.clone()
P = wl.pat.view(npat, m)
out = {"unsorted_kernel_ms": sum(base) / len(base)}
for ksym in (8, 10, 12, 16):
    key = torch.zeros(npat, dtype=torch.int64, device=dev)
    for i in range(ksym):                      # last symbol most significant
        key = key * 8 + P[:, m - 1 - i].to(torch.int64)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    perm = torch.argsort(key)
    torch.cuda.synchronize(); sort_ms = (time.perf_counter() - t0) * 1e3
    sp = P[perm].contiguous().view(-1)
    wl.count(pat=sp); torch.cuda.synchronize()
    ms = [wl.timed_kernel(lambda: wl.count(pat=sp))[0] for _ in range(5)]
    assert bool((wl.d_c == ref_cnt[perm]).all())
    out["sorted_by_last_%d" % ksym] = {"kernel_ms": sum(ms) / len(ms), "torch_argsort_ms": round(sort_ms, 3)}
print(json.dumps(out))
