#!/usr/bin/env python3
"""Does the count gather of step k really run under the search of step k+1?  1-rank RCCL communicator on
this GPU, sharding.CountGatherPipeline with tracing, search launched on the default stream or on a side
stream.  usage: rccl_overlap_probe.py [default|side] [log2n] [npat]   (GPU_MAX_HW_QUEUES etc. from the env)"""
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
import fm_index_amd as F  # noqa: E402
from fm_index_amd import _lib as L, launcher, sharding, workload as W  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "default"
log2n = int(sys.argv[2]) if len(sys.argv) > 2 else 26
npat = int(sys.argv[3]) if len(sys.argv) > 3 else 1 << 20
m = 32
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", str(launcher.free_port()))
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
lib = L.lib()
n = 1 << log2n
text = W.dna_text_torch(n, 1, dev)
idx = F.FMIndex.from_device_text(text.data_ptr(), n, 4, device=0)
z = W.splitmix64_torch(3, 0, npat, dev)
src = W.umod_torch(z, n - 1 - m)
pat = text[src[:, None] + torch.arange(m, device=dev)[None, :]].reshape(-1).contiguous()
off = (torch.arange(npat + 1, dtype=torch.int64, device=dev) * m).contiguous()
d_s = torch.empty(npat, dtype=torch.int64, device=dev)
d_e = torch.empty(npat, dtype=torch.int64, device=dev)
stream = torch.cuda.Stream(device=dev) if which == "side" else torch.cuda.current_stream()
sp = C.c_void_p(stream.cuda_stream)


def launch(out64):
    rc = lib.fmx_count_batch_dev(idx.handle(), C.c_void_p(pat.data_ptr()), C.c_void_p(off.data_ptr()), npat, None,
                                 C.c_void_p(d_s.data_ptr()), C.c_void_p(d_e.data_ptr()), C.c_void_p(out64.data_ptr()), sp)
    assert rc == 0


res = {"stream": which, "GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES"), "log2n": log2n, "npat": npat}
with torch.cuda.stream(stream):
    for pipelined in (True, False):
        pipe = sharding.CountGatherPipeline(npat, 1, n, dev, backend="nccl", pipelined=pipelined,
                                            force_collective=True, trace=pipelined)
        for _ in range(5):
            pipe.step(launch)
        pipe.drain()
        torch.cuda.synchronize()
        pipe.events.clear()
        t0 = time.perf_counter()
        for _ in range(40):
            pipe.step(launch)
        pipe.drain()
        torch.cuda.synchronize()
        res["pipelined_ms" if pipelined else "sync_ms"] = (time.perf_counter() - t0) / 40 * 1e3
        if pipelined:
            tr = pipe.trace_report()
            tr["timeline_us"] = tr["timeline_us"][:4]
            res["trace"] = tr
    # search alone
    o = torch.empty(npat, dtype=torch.int64, device=dev)
    for _ in range(5):
        launch(o)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(40):
        launch(o)
    torch.cuda.synchronize()
    res["search_alone_ms"] = (time.perf_counter() - t0) / 40 * 1e3
print(json.dumps(res))
dist.destroy_process_group()
