import time, torch, numpy as np
dev=torch.device("cuda",0)
def t(f,reps=20):
    f(); torch.cuda.synchronize()
    t0=time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize()
    return (time.perf_counter()-t0)/reps*1e3
for mb in (8,32):
    n=mb<<20
    hp=torch.empty(n,dtype=torch.uint8); hp.fill_(1)
    hpin=torch.empty(n,dtype=torch.uint8).pin_memory(); hpin.fill_(1)
    d=torch.empty(n,dtype=torch.uint8,device=dev)
    print(mb,"MiB pageable H2D ms",t(lambda: d.copy_(hp)), "pinned H2D", t(lambda: d.copy_(hpin,non_blocking=True)),
          "pageable D2H", t(lambda: hp.copy_(d)), "pinned D2H", t(lambda: hpin.copy_(d,non_blocking=True)),
          "host memcpy", t(lambda: hpin.copy_(hp)))
