"""How long do dist.barrier() / a one-element all_reduce + synchronize take on a one-rank RCCL group (the closing
fence of bench.py's timed region)?  usage: python tools/probe_barrier.py"""
import os
import time

import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
one = torch.ones(1, dtype=torch.float64, device=dev)
x = torch.randn(4096, 4096, device=dev)
for name, fn in (("barrier", lambda: dist.barrier()),
                 ("all_reduce(1)+sync", lambda: (dist.all_reduce(one), torch.cuda.synchronize()))):
    out = []
    for i in range(6):
        (x @ x).sum().item()                      # some GPU work, finished before the fence
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        out.append(round((time.perf_counter() - t0) * 1e3, 3))
    print(name, out, "ms")
dist.destroy_process_group()
