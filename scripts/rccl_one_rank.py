#!/usr/bin/env python3
"""One-rank RCCL check on a one-GPU box: the calls bench.py makes for N > 1 (process group on the nccl backend with a
device id, barrier, reduce(sum) and all_reduce(max / sum) of binary64 tensors of the film's size) run through RCCL."""
import os
import time

import torch
import torch.distributed as dist

os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
os.environ.setdefault('MASTER_PORT', '29641')
torch.cuda.set_device(0)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
acc = torch.full((1024, 1024, 4), 1.5, dtype=torch.float64, device='cuda')
dist.barrier()
torch.cuda.synchronize()
for _ in range(3):
    dist.reduce(acc, dst=0)
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(20):
    dist.reduce(acc, dst=0)
torch.cuda.synchronize()
ms = (time.perf_counter() - t) / 20 * 1e3
v = torch.tensor([3.25], dtype=torch.float64, device='cuda')
dist.all_reduce(v, op=dist.ReduceOp.MAX)
w = torch.tensor([1.0, 2.0], dtype=torch.float64, device='cuda')
dist.all_reduce(w)
assert float(acc.sum().item()) == 1.5 * acc.numel() and float(v.item()) == 3.25 and float(w.sum().item()) == 3.0
dist.barrier()
dist.destroy_process_group()
print('rccl one-rank ok: reduce of %d MiB f64 %.3f ms per call' % (acc.numel() * 8 >> 20, ms))
