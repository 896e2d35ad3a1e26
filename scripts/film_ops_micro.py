#!/usr/bin/env python3
"""Times the per-step film operations around pt_render on one GPU: accumulator clear, an on-device add of a second
accumulator (the arithmetic half of the N > 1 reduce) and pt_film_resolve."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from pytracer_amd import host  # noqa: E402
from pytracer_amd._lib import Context  # noqa: E402

cam, film = host.head_mesh_camera_film(1024, 1024)
ctx = Context(0)
acc = torch.zeros((film.y_cnt, film.x_cnt, 4), dtype=torch.float64, device='cuda')
other = torch.ones_like(acc)


def timed(name, f, n=50):
    f()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        f()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / n
    print('%-28s %8.3f ms  (%s)' % (name, dt * 1e3, tuple(acc.shape)), flush=True)


timed('acc.zero_()', lambda: acc.zero_())
timed('acc.add_(other)', lambda: acc.add_(other))
timed('acc.view(-1).add_(other.view(-1))', lambda: acc.view(-1).add_(other.view(-1)))
timed('torch.add(acc, other, out=acc)', lambda: torch.add(acc, other, out=acc))
a32 = acc.view(torch.float32)
timed('f32 view add_ (same bytes)', lambda: a32.add_(a32))
timed('pt_film_resolve', lambda: ctx.film_resolve(acc))
ctx.close()
