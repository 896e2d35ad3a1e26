#!/usr/bin/env python3
"""Two contexts on ONE GPU render the two interleaved-row halves of the headline job concurrently, the second one
delayed by d ms (a spin kernel on its stream), so that its VALU-bound level-0 launches meet the first one's
request-bound deeper levels.  Time per whole image against the one-context render."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import pytracer_amd as pa  # noqa: E402
from pytracer_amd import _abi as A, host  # noqa: E402
from pytracer_amd._lib import Context  # noqa: E402
from pytracer_amd.renderer import shard_tile  # noqa: E402

flat = pa.flat_scene_from_npz(dict(np.load(os.path.join(ROOT, 'tests', 'golden', 'head_mesh_scene.npz'))))
res, xs = 1024, 8
cam, film = host.head_mesh_camera_film(res, res)
smp = pa.sampler_desc(0, res, 0, res, xs, xs, True, 0)
integ = pa.integrator_desc(A.PT_INTEG_PATH, 5, False)
acc = torch.zeros((film.y_cnt, film.x_cnt, 4), dtype=torch.float64, device='cuda')
ctxs = [Context(0), Context(0)]
for c in ctxs:
    c.upload(flat)
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
tiles = [shard_tile(smp, r, 2, 'interleaved') for r in range(2)]
full = shard_tile(smp, 0, 1)
# spin cycles per ms
torch.cuda.synchronize()
t = time.perf_counter(); torch.cuda._sleep(200_000_000); torch.cuda.synchronize()
cyc_per_ms = 200_000_000 / ((time.perf_counter() - t) * 1e3)


def run(delay_ms, n=8):
    def step():
        acc.zero_()
        torch.cuda.synchronize()
        if delay_ms is None:
            ctxs[0].render(cam, film, smp, integ, full, acc, stream=streams[0].cuda_stream)
        else:
            ctxs[0].render(cam, film, smp, integ, tiles[0], acc, stream=streams[0].cuda_stream)
            if delay_ms > 0:
                with torch.cuda.stream(streams[1]):
                    torch.cuda._sleep(int(delay_ms * cyc_per_ms))
            ctxs[1].render(cam, film, smp, integ, tiles[1], acc, stream=streams[1].cuda_stream)
        torch.cuda.synchronize()
    step(); step()
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    return (time.perf_counter() - t0) / n * 1e3, float(acc.sum().item())


print('one context, whole image: %.2f ms (sum %.10e)' % run(None), flush=True)
for d in (0, 3, 6, 9, 12, 16, 20, 26):
    print('two halves, second delayed %2d ms: %.2f ms (sum %.10e)' % ((d,) + run(d)), flush=True)
for c in ctxs:
    c.close()
