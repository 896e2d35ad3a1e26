#!/usr/bin/env python3
"""Feasibility of software-pipelined half-renders: P contexts on ONE GPU render the P interleaved-row shards of the
headline job concurrently on P (x2) streams into one film; time per whole image against the one-context render.
With an argument K the job is shard 0 of K (what one rank of a K-GPU run renders), split into P sub-shards."""
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
sums = {}
K = int(sys.argv[1]) if len(sys.argv) > 1 else 1
for P in (1, 2, 3, 4):
    ctxs = [Context(0) for _ in range(P)]
    for c in ctxs:
        c.upload(flat)
    streams = [torch.cuda.Stream() for _ in range(P)]
    tiles = [shard_tile(smp, r * K, P * K, 'interleaved') for r in range(P)]      # rows r K, r K + P K, ...

    def step():
        acc.zero_()
        torch.cuda.synchronize()
        for r in range(P):
            ctxs[r].render(cam, film, smp, integ, tiles[r], acc, stream=streams[r].cuda_stream)
        torch.cuda.synchronize()
    step()
    step()
    t = time.perf_counter()
    n = 8
    for _ in range(n):
        step()
    dt = (time.perf_counter() - t) / n
    sums[P] = float(acc.sum().item())
    print('K = %d, P = %d concurrent shard renders: %.3f ms per job, film sum %.10e' % (K, P, dt * 1e3, sums[P]), flush=True)
    for c in ctxs:
        c.close()
