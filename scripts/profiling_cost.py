#!/usr/bin/env python3
"""Wall-clock cost of the per-stage HIP events (Context.set_profiling) on the headline render."""
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

flat = pa.flat_scene_from_npz(dict(np.load(os.path.join(ROOT, 'tests', 'golden', 'head_mesh_scene.npz'))))
res, xs = 1024, 8
cam, film = host.head_mesh_camera_film(res, res)
smp = pa.sampler_desc(0, res, 0, res, xs, xs, True, 0)
tile = pa.tile_desc(0, res, 0, res, 0, xs * xs)
integ = pa.integrator_desc(A.PT_INTEG_PATH, 5, False)
acc = torch.zeros((film.y_cnt, film.x_cnt, 4), dtype=torch.float64, device='cuda')
ctx = Context(0)
ctx.upload(flat)
for rnd in range(3):
    for prof, stats in ((True, True), (False, True), (False, False)):
        ctx.set_profiling(prof)
        for _ in range(2):
            ctx.render(cam, film, smp, integ, tile, acc, stats=A.pt_stats() if stats else None)
        torch.cuda.synchronize()
        t = time.perf_counter()
        n = 10
        for _ in range(n):
            acc.zero_()
            ctx.render(cam, film, smp, integ, tile, acc, stats=A.pt_stats() if stats else None)
            ctx.film_resolve(acc)
        torch.cuda.synchronize()
        print('round %d profiling %-5s stats %-5s: %.2f ms per step' % (rnd, prof, stats, (time.perf_counter() - t) / n * 1e3), flush=True)
ctx.close()
