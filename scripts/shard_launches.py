#!/usr/bin/env python3
"""Per-launch times of the headline render and of shard 0 of K (pt_launch_log), one stream: which launches do not
shrink with the shard (the fixed cost that limits strong scaling).  usage: scripts/shard_launches.py [K ...]"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import pytracer_amd as pa
    from pytracer_amd import _abi as A, host
    from pytracer_amd._lib import Context
    from pytracer_amd.renderer import shard_tile
    Ks = [int(k) for k in sys.argv[1:]] or [1, 8]
    flat = pa.flat_scene_from_npz(dict(np.load(os.path.join(ROOT, 'tests', 'golden', 'head_mesh_scene.npz'))))
    os.environ['PT_OVERLAP'] = '0'
    ctx = Context(0)
    ctx.upload(flat)
    ctx.set_profiling(1)
    cam, film = host.head_mesh_camera_film(1024, 1024)
    smp = pa.sampler_desc(0, 1024, 0, 1024, 8, 8, True, 0)
    integ = pa.integrator_desc(A.PT_INTEG_PATH, 5, False)
    acc = torch.zeros((film.y_cnt, film.x_cnt, 4), dtype=torch.float64, device='cuda')
    out = {}
    for K in Ks:
        t = shard_tile(smp, 0, K, 'interleaved')
        for _ in range(3):
            ctx.render(cam, film, smp, integ, t, acc, stats=A.pt_stats())
        torch.cuda.synchronize()
        out[K] = ctx.launch_log()
    base = out[Ks[0]]
    print('%-14s %-20s %5s | ' % ('stage', 'kernel', 'level') + ' | '.join('K=%d ms (x K / K=%d)' % (K, Ks[0]) for K in Ks))
    for i, r in enumerate(base):
        cells = []
        for K in Ks:
            ms = out[K][i]['ms']
            cells.append('%8.3f (%5.2f)' % (ms, ms * K / max(r['ms'] * Ks[0], 1e-9)))
        print('%-14s %-20s %5d | ' % (r['stage'], r['kernel'], r['level']) + ' | '.join(cells))
    for K in Ks:
        print('K=%d sum of launches %.3f ms' % (K, sum(r['ms'] for r in out[K])))
    json.dump({str(K): out[K] for K in Ks}, open(os.path.join(ROOT, 'gpurun_out', 'shard_launches.json'), 'w'))


if __name__ == '__main__':
    main()
