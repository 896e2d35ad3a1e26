#!/usr/bin/env python3
"""Micro-benchmark of the traversal kernels through pt_intersect: coherent camera rays and
incoherent secondary rays (random directions from the camera hit points) on the head mesh.

    python scripts/trace_micro.py [--res 1024] [--spp 4] [--reps 5]
Prints Mrays/s from HIP events (pt_set_profiling) for closest and any-hit on both ray sets."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--res', type=int, default=1024)
    ap.add_argument('--spp', type=int, default=4)
    ap.add_argument('--reps', type=int, default=5)
    ap.add_argument('--scene', default='head_mesh_scene.npz')
    args = ap.parse_args()
    import torch
    import pytracer_amd as pa
    from pytracer_amd import _abi as A
    from pytracer_amd._lib import Context
    d = dict(np.load(os.path.join(ROOT, 'tests', 'golden', 'head_mesh_scene.npz')))
    flat = pa.flat_scene_from_npz(d)
    ctx = Context(0)
    ctx.upload(flat)
    ctx.set_profiling(True)
    dev = torch.device('cuda')
    n, spp = args.res, args.spp
    g = torch.Generator(device=dev); g.manual_seed(1)
    # camera rays in (y, x, s) order like the renderer: 64 consecutive rays share a pixel neighbourhood
    ys, xs = torch.meshgrid(torch.arange(n, device=dev, dtype=torch.float64),
                            torch.arange(n, device=dev, dtype=torch.float64), indexing='ij')
    px = (xs[..., None] + torch.rand((n, n, spp), device=dev, dtype=torch.float64, generator=g)).reshape(-1)
    py = (ys[..., None] + torch.rand((n, n, spp), device=dev, dtype=torch.float64, generator=g)).reshape(-1)
    r2c = torch.tensor(d['cam_r2c'], device=dev); r2c[0, 0] *= 128. / n; r2c[1, 1] *= 128. / n
    pr = torch.stack([px, py, torch.zeros_like(px), torch.ones_like(px)])
    pc = r2c @ pr
    pc = pc[:3] / pc[3]
    dirs = pc / pc.norm(dim=0)
    c2w = torch.tensor(d['cam_c2w'], device=dev)
    N = px.numel()
    cam = torch.empty((8, N), device=dev, dtype=torch.float64)
    cam[0:3] = c2w[:3, 3:4]
    cam[3:6] = c2w[:3, :3] @ dirs
    cam[6] = 0.
    cam[7] = float('inf')
    prim, t, _, _ = ctx.intersect(cam)
    hit = prim >= 0
    p = cam[0:3] + cam[3:6] * t
    v = torch.randn((3, N), device=dev, dtype=torch.float64, generator=g)
    v = v / v.norm(dim=0)
    sec = torch.empty((8, N), device=dev, dtype=torch.float64)
    sec[0:3] = p
    sec[3:6] = v
    sec[6] = 1e-3 * t
    sec[7] = float('inf')
    sec = sec[:, hit].contiguous()
    # shadow-like: towards the light at (0,0,-60), finite maxt
    L = torch.tensor([0., 0., -60.], device=dev, dtype=torch.float64)[:, None]
    sh = sec.clone()
    dl = L - sh[0:3]
    dist = dl.norm(dim=0)
    sh[3:6] = dl / dist
    sh[7] = dist * (1 - 1e-5)
    # the same secondary rays regrouped: does direction coherence inside a wave pay?
    octant = ((sec[3] < 0).long() * 4 + (sec[4] < 0).long() * 2 + (sec[5] < 0).long())
    M = sec.shape[1]
    idx = torch.arange(N, device=dev)[hit]            # camera-sample ordinal of every secondary ray
    pix = idx // spp
    py_, px_ = pix // n, pix % n
    sets = [('camera', cam), ('random-secondary', sec), ('to-light', sh)]
    sets.append(('sec octant/all', sec[:, torch.argsort(octant, stable=True)].contiguous()))
    ax = sec[3:6].abs()
    fine = octant * 16 + (ax[0] * 3.999).long().clamp(0, 3) * 4 + (ax[1] * 3.999).long().clamp(0, 3)
    for T in (8, 16, 32, 64):
        tile = (py_ // T) * (n // T) + (px_ // T)
        sets.append(('sec tile%d,octant' % T, sec[:, torch.argsort(tile * 8 + octant, stable=True)].contiguous()))
        sets.append(('sec tile%d,fine128' % T, sec[:, torch.argsort(tile * 128 + fine, stable=True)].contiguous()))
        sets.append(('sec octant,tile%d' % T, sec[:, torch.argsort(octant * (tile.max() + 1) + tile, stable=True)].contiguous()))
    torch.cuda.synchronize()
    for name, rays in sets:
        for any_hit in (False, True):
            best = None
            for _ in range(args.reps):
                st = A.pt_stats()
                ctx.intersect(rays, any_hit=any_hit, stats=st, want_bary=True)
                ms = st.ms_trace_any if any_hit else st.ms_trace_closest
                best = ms if best is None else min(best, ms)
            nr = rays.shape[1]
            print('%-18s %-7s %9d rays  %7.3f ms  %8.1f Mrays/s  visits/ray %.1f tri/ray %.2f' % (
                name, 'any' if any_hit else 'closest', nr, best, nr / best / 1e3, st.node_visits / nr,
                st.tri_tests / nr), flush=True)
    ctx.close()


if __name__ == '__main__':
    main()
