#!/usr/bin/env python3
"""A/B of kernel variants in ONE process, interleaved rounds (cdna_hip_programming.md §5.4 rule 24).

    python scripts/ab_bench.py --env PT_TRACE_TOP=0 --env PT_TRACE_TOP=1 [--rounds 3] [--spp 8]

Each --env arm gets its own pt_ctx (the knobs are read at context creation); every round renders
the bench workload once per arm and prints per-stage kernel time (HIP events) per arm."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--env', action='append', default=[])
    ap.add_argument('--rounds', type=int, default=3)
    ap.add_argument('--res', type=int, default=1024)
    ap.add_argument('--spp', type=int, default=8)
    ap.add_argument('--integrator', default='path')
    ap.add_argument('--workload', choices=['head', 'c2', 'c4', 'c5'], default='head')
    args = ap.parse_args()
    import torch
    import pytracer_amd as pa
    from pytracer_amd import _abi as A, host
    from pytracer_amd._lib import Context
    from pytracer_amd import workloads
    G = os.path.join(ROOT, 'tests', 'golden')
    xres = yres = args.res
    ys = args.spp
    if args.workload == 'head':
        flat = pa.flat_scene_from_npz(dict(np.load(os.path.join(G, 'head_mesh_scene.npz'))))
    elif args.workload == 'c2':
        flat = workloads.c2_head_loop(os.path.join(G, 'c2_head_loop.npz'), os.path.join(G, 'loop_subdiv.npz'))
        xres = yres = 512 if args.res == 1024 else args.res
        args.spp = ys = 4 if args.spp == 8 else args.spp
        args.integrator = 'direct'
    elif args.workload == 'c5':
        flat = workloads.c5_head_measured(os.path.join(G, 'head_mesh_scene.npz'))
        args.spp = 16 if args.spp == 8 else args.spp
        ys = max(args.spp // 2, 1)
    else:
        flat = workloads.c4_soup()
        xres, yres = 1920, 1080
        args.spp = ys = 4 if args.spp == 8 else args.spp
    arms = []
    for e in args.env or ['']:
        kv = dict(x.split('=') for x in e.split(',') if x)
        f32 = kv.pop('F32', '0') == '1'           # pseudo-variable: this arm renders with shade_f32 = 1
        os.environ.update(kv)
        c = Context(0)
        c.upload(flat)
        c.set_profiling(True)
        for k in kv:
            del os.environ[k]
        arms.append((e, c, f32))
    cam, film = (workloads.c4_camera_film(xres, yres) if args.workload == 'c4'
                 else host.head_mesh_camera_film(xres, yres))
    kind = A.PT_INTEG_PATH if args.integrator == 'path' else A.PT_INTEG_DIRECT
    integs = {False: pa.integrator_desc(kind, 5, False), True: pa.integrator_desc(kind, 5, False, shade_f32=True)}
    smp = pa.sampler_desc(0, xres, 0, yres, args.spp, ys, True, 0)
    tile = pa.tile_desc(0, xres, 0, yres, 0, args.spp * ys)
    acc = torch.zeros((film.y_cnt, film.x_cnt, 4), dtype=torch.float64, device='cuda')
    sums = {}
    for rnd in range(args.rounds + 1):
        for name, c, f32 in arms:
            acc.zero_()
            st = A.pt_stats()
            c.render(cam, film, smp, integs[f32], tile, acc, stats=st)
            torch.cuda.synchronize()
            if rnd == 0:
                sums[name] = float(acc.sum().item())
                continue
            t = st.timing()
            rays = st.rays_closest + st.rays_any
            tot = sum(t.values())
            print('%-28s round %d: %s total %.1f ms  %.0f Mrays/s(kernel time)  closest %.0f any %.0f Mrays/s' % (
                name, rnd, ' '.join('%s %.1f' % (k[3:], v) for k, v in t.items()), tot, rays / tot / 1e3,
                st.rays_closest / t['ms_trace_closest'] / 1e3, st.rays_any / t['ms_trace_any'] / 1e3), flush=True)
    print('film sums', sums)


if __name__ == '__main__':
    main()
