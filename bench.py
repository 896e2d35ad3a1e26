#!/usr/bin/env python3
"""bench.py — headline benchmark of the hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (BASELINE.json configs[2], the one the metric is quoted on): head mesh (2 778
triangles, 5 537 BVH nodes; tests/golden/head_mesh_scene.npz, a dump of the reference's own
examples/head_mesh.py scene), 1024x1024, 8x8 = 64 spp stratified, PathIntegrator max_depth 5,
Lanczos(4,4,3) film.  One "step" = one full render: sample generation → camera rays → BVH
closest/any hit → shading/light sampling → film splat (+ RCCL reduce of the film for N > 1)
→ film resolve.  Inputs (scene, descriptors) are resident in HBM before the timed region.

Multi-GPU (default, "strong" scaling — the metric's job is the FIXED 1024^2 x 64 spp render at 1/2/4/8 GPUs):
replicated scene, same seed on every rank, rank r renders `renderer.shard_tile(sampler, r, N, 'interleaved')` = the
rows r, r + N, r + 2N, … with all 64 samples of their pixels (balanced like a sample-index split, but every pixel
stays on one rank: coherent waves, one film-footprint pass; `--shard-mode samples|rows` for the other splits; the
reference's own decomposition hook is
Sampler.compute_subwindow, sampler/sampler/__init__.py:87-111, unused there: renderer.py:97-109); the only
collective is one reduce(sum) of the f64 film accumulators to rank 0.  Sample streams are functions of (pixel,
sample index, seed), so the N-rank film equals the 1-rank film up to the order of the additions: rank 0 renders the
1-rank image once more OUTSIDE the timed region and prints both checksums (`image_checksum`, `image_checksum_n1`).
`--weak` keeps the round-1 behaviour (every rank its own 64-spp stream, seed = rank: a 64 N spp image).
`--shard-of 1,2,4,8` (one GPU): times shard 0 of K for every K listed and prints `shard_proxy` =
t(64 spp) / (K t(64/K spp)), the efficiency an N = K run can reach before the film reduce.

Prints ONE JSON line on rank 0 (see the driver contract); extra keys: `roofline` for the
dominant kernel and `cpu_baseline` (the CPU oracle timed on this box's host, rank 0, N=1 only).

`roofline` (N = 1): the top-level achieved / frac / avg_launch_ms are HIP events around the dominant stage's
launches INSIDE the timed region (pt_set_profiling level 2: only the trace launches are bracketed), i.e. the
configuration `value` is quoted on — two streams, kernels sharing the machine.  `roofline.exclusive` is the same
kernel alone on the GPU (two steps after the timed region on a second context with PT_OVERLAP=0), and
`roofline.per_symbol` / `roofline.per_launch` price every traversal kernel symbol and every launch of one such step
by its OWN census (pt_launch_log: rays, node visits, triangle tests and ms per launch).
N > 1: no events in the timed region (none of this is needed there; `collective` says what RCCL saw).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s (spec)


def algorithmic_bytes(n_rays, n_nodes, n_tris):
    """B_ray = 32·N_node + 36·N_tri + 48 (32 B node, 36 B triangle, 48 B ray+hit)."""
    return 32.0 * n_nodes + 36.0 * n_tris + 48.0 * n_rays


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--res', type=int, default=1024)
    ap.add_argument('--spp', type=int, default=8, help='strata per axis (spp = spp^2; c5: spp x spp/2)')
    ap.add_argument('--integrator', choices=['path', 'direct'], default='path')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--shade-f32', action='store_true',
                    help='production-only binary32 BSDF / light arithmetic (pt_integrator_desc.shade_f32); the default and '
                         'the reported headline compute everything in binary64')
    ap.add_argument('--weak', action='store_true', help='N > 1: per-rank sample streams (seed = rank), 64 spp per rank')
    ap.add_argument('--shard-of', default='', help='one GPU: comma list of K; time shard 0 of K for each (scaling proxy)')
    ap.add_argument('--shard-mode', choices=['interleaved', 'samples', 'rows'], default='interleaved')
    ap.add_argument('--workload', choices=['head', 'c2', 'c4', 'c5'], default='head',
                    help='head = BASELINE.json configs[2] (the metric; default).  Secondary, for DESIGN.md only: '
                         'c2 = configs[1] head_loop 512^2 x 16 spp Direct; c4 = configs[3] 10^6-triangle soup '
                         '1920x1080 Path (use --spp 16 for the full 256 spp); c5 = configs[4] as far as the reference '
                         'can run it: head mesh with a measured isotropic BRDF (RGB) + checkerboard backdrop, 1024^2, '
                         '16x8 = 128 spp Path')
    return ap.parse_args()


def cpu_baseline(flat, cam, film, integ, res, xs):
    """The oracle (CPU restatement, 1 thread) on a bounded window of the same workload."""
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import oracle_lib as O
    import pytracer_amd as pa
    from pytracer_amd import _abi as A
    osc = O.OracleScene(flat)
    win = min(320, res)                         # 320x320 pixels x 64 spp ≈ 6.6 M camera samples, ≈10 s
    x0 = (res - win) // 2
    smp = pa.sampler_desc(0, res, 0, res, xs, xs, True, 0)
    tile = pa.tile_desc(x0, x0 + win, x0, x0 + win, 0, xs * xs)
    st = A.pt_stats()
    t0 = time.perf_counter()
    osc.render(cam, film, smp, integ, tile, stats=st, want_L=False)
    dt = time.perf_counter() - t0
    rays = st.rays_closest + st.rays_any
    out = {'value': rays / dt / 1e6, 'unit': 'Mrays/s', 'cores': 1, 'kind': 'port',
           'samples_per_s': st.camera_samples / dt,
           'sample': 'oracle/pt_oracle.c (1 thread), %dx%d-pixel centre window of the same %dx%d x %d spp render: '
                     '%d camera samples, %d rays in %.1f s' % (win, win, res, res, xs * xs, st.camera_samples, rays, dt)}
    # also "at all host cores" — the same restatement on T threads (pto_render is re-entrant,
    # ctypes drops the GIL), each thread one band of rows of a 2x larger window, its own film accumulator.
    import threading
    try:
        T = len(os.sched_getaffinity(0))
    except AttributeError:
        T = os.cpu_count() or 1
    T = max(1, min(T, 16))                      # the GPU box's CPU share for one GPU is 16
    win2 = min(2 * win, res)
    x0 = (res - win2) // 2
    stats = [A.pt_stats() for _ in range(T)]
    bands = [(x0 + (win2 * k) // T, x0 + (win2 * (k + 1)) // T) for k in range(T)]

    def work(k):
        t = pa.tile_desc(x0, x0 + win2, bands[k][0], bands[k][1], 0, xs * xs)
        osc.render(cam, film, smp, integ, t, stats=stats[k], want_L=False)
    th = [threading.Thread(target=work, args=(k,)) for k in range(T)]
    t0 = time.perf_counter()
    for t in th:
        t.start()
    for t in th:
        t.join()
    dt = time.perf_counter() - t0
    rays = sum(s_.rays_closest + s_.rays_any for s_ in stats)
    out['all_cores'] = {'value': rays / dt / 1e6, 'unit': 'Mrays/s', 'cores': T,
                        'samples_per_s': sum(s_.camera_samples for s_ in stats) / dt,
                        'sample': 'same restatement, %d threads, %dx%d-pixel window in %d row bands: %d rays in %.1f s'
                                  % (T, win2, win2, T, rays, dt)}
    return out


def main():
    args = parse()
    import torch
    import pytracer_amd as pa
    from pytracer_amd import _abi as A
    from pytracer_amd import host
    from pytracer_amd._lib import Context

    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        # PT_BENCH_BACKEND=gloo: rehearsal of the N > 1 control flow on a box with fewer GPUs than ranks
        # (ranks share devices, the film reduce goes through the host); never used for reported numbers
        backend = os.environ.get('PT_BENCH_BACKEND', 'nccl')
        if backend != 'nccl':
            local = local % torch.cuda.device_count()
        torch.cuda.set_device(local)
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device('cuda', local))
        else:
            dist.init_process_group(backend)
    else:
        torch.cuda.set_device(0)
    dev = torch.device('cuda', local if world > 1 else 0)

    res, xs = args.res, args.spp
    ys = None
    kind = A.PT_INTEG_PATH if args.integrator == 'path' else A.PT_INTEG_DIRECT
    xres = yres = res
    if args.workload == 'head':
        d = dict(np.load(os.path.join(ROOT, 'tests', 'golden', 'head_mesh_scene.npz')))
        flat = pa.flat_scene_from_npz(d)
        cam, film = host.head_mesh_camera_film(res, res)
        wname = 'head_mesh %dx%d, %d spp stratified, %s max_depth 5, Lanczos(4,4,3) film (BASELINE.json configs[2])'
    elif args.workload == 'c2':
        from pytracer_amd import workloads
        flat = workloads.c2_head_loop(os.path.join(ROOT, 'tests', 'golden', 'c2_head_loop.npz'),
                                      os.path.join(ROOT, 'tests', 'golden', 'loop_subdiv.npz'))
        xres = yres = res = 512 if args.res == 1024 else args.res
        xs = 4 if args.spp == 8 else args.spp
        kind = A.PT_INTEG_DIRECT
        cam, film = host.head_mesh_camera_film(res, res)
        wname = 'head_loop (Loop-subdivided head) %dx%d, %d spp, %s, Lanczos(4,4,3) film (BASELINE.json configs[1])'
    elif args.workload == 'c5':
        from pytracer_amd import workloads
        flat = workloads.c5_head_measured(os.path.join(ROOT, 'tests', 'golden', 'head_mesh_scene.npz'))
        xs = 16 if args.spp == 8 else args.spp
        ys = max(xs // 2, 1)
        cam, film = host.head_mesh_camera_film(res, res)
        wname = ('head_mesh with a measured isotropic BRDF (1000 RGB samples) + checkerboard backdrop %dx%d, %d spp, %s '
                 'max_depth 5 (BASELINE.json configs[4], RGB: the reference hard-wires Spectrum = RGBSpectrum)')
    else:
        from pytracer_amd import workloads
        flat = workloads.c4_soup()
        xres, yres = 1920, 1080
        xs = 4 if args.spp == 8 else args.spp
        cam, film = workloads.c4_camera_film(xres, yres)
        wname = '10^6-triangle soup %dx%d, %d spp, %s max_depth 5, spot + sphere lights (BASELINE.json configs[3])'
    ctx = Context(dev.index)
    ctx.upload(flat)
    # N = 1: HIP events around the trace launches of the timed region (the roofline's live figure); N > 1: none
    ctx.set_profiling(2 if world == 1 else 0)
    integ = pa.integrator_desc(kind, 5, pretest=False, shade_f32=args.shade_f32)
    ys = xs if ys is None else ys
    from pytracer_amd.renderer import shard_tile
    strong = world > 1 and not args.weak
    smp = pa.sampler_desc(0, xres, 0, yres, xs, ys, True, seed=rank if (world > 1 and args.weak) else 0)
    full_tile = pa.tile_desc(0, xres, 0, yres, 0, xs * ys)
    tile = shard_tile(smp, rank, world, args.shard_mode) if strong else full_tile
    acc = torch.zeros((film.y_cnt, film.x_cnt, 4), dtype=torch.float64, device=dev)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    reduce_ev = []                              # (start, end) CUDA events around every film reduce of the timed region

    def step(stats, t=None, timed=False):
        acc.zero_()
        ctx.render(cam, film, smp, integ, tile if t is None else t, acc, stats=stats)
        if dist is not None:
            if timed:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            dist.reduce(acc, dst=0)             # RCCL sum of film accumulators over xGMI
            if timed:
                e1.record()
                reduce_ev.append((e0, e1))
        return ctx.film_resolve(acc)

    for _ in range(args.warmup):
        step(A.pt_stats())
    barrier()
    st = A.pt_stats()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        rgb = step(st, timed=True)
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        cnt = torch.tensor([st.rays_closest + st.rays_any, st.camera_samples], dtype=torch.float64, device=dev)
        dist.all_reduce(cnt)
        rays_total, samples_total = float(cnt[0].item()), float(cnt[1].item())
        # what the collective saw, from this process: backend, world size, and the time of the film reduce as the
        # render stream saw it (events on the current stream around dist.reduce; max over ranks)
        rms = torch.tensor([sum(a.elapsed_time(b) for a, b in reduce_ev) / max(len(reduce_ev), 1)], dtype=torch.float64, device=dev)
        dist.all_reduce(rms, op=dist.ReduceOp.MAX)
        collective = {'op': 'reduce(sum) of the f64 film accumulators to rank 0', 'backend': dist.get_backend(),
                      'world_size': dist.get_world_size(), 'bytes': acc.numel() * 8,
                      'reduce_ms_per_step_max_over_ranks': float(rms.item()),
                      'devices': torch.cuda.device_count(), 'device_name': torch.cuda.get_device_name(dev)}
    else:
        collective = None
        rays_total, samples_total = float(st.rays_closest + st.rays_any), float(st.camera_samples)

    # N > 1, strong scaling: the same job on one rank, outside the timed region, for the checksum comparison
    checksum_n1 = None
    if strong and rank == 0:
        acc1 = torch.zeros_like(acc)
        ctx.render(cam, film, smp, integ, full_tile, acc1, stats=A.pt_stats())
        checksum_n1 = float(ctx.film_resolve(acc1).sum().item())
        del acc1
    # one GPU: shard 0 of K for every K asked for (what an N = K run can reach before the film reduce)
    proxy = None
    if world == 1 and args.shard_of:
        proxy = {}
        t_full = None
        ctx.set_profiling(0)                    # as a rank of an N > 1 run renders: no events around the launches
        for K in sorted(set([1] + [int(k) for k in args.shard_of.split(',') if k.strip()])):
            tk = shard_tile(smp, 0, K, args.shard_mode)
            step(A.pt_stats(), tk)
            torch.cuda.synchronize()
            stk = A.pt_stats()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                step(stk, tk)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t1) / args.steps * 1e3
            if K == 1:
                t_full = ms
            proxy[str(K)] = {'ms_per_step': ms, 'spp_per_shard': tk.s1 - tk.s0, 'rows_per_shard': tk.rows(),
                             'predicted_efficiency': t_full / (K * ms)}
        # the reduce itself cannot be timed on one GPU: its on-device half (acc += other, 32 MiB f64) can
        other = torch.ones_like(acc)
        acc.add_(other)                         # first use loads torch's kernel: not part of the figure
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(20):
            acc.add_(other)
        torch.cuda.synchronize()
        proxy['film_add_ms'] = (time.perf_counter() - t1) / 20 * 1e3
        proxy['film_bytes'] = acc.numel() * 8
        proxy['note'] = ('predicted_efficiency = t(K=1) / (K t(shard 0 of K)), same process, film resolve included; the '
                         'xGMI transfer of the reduce is not measurable on one GPU: film_bytes over a ring at ~153 GB/s '
                         'per link is ~0.2-0.4 ms (SURVEY.md section 5)')
        del other

    # The render runs its any-hit traces on a second stream beside the next closest-hit trace and shading
    # (PT_OVERLAP, default on), so per-kernel durations of the timed region include the slow-down of sharing the
    # machine.  For the roofline of the kernel BY ITSELF, two extra steps on a second context with the overlap off
    # (outside the timed region).
    excl, excl_log = None, None
    if rank == 0 and world == 1:
        prev = os.environ.get('PT_OVERLAP')
        os.environ['PT_OVERLAP'] = '0'
        try:
            cx = Context(dev.index)
        finally:
            if prev is None:
                del os.environ['PT_OVERLAP']
            else:
                os.environ['PT_OVERLAP'] = prev
        cx.upload(flat)
        cx.set_profiling(1)
        accx = torch.zeros_like(acc)
        cx.render(cam, film, smp, integ, tile, accx, stats=A.pt_stats())
        excl = A.pt_stats()
        for _ in range(2):
            accx.zero_()
            cx.render(cam, film, smp, integ, tile, accx, stats=excl)
        torch.cuda.synchronize()
        excl_log = cx.launch_log()              # every launch of the last of these steps, with its own census
        cx.close()
        del accx

    if rank == 0:
        # Dominant kernel: whichever of the two traversal stages takes more time BY ITSELF (the exclusive steps; the
        # timed region's own events when there are none).  Top-level figures: HIP events over the timed region, on
        # the stream each kernel is launched on — with the render's two streams (PT_OVERLAP, default) those durations
        # include the time a kernel shares the machine with the other stream's kernels; `exclusive` prices the
        # kernel alone (what rocprofv3 --kernel-trace reports for the one-stream command, profiles/).
        def per_kernel(stt):
            n_any, n_clo = stt.rays_any, stt.rays_closest
            nv_any, nt_any = stt.node_visits_any, stt.tri_tests_any
            nv_clo, nt_clo = stt.node_visits - nv_any, stt.tri_tests - nt_any
            return {
                'k_trace<closest>': (stt.ms_trace_closest, stt.launches_trace_closest,
                                     algorithmic_bytes(n_clo, nv_clo, nt_clo), n_clo, nv_clo, nt_clo),
                'k_trace<any>': (stt.ms_trace_any, stt.launches_trace_any, algorithmic_bytes(n_any, nv_any, nt_any),
                                 n_any, nv_any, nt_any),
            }

        def figures(stt, steps, name):
            ms, launches, abytes, nr, nv, nt = per_kernel(stt)[name]
            ach = abytes / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
            return {'achieved': ach, 'frac': ach / HBM_PEAK_GBS, 'launches': int(launches),
                    'avg_launch_ms': ms / max(launches, 1), 'algorithmic_bytes_per_launch': abytes / max(launches, 1),
                    'rays': int(nr), 'node_visits_per_ray': nv / max(nr, 1), 'tri_tests_per_ray': nt / max(nr, 1),
                    'kernel_Mrays_per_s': nr / (ms * 1e-3) / 1e6 if ms > 0 else 0.0,
                    'stage_ms_per_step': {k: v / steps for k, v in stt.timing().items()}}
        pick = per_kernel(excl if excl is not None else st)
        name = max(pick, key=lambda k: pick[k][0])
        live = figures(st, args.steps, name)
        # `traffic` is NOT measured by this run: PMC counters need rocprofv3 passes of their own.  It is the recorded
        # figure of profiles/pmc_summary.json (2 x FETCH_SIZE + WRITE_SIZE per launch, summed over every kernel
        # symbol of this stage on the headline workload), valid for the commit named there; other workloads get null.
        traffic, traffic_source, traffic_sym = None, None, None
        pmc = os.path.join(ROOT, 'profiles', 'pmc_summary.json')
        headline = args.workload == 'head' and res == 1024 and xs * ys == 64 and kind == A.PT_INTEG_PATH and world == 1
        if os.path.exists(pmc) and headline:
            try:
                pj = json.load(open(pmc))
                traffic = pj.get(name, {}).get('hbm_bytes_per_launch')
                traffic_sym = {k: {'launches': v.get('launches'), 'hbm_bytes_per_launch': v.get('hbm_bytes_per_launch')}
                               for k, v in pj.get('_symbols', {}).items() if v.get('stage') == name}
                meta = pj.get('_meta', {})
                traffic_source = {'file': 'profiles/pmc_summary.json', 'csv': pj.get(name, {}).get('source'),
                                  'commit': meta.get('commit'), 'recorded': meta.get('recorded'),
                                  'note': 'recorded rocprofv3 --pmc passes (one stream), not measured in this run'}
            except Exception:
                traffic = None
        roof = {'bound': 'hbm', 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'kernel': name}
        if world > 1:                           # no events in an N > 1 timed region: the roofline is an N = 1 figure
            live = {k: (None if k in ('achieved', 'frac', 'avg_launch_ms') else v) for k, v in live.items()}
            live['stage_ms_per_step'] = None
        roof.update(live)
        roof.update({
            'measured': 'HIP events around every launch of this stage inside the timed region, on the stream it is launched '
                        'on' + (' (two streams: the durations include sharing the machine with the other stream\'s kernels; '
                                'see `exclusive`)' if os.environ.get('PT_OVERLAP', '1') != '0' else ' (one stream)'),
            'traffic': traffic, 'traffic_per_symbol': traffic_sym, 'traffic_source': traffic_source,
            # `frac` prices every node / triangle byte as if streamed from HBM.  The tree is
            # cache-resident: what the HBM actually carries per launch is `traffic`, and the measured limiter is below.
            'bound_measured': 'cache latency: L1/L2-resident tree, dependent fetch chain per visit (not HBM bandwidth)',
            'limiter': 'the dependent chain of a visit (record fetch from L1/L2 -> binary64 slab test -> stack -> next '
                       'fetch, plus a triangle block in most steps of a divergent wave), not the number of requests '
                       '(a 16-byte-node form with a third of them runs as fast); HBM carries the ray/hit streams only; '
                       'the level-0 launches (k_trace_packet) are binary64-VALU-bound (DESIGN.md section 5-6)',
        })
        if excl is not None:
            ex = figures(excl, 2, name)
            ex['measured'] = ('2 steps after the timed region on a second context with PT_OVERLAP=0 (every kernel alone on '
                              'the GPU, every launch timed)')
            ex['hbm_traffic_frac'] = (traffic / (ex['avg_launch_ms'] * 1e-3) / 1e9 / HBM_PEAK_GBS) if (traffic and ex['avg_launch_ms'] > 0) else None
            roof['exclusive'] = ex
            # every trace launch of ONE exclusive step, priced by its own census, and the same per kernel symbol
            per_launch, per_symbol = [], {}
            for r in excl_log:
                if not r['stage'].startswith('trace') or r['ms'] <= 0 or r['rays'] == 0:
                    continue
                b = algorithmic_bytes(r['rays'], r['node_visits'], r['tri_tests'])
                sym = '%s<%s>' % (r['kernel'], 'true' if r['stage'] == 'trace_any' else 'false')
                e = {'symbol': sym, 'level': r['level'], 'batch': r['batch'], 'rays': r['rays'],
                     'node_visits_per_ray': r['node_visits'] / r['rays'], 'tri_tests_per_ray': r['tri_tests'] / r['rays'],
                     'B_ray': b / r['rays'], 'ms': r['ms'], 'Grays_per_s': r['rays'] / r['ms'] / 1e6,
                     'achieved': b / r['ms'] / 1e6, 'frac': b / r['ms'] / 1e6 / HBM_PEAK_GBS}
                per_launch.append(e)
                a = per_symbol.setdefault(sym, {'launches': 0, 'rays': 0, 'bytes': 0.0, 'ms': 0.0})
                a['launches'] += 1; a['rays'] += r['rays']; a['bytes'] += b; a['ms'] += r['ms']
            for sym, a in per_symbol.items():
                a['B_ray'] = a['bytes'] / max(a['rays'], 1)
                a['avg_launch_ms'] = a['ms'] / a['launches']
                a['Grays_per_s'] = a['rays'] / a['ms'] / 1e6
                a['achieved'] = a['bytes'] / a['ms'] / 1e6
                a['frac'] = a['achieved'] / HBM_PEAK_GBS
                a['bound_measured'] = ('binary64 VALU issue (one walk per wave, scalar record loads)' if sym.startswith('k_trace_packet')
                                       else 'latency + L2 misses (tree beyond the L2)' if flat.n_nodes >= (1 << 18)
                                       else 'L1/L2 latency chain per visit')
                del a['bytes']
            roof['per_symbol'] = per_symbol
            if len(per_launch) > 64:                       # many wavefront batches (C4 at 256 spp): first batch only
                per_launch = [e for e in per_launch if e['batch'] == 0]
            roof['per_launch'] = per_launch
            roof['per_launch_note'] = 'one exclusive step (PT_OVERLAP=0), each launch with its own census (pt_launch_log)'
        out = {
            'metric': 'Mrays/s', 'value': rays_total / dt / 1e6, 'unit': 'Mrays/s',
            'samples_per_s': samples_total / dt,
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': dt / args.steps * 1e3,
            'higher_is_better': True, 'scaling': 'weak' if args.weak else 'strong', 'vs_baseline': None,
            'dtype': 'f64' if not args.shade_f32 else 'f64 (traversal, points, state, film) + f32 (BSDF / light arithmetic)',
            'data': 'synthetic',
            'config': {'workload': wname % (xres, yres, xs * ys, 'PathIntegrator' if kind else 'DirectLightingIntegrator'),
                       'triangles': flat.n_tris, 'bvh_nodes': flat.n_nodes,
                       'rays_per_sample': rays_total / max(samples_total, 1),
                       'parallelism': ('single GPU' if world == 1 else
                                       'replicated scene, per-rank sample streams (seed = rank), RCCL film reduce' if args.weak
                                       else 'replicated scene, fixed job sharded by %s (shard_tile), one RCCL film reduce'
                                       % {'samples': 'sample index', 'rows': 'row bands', 'interleaved': 'interleaved rows'}[args.shard_mode]),
                       'image_checksum': float(rgb.sum().item())},
            'roofline': roof,
        }
        if checksum_n1 is not None:
            out['config']['image_checksum_n1'] = checksum_n1
            out['config']['checksum_rel_diff'] = abs(out['config']['image_checksum'] - checksum_n1) / max(abs(checksum_n1), 1e-300)
            assert out['config']['checksum_rel_diff'] <= 1e-9, 'the sharded film differs from the 1-rank film: %r' % out['config']
        if proxy is not None:
            out['shard_proxy'] = proxy
        if collective is not None:
            out['collective'] = collective
        if world == 1 and not args.no_cpu_baseline and args.workload == 'head':
            out['cpu_baseline'] = cpu_baseline(flat, cam, film, integ, res, xs)
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


if __name__ == '__main__':
    main()
