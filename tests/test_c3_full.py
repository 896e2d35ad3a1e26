"""BASELINE.json configs[2] at its real size on the GPU — head mesh 1024x1024, 8x8 = 64 spp, PathIntegrator depth 5 —
(a) as the multi-GPU job: the K shards `renderer.shard_tile` hands to K ranks, rendered one after another on one
GPU, add up to the one-rank film (what bench.py --gpus K reduces over RCCL; the reference's hook is
Sampler.compute_subwindow, sampler/sampler/__init__.py:87-111), for both shard modes;
(b) one window of the full-size render against the oracle, per sample (1e-9) and on the film;
(c) the WHOLE default render — one 2^26-path batch, k_trace_packet at level 0, k_trace_meta below, chunk cursors, two
streams: the kernels and the schedule the headline number is measured on, at the size they run at — with per-sample
radiance kept for all 67 M samples, of which three windows' rows are compared with the oracle (1e-9)."""
import os

import numpy as np
import pytest

import pytracer_amd as pa
from pytracer_amd import _abi as A
from pytracer_amd import host
from pytracer_amd.renderer import shard_tile

pytestmark = pytest.mark.gpu

RES, XS = 1024, 8


@pytest.fixture(scope='module')
def c3(golden_dir):
    import torch
    from pytracer_amd._lib import Context
    assert torch.cuda.is_available()
    d = dict(np.load(os.path.join(golden_dir, 'head_mesh_scene.npz')))
    flat = pa.flat_scene_from_npz(d)
    ctx = Context(0)
    ctx.upload(flat)
    cam, film = host.head_mesh_camera_film(RES, RES)
    smp = pa.sampler_desc(0, RES, 0, RES, XS, XS, True, 0)
    integ = pa.integrator_desc(A.PT_INTEG_PATH, 5, False)
    yield torch, ctx, flat, cam, film, smp, integ
    ctx.close()


def _render(torch, ctx, cam, film, smp, integ, tiles):
    acc = torch.zeros((film.y_cnt, film.x_cnt, 4), dtype=torch.float64, device='cuda')
    st = A.pt_stats()
    for t in tiles:
        ctx.render(cam, film, smp, integ, t, acc, stats=st)
    torch.cuda.synchronize()
    return acc, st.as_dict()


def test_c3_shards_add_up_to_the_one_rank_film(c3):
    torch, ctx, flat, cam, film, smp, integ = c3
    whole, s1 = _render(torch, ctx, cam, film, smp, integ, [pa.tile_desc(0, RES, 0, RES, 0, XS * XS)])
    assert s1['camera_samples'] == RES * RES * XS * XS
    scale = float(whole.abs().max())
    for K, mode in ((8, 'interleaved'), (3, 'interleaved'), (2, 'samples'), (8, 'samples'), (4, 'rows')):
        tiles = [shard_tile(smp, r, K, mode) for r in range(K)]
        if mode == 'interleaved':
            assert sum(t.rows() for t in tiles) == RES and all((t.y_step, t.y_phase) == (K, r) for r, t in enumerate(tiles))
        if mode == 'samples':
            assert [(t.s0, t.s1) for t in tiles] == [(64 * r // K, 64 * (r + 1) // K) for r in range(K)]
        parts, sk = _render(torch, ctx, cam, film, smp, integ, tiles)
        # same samples, same arithmetic; only the order of the f64 atomic additions differs
        assert float((parts - whole).abs().max()) <= 1e-12 * scale, (K, mode)
        for k in ('camera_samples', 'rays_closest', 'rays_any', 'node_visits', 'tri_tests', 'guard_nan', 'guard_neg',
                  'guard_inf'):
            assert sk[k] == s1[k], (K, mode, k)
        del parts


def test_c3_full_size_window_against_the_oracle(c3):
    import oracle_lib as O
    torch, ctx, flat, cam, film, smp, integ = c3
    osc = O.OracleScene(flat)
    # on the head / head silhouette + backdrop / every third row of a taller window (an interleaved shard)
    for (x0, y0, w, h, step, phase) in ((560, 300, 24, 24, 1, 0), (180, 700, 24, 24, 1, 0), (560, 300, 24, 40, 3, 1)):
        wt = pa.tile_desc(x0, x0 + w, y0, y0 + h, 0, XS * XS, step, phase)
        n = wt.n_samples()
        assert n == w * len(range(y0 + phase, y0 + h, step)) * XS * XS
        acc = torch.zeros((film.y_cnt, film.x_cnt, 4), dtype=torch.float64, device='cuda')
        L = torch.zeros((n, 3), dtype=torch.float64, device='cuda')
        sg, so = A.pt_stats(), A.pt_stats()
        ctx.render(cam, film, smp, integ, wt, acc, sample_L=L, stats=sg)
        torch.cuda.synchronize()
        oacc, oL = osc.render(cam, film, smp, integ, wt, stats=so)
        gL = L.cpu().numpy()
        assert np.isfinite(gL).all()
        scale = np.maximum(np.abs(oL).max(axis=-1, keepdims=True), 1e-12)        # per sample, as in the other parity tests
        assert (np.abs(gL - oL) / scale).max() <= 1e-9
        g = acc.cpu().numpy()
        assert np.abs(g - oacc).max() <= 1e-9 * np.abs(oacc).max()
        for k in ('camera_samples', 'rays_closest', 'rays_any'):
            assert sg.as_dict()[k] == so.as_dict()[k], k
        assert sg.camera_samples == n


WINDOWS = ((560, 300, 24, 24, 1, 0), (180, 700, 24, 24, 1, 0), (560, 300, 24, 40, 3, 1))


def test_c3_whole_default_render_rows_against_the_oracle(c3):
    """Not by transitivity: the production-default launch path at full size, against the oracle directly."""
    import oracle_lib as O
    torch, ctx, flat, cam, film, smp, integ = c3
    osc = O.OracleScene(flat)
    spp = XS * XS
    full = pa.tile_desc(0, RES, 0, RES, 0, spp)
    acc = torch.zeros((film.y_cnt, film.x_cnt, 4), dtype=torch.float64, device='cuda')
    L = torch.empty((RES * RES * spp, 3), dtype=torch.float64, device='cuda')        # 1.6 GB
    ctx.set_profiling(1)
    try:
        sg = A.pt_stats()
        ctx.render(cam, film, smp, integ, full, acc, sample_L=L, stats=sg)
        torch.cuda.synchronize()
        log = ctx.launch_log()
    finally:
        ctx.set_profiling(0)
    # the launch path really is the production default: one batch, level 0 per wave, the rest k_trace_meta / k_trace
    traces = [r for r in log if r['stage'].startswith('trace')]
    assert {r['batch'] for r in log} == {0}
    assert [r['kernel'] for r in traces if r['level'] == 0] == ['k_trace_packet', 'k_trace_packet']
    assert {r['kernel'] for r in traces if r['level'] > 0 and r['stage'] == 'trace_closest'} == {'k_trace_meta'}
    assert sum(r['rays'] for r in traces) == sg.rays_closest + sg.rays_any
    assert sum(r['node_visits'] for r in traces) == sg.node_visits and sg.camera_samples == RES * RES * spp
    for (x0, y0, w, h, step, phase) in WINDOWS:
        wt = pa.tile_desc(x0, x0 + w, y0, y0 + h, 0, spp, step, phase)
        rows = np.arange(y0 + phase, y0 + h, step)
        idx = ((rows[:, None, None] * RES + (x0 + np.arange(w))[None, :, None]) * spp + np.arange(spp)[None, None, :]).ravel()
        gL = L[torch.from_numpy(idx).cuda()].cpu().numpy()
        _, oL = osc.render(cam, film, smp, integ, wt)
        assert gL.shape == oL.shape and np.isfinite(gL).all()
        scale = np.maximum(np.abs(oL).max(axis=-1, keepdims=True), 1e-12)
        assert (np.abs(gL - oL) / scale).max() <= 1e-9, (x0, y0)
    del L
