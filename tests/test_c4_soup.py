"""C4 (BASELINE.json configs[3]): synthetic triangle soup, spot + sphere area light, Path depth 5.

The reference cannot build this scene in any reasonable time (hours of Python), so parity rests on
(1) the reference-pinned miniature (tests/test_c4mini.py: same light and material kinds), (2) the
reference-exact builder (tests/test_bvh_build_cpu.py), and here (3) the oracle on a 20 k-triangle
soup of the same recipe and (4) the 10^6-triangle scene through size-independent properties and a
ray batch checked against the oracle (closest hit brute-forced for a subset)."""
import os

import numpy as np
import pytest

import pytracer_amd as pa
from pytracer_amd import _abi as A, workloads as W
import oracle_lib as O

RTOL = 1e-9


def rel_err(a, ref):
    scale = np.maximum(np.abs(ref).max(axis=-1, keepdims=True), 1e-12)
    return (np.abs(a - ref) / scale).max()


def test_c4_recipe_on_cpu():
    """Host side only: the scene description is deterministic and well-formed."""
    flat = W.c4_soup(5000, seed=1)
    assert flat.n_tris == 5000 and flat.n_nodes == 2 * 5000 - 1 and flat.n_lights == 2
    assert sorted(int(r['kind']) for r in flat.light_records) == [A.PT_LIGHT_DIFFUSE_SPHERE, A.PT_LIGHT_SPOT]
    assert np.array_equal(flat.tri_verts, W.c4_soup(5000, seed=1).tri_verts)
    cam, film = W.c4_camera_film(192, 108)
    assert (film.x_cnt, film.y_cnt) == (192, 108)
    # the centre pixel looks down +z from (0,0,-30)
    rays = O.camera_rays(cam, np.array([[96., 54., 0., 0., 0.]]))
    np.testing.assert_allclose(rays[0], [0, 0, -30, 0, 0, 1], atol=1e-12)


@pytest.fixture(scope='module')
def ctx():
    import torch
    assert torch.cuda.is_available(), 'these tests need the MI355X'
    from pytracer_amd._lib import Context
    c = Context(0)
    yield c
    c.close()


def _render(ctx, flat, res, spp, kind, tile=None, seed=3, stats=None):
    import torch
    cam, film = W.c4_camera_film(*res)
    smp = pa.sampler_desc(0, res[0], 0, res[1], spp, spp, True, seed)
    if tile is None:
        tile = pa.tile_desc(0, res[0], 0, res[1], 0, spp * spp)
    n = (tile.x1 - tile.x0) * (tile.y1 - tile.y0) * (tile.s1 - tile.s0)
    acc = torch.zeros((film.y_cnt, film.x_cnt, 4), dtype=torch.float64, device='cuda')
    L = torch.zeros((n, 3), dtype=torch.float64, device='cuda')
    ctx.render(cam, film, smp, pa.integrator_desc(kind, 5, False), tile, acc, sample_L=L, stats=stats)
    torch.cuda.synchronize()
    return acc.cpu().numpy(), L.cpu().numpy(), (cam, film, smp, tile)


@pytest.mark.gpu
@pytest.mark.parametrize('kind', [A.PT_INTEG_DIRECT, A.PT_INTEG_PATH])
def test_small_soup_vs_oracle(ctx, kind):
    """20 k triangles, scaled up so the image is busy: whole renders against the oracle."""
    flat = W.c4_soup(20000, seed=7, edge_sigma=.35)
    ctx.upload(flat)
    sg, so = A.pt_stats(), A.pt_stats()
    acc, L, (cam, film, smp, tile) = _render(ctx, flat, (96, 54), 2, kind, stats=sg)
    oacc, oL = O.OracleScene(flat).render(cam, film, smp, pa.integrator_desc(kind, 5, False), tile, stats=so)
    assert (np.abs(oL).sum(axis=1) > 0).mean() > .02
    assert rel_err(L, oL) < RTOL
    # film pixels sum signed Lanczos weights x radiance spanning many decades (spot falloff): the
    # absolute floor is relative to the largest accumulator
    np.testing.assert_allclose(acc, oacc, rtol=RTOL, atol=RTOL * np.abs(oacc).max())
    a, b = sg.as_dict(), so.as_dict()
    for k in ('camera_samples', 'rays_closest', 'rays_any', 'guard_nan', 'guard_neg', 'guard_inf'):
        assert a[k] == b[k], (k, a, b)


@pytest.mark.gpu
def test_million_triangle_soup(ctx):
    import torch
    flat = W.c4_soup(10 ** 6)
    assert flat.n_nodes == 2 * 10 ** 6 - 1
    ctx.upload(flat)
    osc = O.OracleScene(flat)
    # (a) a ray batch: camera-like rays + random segments, GPU == oracle bitwise, counters included
    rng = np.random.default_rng(5)
    n = 20000
    o = np.tile(np.array([0., 0., -30.]), (n, 1)) + rng.normal(0, .01, (n, 3))
    tgt = rng.uniform(-10, 10, (n, 3))
    rays = np.zeros((8, n))
    rays[0:3] = o.T
    rays[3:6] = (tgt - o).T
    rays[7] = np.where(rng.uniform(size=n) < .5, np.inf, 1.)
    rays[:, n // 2:][0:3] = rng.uniform(-10, 10, (3, n - n // 2))          # segments inside the soup
    rt = torch.from_numpy(rays).cuda()
    for any_hit in (False, True):
        sg, so = A.pt_stats(), A.pt_stats()
        prim, t, b1, b2 = ctx.intersect(rt, any_hit=any_hit, stats=sg)
        torch.cuda.synchronize()
        op, ot, ob1, ob2 = osc.intersect(rays, any_hit=any_hit, stats=so)
        assert np.array_equal(prim.cpu().numpy(), op)
        if not any_hit:
            assert np.array_equal(t.cpu().numpy(), ot) and np.array_equal(b1.cpu().numpy(), ob1)
            hits = int((op >= 0).sum())
            assert hits > n // 20
            closest_prim, closest_t = op, ot
        else:
            assert np.array_equal(op > 0, closest_prim >= 0)               # any hit <=> a closest hit exists
        assert sg.as_dict() == so.as_dict()
    # (b) brute force over ALL triangles for a few rays: the tree loses nothing
    tv = flat.tri_verts.reshape(-1, 3, 3)
    p1, e1, e2 = tv[:, 0], tv[:, 1] - tv[:, 0], tv[:, 2] - tv[:, 0]
    for i in np.nonzero(closest_prim >= 0)[0][:6]:
        oo, dd = rays[0:3, i], rays[3:6, i]
        s1 = np.cross(dd, e2)
        div = (s1 * e1).sum(1)
        ok = div != 0
        inv = np.where(ok, 1. / np.where(ok, div, 1.), 0.)
        dv = oo - p1
        bb1 = (dv * s1).sum(1) * inv
        s2 = np.cross(dv, e1)
        bb2 = (dd * s2).sum(1) * inv
        tt = (e2 * s2).sum(1) * inv
        ok &= (bb1 >= 0) & (bb1 <= 1) & (bb2 >= 0) & (bb1 + bb2 <= 1) & (tt >= rays[6, i]) & (tt <= rays[7, i])
        assert ok.any() and abs(tt[ok].min() - closest_t[i]) <= 1e-12 * max(1., abs(closest_t[i]))
    # (c) the render: census identities and tile additivity (the multi-GPU split of configs[3])
    res = (480, 270)
    st = A.pt_stats()
    full, L, _ = _render(ctx, flat, res, 2, A.PT_INTEG_PATH, stats=st)
    assert st.camera_samples == res[0] * res[1] * 4 and st.guard_neg == 0
    assert st.guard_inf > 0                       # spot light outside its cone: I/0 (Q9) → guarded samples
    assert np.isfinite(full).all() and (L >= 0).all()
    parts = np.zeros_like(full)
    for k in range(4):                            # 4 row bands
        t_ = pa.tile_desc(0, res[0], (res[1] * k) // 4, (res[1] * (k + 1)) // 4, 0, 4)
        a, _, _ = _render(ctx, flat, res, 2, A.PT_INTEG_PATH, tile=t_)
        parts += a
    np.testing.assert_allclose(parts, full, rtol=1e-11, atol=1e-11)
    # (d) one window of that render against the oracle
    wt = pa.tile_desc(200, 216, 120, 132, 0, 4)
    _, Lw, (cam, film, smp, _) = _render(ctx, flat, res, 2, A.PT_INTEG_PATH, tile=wt)
    _, oL = osc.render(cam, film, smp, pa.integrator_desc(A.PT_INTEG_PATH, 5, False), wt)
    assert rel_err(Lw, oL) < RTOL


@pytest.mark.gpu
def test_c4_full_frame_through_two_wavefront_batches(ctx):
    """configs[3] at its real resolution through more than one LARGE wavefront batch: 1920 x 1080 x 36 spp = 74.6 M
    paths > 2^26, so with PT_BATCH = 2^26 (the default until it became 2^28; what configs[3] at 256 spp runs in is
    two batches of 2^28) pt_render runs two batches, the first of 67 M paths — tests elsewhere force small batches;
    k_trace_pair_refill and k_trace_packet on the 2 M-node tree.  Census identities, and the four row
    bands of the frame (one batch each) add up to the two-batch frame with an equal census."""
    import torch
    flat = W.c4_soup(10 ** 6)
    ctx.upload(flat)
    res, xs = (1920, 1080), 6
    cam, film = W.c4_camera_film(*res)
    smp = pa.sampler_desc(0, res[0], 0, res[1], xs, xs, True, 3)
    integ = pa.integrator_desc(A.PT_INTEG_PATH, 5, False)

    def render(tiles, log=False):
        acc = torch.zeros((film.y_cnt, film.x_cnt, 4), dtype=torch.float64, device='cuda')
        st = A.pt_stats()
        batches = set()
        for t in tiles:
            ctx.render(cam, film, smp, integ, t, acc, stats=st)
            if log:
                batches |= {(r['batch'], r['kernel']) for r in ctx.launch_log() if r['stage'].startswith('trace')}
        torch.cuda.synchronize()
        return acc, st.as_dict(), batches
    ctx.set_profiling(1)
    os.environ['PT_BATCH'] = str(1 << 26)            # read by every pt_render call
    try:
        full, s1, batches = render([pa.tile_desc(0, res[0], 0, res[1], 0, xs * xs)], log=True)
    finally:
        del os.environ['PT_BATCH']
        ctx.set_profiling(0)
    n = res[0] * res[1] * xs * xs
    assert n > (1 << 26) and {b for b, _ in batches} == {0, 1}
    assert {k for _, k in batches} == {'k_trace_packet', 'k_trace_pair_refill'}
    assert s1['camera_samples'] == n and s1['rays_closest'] >= n and s1['guard_neg'] == 0 and s1['guard_inf'] > 0
    assert s1['node_visits'] > 100 * (s1['rays_closest'] + s1['rays_any'])      # ~160 visits per ray on this tree
    assert bool(torch.isfinite(full).all())
    bands = [pa.tile_desc(0, res[0], (res[1] * k) // 4, (res[1] * (k + 1)) // 4, 0, xs * xs) for k in range(4)]
    parts, s4, _ = render(bands)
    assert s4 == s1
    scale = float(full.abs().max())
    assert float((parts - full).abs().max()) <= 1e-11 * scale
    # the default: ONE batch of 74.6 M paths (queue indices beyond 2^26; the cap is 2^28 paths within half of the free HBM)
    ctx.set_profiling(1)
    try:
        one, s0, batches = render([pa.tile_desc(0, res[0], 0, res[1], 0, xs * xs)], log=True)
    finally:
        ctx.set_profiling(0)
    assert {b for b, _ in batches} == {0}
    assert s0 == s1
    assert float((one - full).abs().max()) <= 1e-11 * scale
