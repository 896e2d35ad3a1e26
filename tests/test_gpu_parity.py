"""GPU parity tests proper: the HIP path, called through the C ABI, against (a) the CPU oracle
on identical inputs and (b) the committed reference outputs (tests/golden).

Bars (BASELINE.json north_star): hit/miss and primitive ids bit-exact; radiance / film within
1e-4 relative of the float64 reference.  The kernels compute in binary64 with contraction off,
so traversal outputs are compared for bitwise equality with the oracle and shading outputs at
1e-9 (device libm differs from glibc in the last ulps of pow/sin/cos)."""
import os

import numpy as np
import pytest

import pytracer_amd as pa
from pytracer_amd import _abi as A

pytestmark = pytest.mark.gpu

REF_RTOL = 1e-4      # stated fp tolerance vs the reference
ORACLE_RTOL = 1e-9   # what the f64 kernels actually have to meet vs the oracle


@pytest.fixture(scope='module')
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), 'these tests need the MI355X'
    return torch


@pytest.fixture(scope='module')
def ctx(torch_cuda):
    from pytracer_amd._lib import Context
    c = Context(0)
    yield c
    c.close()


def load(golden_dir, name):
    return dict(np.load(os.path.join(golden_dir, name)))


@pytest.fixture(scope='module')
def head(golden_dir, ctx):
    import oracle_lib as O
    d = load(golden_dir, 'head_mesh_scene.npz')
    flat = pa.flat_scene_from_npz(d)
    return d, flat, O.OracleScene(flat)


def rays_from_log(log):
    r = np.empty((8, len(log)))
    r[0:6] = log[:, 2:8].T
    r[6] = log[:, 8]
    r[7] = log[:, 9]
    return r


def gpu_intersect(torch, ctx, rays, any_hit, stats=None):
    rt = torch.from_numpy(np.ascontiguousarray(rays)).cuda()
    prim, t, b1, b2 = ctx.intersect(rt, any_hit=any_hit, stats=stats)
    torch.cuda.synchronize()
    f = lambda x: None if x is None else x.cpu().numpy()
    return f(prim), f(t), f(b1), f(b2)


@pytest.mark.parametrize('name', ['head_direct_24x24x4.npz', 'head_path_32x32x4.npz', 'head_c1_window.npz'])
def test_intersect_logged_rays(torch_cuda, ctx, golden_dir, head, name):
    """Every ray the reference traced: hit flag + primitive id equal the REFERENCE's, and all four
    outputs are bitwise equal to the oracle's."""
    _, flat, osc = head
    ctx.upload(flat)
    d = load(golden_dir, name)
    log, res = d['log_rays'], d['log_rres']
    kind = log[:, 1].astype(int)
    rays = rays_from_log(log)
    for any_hit in (0, 1):
        sel = np.nonzero(kind == any_hit)[0]
        so, sg = A.pt_stats(), A.pt_stats()
        op, ot, ob1, ob2 = osc.intersect(rays[:, sel], any_hit=bool(any_hit), stats=so)
        gp, gt, gb1, gb2 = gpu_intersect(torch_cuda, ctx, rays[:, sel], bool(any_hit), stats=sg)
        assert np.array_equal(gp, op)
        if any_hit:
            assert np.array_equal(gp, res[sel, 0].astype(np.int32))
        else:
            assert np.array_equal(gp, res[sel, 1].astype(np.int32))
            assert np.array_equal(gt, ot) and np.array_equal(gb1, ob1) and np.array_equal(gb2, ob2)
            np.testing.assert_allclose(gt, res[sel, 2], rtol=1e-12)
        # the kernel-counted traversal work equals the oracle's (the roofline's N_node, N_tri)
        assert sg.node_visits == so.node_visits and sg.tri_tests == so.tri_tests


@pytest.mark.parametrize('name', ['bvh_soup300.npz', 'bvh_soup2000.npz', 'bvh_soup500dup.npz'])
def test_intersect_soups(torch_cuda, ctx, golden_dir, name):
    import oracle_lib as O
    d = load(golden_dir, name)
    flat = pa.flat_scene_from_npz(d)
    ctx.upload(flat)
    rays = np.ascontiguousarray(d['rays'].T)
    gp, gt, _, _ = gpu_intersect(torch_cuda, ctx, rays, False)
    assert np.array_equal(gp, d['res'][:, 1].astype(np.int32))
    np.testing.assert_allclose(gt, d['res'][:, 2], rtol=1e-12)
    ga, _, _, _ = gpu_intersect(torch_cuda, ctx, rays, True)
    assert np.array_equal(ga, d['res'][:, 3].astype(np.int32))
    op, ot, _, _ = O.OracleScene(flat).intersect(rays, any_hit=False)
    assert np.array_equal(gp, op) and np.array_equal(gt, ot)


def test_axis_parallel_rays_on_box_planes(torch_cuda, ctx, golden_dir):
    """Rays with zero and -0.0 direction components whose origins lie exactly on node box planes (NaN slab
    distances, inv = -inf with dir_is_neg false): the fast slab test must not be used for them — results equal
    the REFERENCE's, outputs and the node/triangle census are bitwise the oracle's."""
    import oracle_lib as O
    d = load(golden_dir, 'bvh_axis_rays.npz')
    flat = pa.flat_scene_from_npz(d)
    ctx.upload(flat)
    osc = O.OracleScene(flat)
    rays = np.ascontiguousarray(d['rays'].T)
    for any_hit in (False, True):
        gs, os_ = A.pt_stats(), A.pt_stats()
        gp, gt, gb1, gb2 = gpu_intersect(torch_cuda, ctx, rays, any_hit, stats=gs)
        op, ot, ob1, ob2 = osc.intersect(rays, any_hit=any_hit, stats=os_)
        assert np.array_equal(gp, op)
        assert (gs.node_visits, gs.tri_tests) == (os_.node_visits, os_.tri_tests)
        if any_hit:
            assert np.array_equal(gp, d['res'][:, 3].astype(np.int32))
        else:
            assert np.array_equal(gt, ot) and np.array_equal(gb1, ob1) and np.array_equal(gb2, ob2)
            assert np.array_equal(gp, d['res'][:, 1].astype(np.int32))
            hit = gp >= 0
            np.testing.assert_allclose(gt[hit], d['res'][hit, 2], rtol=1e-12)
    # the same rays nudged off the planes and off zero take the fast form: still the oracle's, census included
    r2 = rays.copy()
    r2[3:6] += 1e-9
    gs, os_ = A.pt_stats(), A.pt_stats()
    gp, gt, _, _ = gpu_intersect(torch_cuda, ctx, r2, False, stats=gs)
    op, ot, _, _ = osc.intersect(r2, any_hit=False, stats=os_)
    assert np.array_equal(gp, op) and np.array_equal(gt, ot)
    assert (gs.node_visits, gs.tri_tests) == (os_.node_visits, os_.tri_tests)


@pytest.mark.parametrize('knob', ['PT_TRACE_TOP=1', 'PT_TRACE_META=4', 'PT_TRACE_META=4,PT_TRACE_META_SHORT=0', 'PT_TRACE_META=4,PT_TRACE_META_SHORT=1', 'PT_TRACE_META=3', 'PT_TRACE_META=0', 'PT_TRACE_PAIR=3',
                                  'PT_TRACE_DYNAMIC=1', 'PT_TRACE_DYNAMIC=0', 'PT_TRACE_META=4,PT_TRACE_DYNAMIC=0',
                                  'PT_TRACE_PACKET=3', 'PT_TRACE_PACKET=3,PT_OVERLAP=0',
                                  'PT_TRACE_PAIR=3,PT_PAIR_REFILL=1,PT_REFILL_MIN=5',
                                  'PT_TRACE_PAIR=3,PT_PAIR_REFILL=1,PT_REFILL_MIN=5,PT_PAIR_HALF=0',
                                  'PT_TRACE_PAIR=3,PT_PAIR_REFILL=1,PT_REFILL_MIN=5,PT_OVERLAP=1,PT_TRACE_PACKET=0',
                                  'PT_TRACE_PAIR=3,PT_PAIR_REFILL=1,PT_TRACE_PACKET=3,PT_OVERLAP=1'])
def test_trace_kernel_variants_are_bitwise_the_same(torch_cuda, golden_dir, head, knob):
    """k_trace_top (first breadth-first nodes staged in LDS; default only for trees beyond the L2), k_trace_meta
    (child/primitive words of all nodes in LDS; default for closest-hit launches on small trees), k_trace_pair (the
    walk fetched by parent: both child boxes in one 128-byte record, the far child's verdict finished at its pop)
    and the plain k_trace walk the same nodes in the same order: outputs and visit counts equal the oracle's.
    The forms that only render launches can take — k_trace_packet (one walk per wave, the level-0 launches of the
    headline render), k_trace_pair_refill (lane refill, the 10^6-triangle soup's default) and the two-stream
    schedule — are forced on by their knobs and held to the REFERENCE's per-sample radiance and the oracle's census
    in the whole-render leg, on three fixture scenes (head, c4mini: three light kinds, specmini: specular chains)."""
    import oracle_lib as O
    from pytracer_amd._lib import Context
    kv = dict(x.split('=') for x in knob.split(','))
    os.environ.update(kv)
    try:
        c = Context(0)
    finally:
        for k in kv:
            del os.environ[k]
    try:
        cases = [(head[1], head[2], rays_from_log(load(golden_dir, 'head_path_32x32x4.npz')['log_rays']))]
        for name in ('bvh_soup2000.npz', 'bvh_soup500dup.npz'):
            d = load(golden_dir, name)
            flat = pa.flat_scene_from_npz(d)
            cases.append((flat, O.OracleScene(flat), np.ascontiguousarray(d['rays'].T)))
        for flat, osc, rays in cases:
            c.upload(flat)
            for any_hit in (False, True):
                sg, so = A.pt_stats(), A.pt_stats()
                gp, gt, gb1, gb2 = gpu_intersect(torch_cuda, c, rays, any_hit, stats=sg)
                op, ot, ob1, ob2 = osc.intersect(rays, any_hit=any_hit, stats=so)
                assert np.array_equal(gp, op)
                if not any_hit:
                    assert np.array_equal(gt, ot) and np.array_equal(gb1, ob1) and np.array_equal(gb2, ob2)
                assert sg.as_dict() == so.as_dict()
        # and through whole renders: the reference's own per-sample radiance, the oracle's radiance and census
        for name in ('head_path_32x32x4.npz', 'c4mini_path_24x24x4.npz', 'specmini_path_24x24x4.npz'):
            d = load(golden_dir, name)
            flat, osc = (head[1], head[2]) if name.startswith('head') else (pa.flat_scene_from_npz(d), None)
            if osc is None:
                osc = O.OracleScene(flat)
            c.upload(flat)
            integ = pa.integrator_desc(A.PT_INTEG_PATH, 5, True)
            sg, so = A.pt_stats(), A.pt_stats()
            acc, L, (cam, film, smp, tile) = gpu_render(torch_cuda, c, d, d['sub'], d['spp'], integ, table=d['samples'],
                                                        stats=sg)
            assert rel_err(L, d['L']) < REF_RTOL, (knob, name)
            np.testing.assert_allclose(acc[..., 0:3], d['film_Lxyz'], rtol=REF_RTOL, atol=1e-9)
            oacc, oL = osc.render(cam, film, smp, integ, tile, table=d['samples'], stats=so)
            assert rel_err(L, oL) < ORACLE_RTOL, (knob, name)
            assert_stats_close(sg, so)
    finally:
        c.close()


@pytest.mark.parametrize('kind', [A.PT_INTEG_PATH, A.PT_INTEG_DIRECT])
def test_dynamic_chunk_assignment_changes_nothing(torch_cuda, golden_dir, head, kind):
    """PT_TRACE_DYNAMIC / PT_SHADE_DYNAMIC: the waves of a queue region draw their 64-ray chunks from a cursor instead
    of striding over the region; PT_PAIR_REFILL: the lanes of k_trace_pair take new rays from that cursor as they finish
    (k_trace_pair_refill); PT_TRACE_PACKET: the level-0 launches walk the tree once per wave (k_trace_packet).  Which wave or
    lane traces a ray, or how its node records are fetched, is not observable: per-sample radiance equal to the last bits (the shadow
    contributions are atomic adds whose order is free in both forms), census equal, on a production-stream render large
    enough that every region holds many chunks, whole and as an interleaved shard."""
    from pytracer_amd._lib import Context
    from pytracer_amd.renderer import shard_tile
    R = 512                                  # 1 M samples: 16 k chunks for 4 096 waves
    sub, spp = (0, R, 0, R), (2, 2)
    settings = [{'PT_TRACE_DYNAMIC': '0', 'PT_SHADE_DYNAMIC': '0'}, {},
                {'PT_TRACE_PAIR': '3', 'PT_PAIR_REFILL': '0'}, {'PT_TRACE_PAIR': '3', 'PT_PAIR_REFILL': '1'},
                {'PT_TRACE_PAIR': '3', 'PT_PAIR_REFILL': '1', 'PT_PAIR_HALF': '0'},
                {'PT_TRACE_PACKET': '3'}, {'PT_TRACE_PACKET': '0'}]
    out = []
    for env in settings:
        env = dict(env, PT_OVERLAP='0')         # one stream: the order of the adds into L is then the launch order
        os.environ.update(env)
        try:
            c = Context(0)
        finally:
            for k in env:
                del os.environ[k]
        try:
            c.upload(head[1])
            res = []
            smp = pa.sampler_desc(*sub, spp[0], spp[1], True, 0)
            for tile in (None, shard_tile(smp, 1, 3, 'interleaved')):
                st = A.pt_stats()
                cam, film = host_camera_film(R)
                n = R * (R if tile is None else tile.rows()) * 4
                t = tile if tile is not None else pa.tile_desc(*sub, 0, 4)
                acc = torch_cuda.zeros((film.y_cnt, film.x_cnt, 4), dtype=torch_cuda.float64, device='cuda')
                L = torch_cuda.zeros((n, 3), dtype=torch_cuda.float64, device='cuda')
                c.render(cam, film, smp, pa.integrator_desc(kind, 5, False), t, acc, sample_L=L, stats=st)
                torch_cuda.cuda.synchronize()
                res.append((acc.cpu().numpy(), L.cpu().numpy(), st.as_dict()))
            out.append(res)
        finally:
            c.close()
    assert out[0][0][2]['rays_closest'] >= R * R * 4
    for other in out[1:]:
        for (a0, l0, s0), (a1, l1, s1) in zip(out[0], other):
            assert s0 == s1
            np.testing.assert_allclose(l1, l0, rtol=1e-13, atol=0)
            np.testing.assert_allclose(a1, a0, rtol=1e-11, atol=1e-14)


def host_camera_film(res):
    from pytracer_amd import host
    return host.head_mesh_camera_film(res, res)


def test_intersect_edge_cases(torch_cuda, ctx, golden_dir, head):
    """Empty batch, a single ray, a ragged (non multiple of 64/256) batch, zero-length maxt."""
    _, flat, osc = head
    ctx.upload(flat)
    d = load(golden_dir, 'head_direct_24x24x4.npz')
    rays = rays_from_log(d['log_rays'])
    for n in (0, 1, 63, 65, 257, 1000):
        r = rays[:, :n]
        gp, gt, _, _ = gpu_intersect(torch_cuda, ctx, r, False)
        if n:
            op, ot, _, _ = osc.intersect(r, any_hit=False)
            assert np.array_equal(gp, op) and np.array_equal(gt, ot)
        else:
            assert len(gp) == 0
    r = rays[:, :128].copy()
    r[7] = 0.
    gp, _, _, _ = gpu_intersect(torch_cuda, ctx, r, False)
    assert (gp == -1).all()


def test_intersect_million_camera_rays(torch_cuda, ctx, golden_dir, head):
    """BASELINE-size batch (1024x1024 camera rays of the head-mesh view): bitwise equal to the
    oracle on a 1-in-7 subsample, and hit counts consistent over the full batch."""
    d, flat, osc = head
    ctx.upload(flat)
    n = 1024
    ys, xs = np.mgrid[0:n, 0:n]
    # camera rays of configure_camera at 1024^2 (r2c scales with resolution: x/128 → x/1024)
    r2c = d['cam_r2c'].copy()
    r2c[0, 0] *= 128. / n
    r2c[1, 1] *= 128. / n
    pr = np.stack([xs.ravel() + .5, ys.ravel() + .5, np.zeros(n * n), np.ones(n * n)])
    pc = r2c @ pr
    pc = pc[:3] / pc[3]
    dirs = pc / np.linalg.norm(pc, axis=0)
    rays = np.empty((8, n * n))
    rays[0:3] = d['cam_c2w'][:3, 3:4]
    rays[3:6] = d['cam_c2w'][:3, :3] @ dirs
    rays[6] = 0.
    rays[7] = np.inf
    gp, gt, gb1, gb2 = gpu_intersect(torch_cuda, ctx, rays, False)
    ga, _, _, _ = gpu_intersect(torch_cuda, ctx, rays, True)
    assert np.array_equal(ga == 1, gp >= 0)          # any-hit flag == closest-hit flag (renderer.py:168 vs :64)
    # the backdrop fills the view, the head covers part of it (two materials → two prim groups)
    assert (gp >= 0).mean() > 0.99
    head_mat = int(np.argmax([len(l) for l in flat.material_lobes]))
    frac_head = (flat.tri_mat[gp[gp >= 0]] == head_mat).mean()
    assert 0.05 < frac_head < 0.95
    sub = np.arange(0, n * n, 7)
    op, ot, ob1, ob2 = osc.intersect(rays[:, sub], any_hit=False)
    assert np.array_equal(gp[sub], op) and np.array_equal(gt[sub], ot)
    assert np.array_equal(gb1[sub], ob1) and np.array_equal(gb2[sub], ob2)


def gpu_render(torch, ctx, d_cam, sub, spp, integ, table=None, seed=0, tile=None, jitter=True, stats=None):
    cam, film = pa.camera_film_from_npz(d_cam)
    smp = pa.sampler_desc(sub[0], sub[1], sub[2], sub[3], spp[0], spp[1], jitter, seed)
    if tile is None:
        tile = pa.tile_desc(sub[0], sub[1], sub[2], sub[3], 0, int(spp[0] * spp[1]))
    n = (tile.x1 - tile.x0) * (tile.y1 - tile.y0) * (tile.s1 - tile.s0)
    acc = torch.zeros((film.y_cnt, film.x_cnt, 4), dtype=torch.float64, device='cuda')
    L = torch.zeros((n, 3), dtype=torch.float64, device='cuda')
    tt = None if table is None else torch.from_numpy(np.ascontiguousarray(table)).cuda()
    ctx.render(cam, film, smp, integ, tile, acc, table=tt, sample_L=L, stats=stats)
    torch.cuda.synchronize()
    return acc.cpu().numpy(), L.cpu().numpy(), (cam, film, smp, tile)


def assert_stats_close(sg, so):
    """Render-level census: ray counts and guards equal; traversal work within 1e-3 (device libm
    differs from glibc in the last ulp of pow/sin/cos, so a secondary ray may test one triangle
    more or fewer — pt_intersect on identical rays is compared exactly elsewhere)."""
    a, b = sg.as_dict(), so.as_dict()
    for k in ('camera_samples', 'rays_closest', 'rays_any', 'guard_nan', 'guard_neg', 'guard_inf'):
        assert a[k] == b[k], (k, a, b)
    for k in ('node_visits', 'tri_tests'):
        assert abs(a[k] - b[k]) <= 1e-3 * max(b[k], 1), (k, a, b)


def rel_err(a, ref):
    scale = np.maximum(np.abs(ref).max(axis=-1, keepdims=True), 1e-12)
    return (np.abs(a - ref) / scale).max()


@pytest.mark.parametrize('name,kind', [('head_direct_40x40x1.npz', A.PT_INTEG_DIRECT),
                                       ('head_direct_24x24x4.npz', A.PT_INTEG_DIRECT),
                                       ('head_c1_window.npz', A.PT_INTEG_DIRECT),
                                       ('head_path_32x32x4.npz', A.PT_INTEG_PATH)])
def test_render_parity_tables(torch_cuda, ctx, golden_dir, head, name, kind):
    """Parity mode (captured sample stream + injected counter rng): per-sample radiance and film
    accumulators vs the REFERENCE's outputs (1e-4 bar) and vs the oracle (1e-9)."""
    _, flat, osc = head
    ctx.upload(flat)
    d = load(golden_dir, name)
    integ = pa.integrator_desc(kind, 5, pretest=True)
    sg = A.pt_stats()
    acc, L, (cam, film, smp, tile) = gpu_render(torch_cuda, ctx, d, d['sub'], d['spp'], integ, table=d['samples'],
                                                stats=sg)
    assert rel_err(L, d['L']) < REF_RTOL
    np.testing.assert_allclose(acc[..., 0:3], d['film_Lxyz'], rtol=REF_RTOL, atol=1e-9)
    np.testing.assert_allclose(acc[..., 3], d['film_wsum'], rtol=REF_RTOL, atol=1e-9)
    so = A.pt_stats()
    oacc, oL = osc.render(cam, film, smp, integ, tile, table=d['samples'], stats=so)
    assert rel_err(L, oL) < ORACLE_RTOL
    np.testing.assert_allclose(acc, oacc, rtol=ORACLE_RTOL, atol=1e-12)
    assert_stats_close(sg, so)
    # resolved image
    rgb = ctx.film_resolve(torch_cuda.from_numpy(acc).cuda()).cpu().numpy()
    import oracle_lib as O
    np.testing.assert_allclose(rgb, O.film_resolve(oacc), rtol=ORACLE_RTOL, atol=1e-12)


@pytest.mark.parametrize('kind,spp', [(A.PT_INTEG_DIRECT, (2, 2)), (A.PT_INTEG_PATH, (2, 2)),
                                      (A.PT_INTEG_PATH, (1, 1))])
def test_render_production_stream_vs_oracle(torch_cuda, ctx, golden_dir, head, kind, spp):
    """Production mode (in-kernel counter-based sample stream, no tables): the oracle restates the
    same stream, so the whole pipeline is compared at a size the oracle renders in seconds."""
    d, flat, osc = head
    ctx.upload(flat)
    dd = dict(d)
    res = 96
    r2c = d['cam_r2c'].copy(); r2c[0, 0] *= 128. / res; r2c[1, 1] *= 128. / res
    dd['cam_r2c'] = r2c
    dd['film_res'] = np.array([res, res]); dd['film_win'] = np.array([0, res, 0, res])
    integ = pa.integrator_desc(kind, 5, pretest=False)
    sg, so = A.pt_stats(), A.pt_stats()
    acc, L, (cam, film, smp, tile) = gpu_render(torch_cuda, ctx, dd, (0, res, 0, res), spp, integ, seed=7, stats=sg)
    oacc, oL = osc.render(cam, film, smp, integ, tile, stats=so)
    assert (np.abs(oL).sum(axis=1) > 0).mean() > 0.3
    assert rel_err(L, oL) < ORACLE_RTOL
    np.testing.assert_allclose(acc, oacc, rtol=ORACLE_RTOL, atol=1e-12)
    assert_stats_close(sg, so)


def test_render_tile_and_sample_sharding_adds_up(torch_cuda, ctx, golden_dir, head):
    """The multi-GPU decomposition: tiles × sample-index ranges rendered separately and summed give
    the film of the whole window (additivity; atomics make the order free, so 1e-12)."""
    d, flat, _ = head
    ctx.upload(flat)
    integ = pa.integrator_desc(A.PT_INTEG_PATH, 5, pretest=False)
    sub, spp = (0, 128, 0, 128), (2, 2)
    whole, _, _ = gpu_render(torch_cuda, ctx, d, sub, spp, integ, seed=3)
    parts = np.zeros_like(whole)
    for (y0, y1) in ((0, 50), (50, 128)):
        for (s0, s1) in ((0, 1), (1, 4)):
            a, _, _ = gpu_render(torch_cuda, ctx, d, sub, spp, integ, seed=3,
                                 tile=pa.tile_desc(0, 128, y0, y1, s0, s1))
            parts += a
    np.testing.assert_allclose(parts, whole, rtol=1e-11, atol=1e-12)


def test_render_full_size_properties(torch_cuda, ctx, golden_dir, head):
    """BASELINE size (1024^2; 4 spp to keep the run short): properties that need no oracle.
    (1) weight_sum of an unjittered render equals the closed-form Lanczos footprint sum;
    (2) the pre-test flag changes ray counts but not a single film value;
    (3) batching (PT_BATCH) does not change the film."""
    d, flat, _ = head
    ctx.upload(flat)
    res = 1024
    dd = dict(d)
    r2c = d['cam_r2c'].copy(); r2c[0, 0] *= 128. / res; r2c[1, 1] *= 128. / res
    dd['cam_r2c'] = r2c
    dd['film_res'] = np.array([res, res]); dd['film_win'] = np.array([0, res, 0, res])
    integ = pa.integrator_desc(A.PT_INTEG_DIRECT, 5, pretest=False)
    s0 = A.pt_stats()
    acc, L, _ = gpu_render(torch_cuda, ctx, dd, (0, res, 0, res), (2, 2), integ, jitter=False, stats=s0)
    assert s0.camera_samples == res * res * 4
    # closed form: samples at pixel + (.25|.75); per-axis weights from the 16x16 table are separable
    # only through the table lookup, so evaluate the footprint sum per axis offset directly
    tab = d['filt_table']
    w = np.zeros((res, res))
    for sy in (.25, .75):
        for sx in (.25, .75):
            for j in range(-5, 6):
                for i in range(-5, 6):
                    # contribution of the sample in pixel (x-i, y-j) to pixel (x, y)
                    dX, dY = -i + sx - .5, -j + sy - .5      # sample position relative to the target pixel
                    if not (np.ceil(dX - 4.) <= 0 <= np.floor(dX + 4.)):
                        continue
                    if not (np.ceil(dY - 4.) <= 0 <= np.floor(dY + 4.)):
                        continue
                    ifx = min(15, int(np.floor(abs((0 - dX) * 16 * .25))))
                    ify = min(15, int(np.floor(abs((0 - dY) * 16 * .25))))
                    ys = slice(max(0, j), res + min(0, j))
                    xs = slice(max(0, i), res + min(0, i))
                    w[ys, xs] += tab[ify, ifx]
    np.testing.assert_allclose(acc[..., 3], w, rtol=1e-11, atol=1e-12)
    s1 = A.pt_stats()
    acc1, _, _ = gpu_render(torch_cuda, ctx, dd, (0, res, 0, res), (2, 2), pa.integrator_desc(A.PT_INTEG_DIRECT, 5, True),
                            jitter=False, stats=s1)
    np.testing.assert_allclose(acc1, acc, rtol=1e-11, atol=1e-12)
    assert s1.rays_any == s0.rays_any + s0.camera_samples
    os.environ['PT_BATCH'] = str(300000)
    try:
        acc2, L2, _ = gpu_render(torch_cuda, ctx, dd, (0, res, 0, res), (2, 2), integ, jitter=False)
    finally:
        del os.environ['PT_BATCH']
    assert np.array_equal(L2, L)
    np.testing.assert_allclose(acc2, acc, rtol=1e-11, atol=1e-12)


def test_errors_are_reported(torch_cuda, golden_dir):
    from pytracer_amd._lib import Context, PtError
    c = Context(0)
    try:
        r = torch_cuda.zeros((8, 4), dtype=torch_cuda.float64, device='cuda')
        with pytest.raises(PtError) as e:
            c.intersect(r)
        assert e.value.code == A.PT_ERR_NO_SCENE
        d = load(golden_dir, 'bvh_soup300.npz')
        flat = pa.flat_scene_from_npz(d)
        c.upload(flat)
        cam, film = pa.camera_film_from_npz(load(golden_dir, 'head_direct_40x40x1.npz'))
        smp = pa.sampler_desc(0, 8, 0, 8, 1, 1)
        acc = torch_cuda.zeros((film.y_cnt, film.x_cnt, 4), dtype=torch_cuda.float64, device='cuda')
        with pytest.raises(PtError) as e:     # tile outside the sampler window
            c.render(cam, film, smp, pa.integrator_desc(A.PT_INTEG_PATH), pa.tile_desc(0, 9, 0, 8, 0, 1), acc)
        assert e.value.code == A.PT_ERR_ARG
    finally:
        c.close()


@pytest.mark.parametrize('short', ['2', '1', '0'])
def test_meta_walk_on_a_tree_with_a_leaf_too_large_to_pack(torch_cuda, golden_dir, short):
    """The C ABI takes any linearised tree.  A leaf of 1 500 primitives does not fit the 10-bit count of the packed node
    word of k_trace_meta_s<.., 256, true> (pt_scene_upload checks every node): the launch falls back to the 512-thread
    form with 8-byte node words (PT_TRACE_META_SHORT=2, the default, and =1) or runs the 1024-thread form (=0); hits and
    census equal the oracle's on the same tree."""
    import oracle_lib as O
    from pytracer_amd._lib import Context
    d = load(golden_dir, 'bvh_soup2000.npz')
    base = pa.flat_scene_from_npz(d)
    tv = base.tri_verts
    assert len(tv) == 2000

    def box(a):
        p = a.reshape(-1, 3)
        return np.concatenate([p.min(0), p.max(0)])
    bounds = np.stack([box(tv), box(tv[:1500]), box(tv[1500:])])
    meta = np.array([[2, 0, 0], [0, 1500, 0], [1500, 500, 0]], np.int32)
    flat = pa.FlatScene(bounds, meta, tv, base.tri_mat, base.tri_flip, base.material_lobes, base.light_records)
    rays = np.ascontiguousarray(d['rays'].T)[:, :1024]
    env = {'PT_TRACE_META': '4', 'PT_TRACE_META_SHORT': short}
    os.environ.update(env)
    try:
        c = Context(0)
    finally:
        for k in env:
            del os.environ[k]
    try:
        c.upload(flat)
        osc = O.OracleScene(flat)
        for any_hit in (False, True):
            gs, os_ = A.pt_stats(), A.pt_stats()
            g = gpu_intersect(torch_cuda, c, rays, any_hit, gs)
            o = osc.intersect(rays, any_hit=any_hit, stats=os_)
            assert np.array_equal(g[0], o[0])
            if not any_hit:
                assert (g[0] >= 0).sum() > 100
                assert np.array_equal(g[1], o[1])
            assert (gs.node_visits, gs.tri_tests) == (os_.node_visits, os_.tri_tests)
    finally:
        c.close()
