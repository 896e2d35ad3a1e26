"""C4 in miniature (SURVEY.md §8 rows a9 Plastic/Matte-OrenNayar, a13 SpotLight + PointLight, a12 sphere
area light together, a14 with several lights): a triangle soup in front of a wall, rendered by the
REFERENCE with Direct and Path (tools/oracle/gen_fixtures.py:gen_c4mini →
tests/golden/c4mini_*.npz).  CPU part: oracle vs the reference outputs.  GPU part: HIP path vs the
reference outputs (1e-4) and vs the oracle (1e-9); the same tree is also rebuilt by the native
builder from the fixture's input triangles."""
import os

import numpy as np
import pytest

import pytracer_amd as pa
from pytracer_amd import _abi as A, accel
import oracle_lib as O

RTOL = 1e-9
CASES = [('c4mini_direct_24x24x1.npz', A.PT_INTEG_DIRECT), ('c4mini_path_24x24x4.npz', A.PT_INTEG_PATH)]


def load(golden_dir, name):
    return dict(np.load(os.path.join(golden_dir, name)))


def rays_from_log(log):
    r = np.empty((8, len(log)))
    r[0:6] = log[:, 2:8].T
    r[6] = log[:, 8]
    r[7] = log[:, 9]
    return r


def rel_err(a, ref):
    scale = np.maximum(np.abs(ref).max(axis=-1, keepdims=True), 1e-12)
    return (np.abs(a - ref) / scale).max()


def descs(d):
    cam, film = pa.camera_film_from_npz(d)
    sub = d['sub']
    smp = pa.sampler_desc(sub[0], sub[1], sub[2], sub[3], d['spp'][0], d['spp'][1], True, 0)
    tile = pa.tile_desc(sub[0], sub[1], sub[2], sub[3], 0, int(d['spp'][0] * d['spp'][1]))
    return cam, film, smp, tile


def test_scene_has_the_c4_ingredients(golden_dir):
    d = load(golden_dir, CASES[0][0])
    flat = pa.flat_scene_from_npz(d)
    kinds = sorted(int(d['light%d_kind' % i]) for i in range(flat.n_lights))
    assert kinds == [A.PT_LIGHT_DIFFUSE_SPHERE, A.PT_LIGHT_SPOT, A.PT_LIGHT_POINT]
    lobes = [[l[0] for l in m] for m in flat.material_lobes]
    assert [A.PT_LOBE_LAMBERT] in lobes and [A.PT_LOBE_LAMBERT, A.PT_LOBE_MICROFACET_BLINN] in lobes
    assert [A.PT_LOBE_OREN_NAYAR] in lobes


def test_native_builder_reproduces_the_fixture_tree(golden_dir):
    d = load(golden_dir, CASES[0][0])
    # BVH([mesh0, mesh1, wall]): each mesh's triangles refined in reverse (primitive.py:68-75)
    tv = np.concatenate([d['in_tris%d' % k][::-1] for k in range(3)])
    mat = np.concatenate([np.full(len(d['in_tris%d' % k]), k, np.int32) for k in range(3)])
    nb, nm, order, _ = accel.build_bvh(accel.triangle_bounds(tv))
    assert np.array_equal(nb, d['node_bounds']) and np.array_equal(nm, d['node_meta'])
    assert np.array_equal(tv[order], d['tri_verts'])
    # the fixture numbers materials by first appearance in BVH order: same partition, other labels
    pairs = set(zip(mat[order].tolist(), d['tri_mat'].tolist()))
    assert len(pairs) == 3 and len({a for a, _ in pairs}) == 3 and len({b for _, b in pairs}) == 3


@pytest.mark.parametrize('name,kind', CASES)
def test_oracle_rays_and_light_samples(golden_dir, name, kind):
    d = load(golden_dir, name)
    osc = O.OracleScene(pa.flat_scene_from_npz(d))
    log, res = d['log_rays'], d['log_rres']
    rays = rays_from_log(log)
    for any_hit in (0, 1):
        sel = np.nonzero(log[:, 1].astype(int) == any_hit)[0]
        assert len(sel) > 100
        prim, t, _, _ = osc.intersect(rays[:, sel], any_hit=bool(any_hit))
        if any_hit:
            assert np.array_equal(prim, res[sel, 0].astype(np.int32))
        else:
            assert np.array_equal(prim, res[sel, 1].astype(np.int32))
            np.testing.assert_allclose(t, res[sel, 2], rtol=RTOL)
    sl, which = d['log_sl'], d['log_sl_light']
    seen = set()
    for row, li in zip(sl[:900], which[:900]):
        out = osc.sample_l(int(li), row[0:3], row[3], row[4], row[5])
        np.testing.assert_allclose(out[0:7], row[7:14], rtol=RTOL, atol=1e-12)      # Li (inf outside the cone, Q9), wi, pdf
        np.testing.assert_allclose(out[7:15], row[14:22], rtol=RTOL, atol=1e-12)    # shadow ray
        seen.add(int(li))
    assert seen == {0, 1, 2}


@pytest.mark.parametrize('name,kind', CASES)
def test_oracle_render_matches_reference(golden_dir, name, kind):
    d = load(golden_dir, name)
    flat = pa.flat_scene_from_npz(d)
    osc = O.OracleScene(flat)
    assert d['samples'].shape[1] == flat.table_dims(kind)
    cam, film, smp, tile = descs(d)
    st = A.pt_stats()
    acc, L = osc.render(cam, film, smp, pa.integrator_desc(kind, 5, pretest=True), tile, table=d['samples'], stats=st)
    assert rel_err(L, d['L']) < 1e-8
    np.testing.assert_allclose(acc[..., 0:3], d['film_Lxyz'], rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(acc[..., 3], d['film_wsum'], rtol=1e-10, atol=1e-12)
    # the spot light's I/0 outside its cone makes whole samples infinite; the renderer's guard zeroes them
    assert st.guard_inf + st.guard_nan > 0
    assert int((d['L'].sum(axis=1) == 0).sum()) >= st.guard_inf + st.guard_nan


# ------------------------------------------------------------------------------------------
@pytest.fixture(scope='module')
def ctx():
    import torch
    assert torch.cuda.is_available(), 'these tests need the MI355X'
    from pytracer_amd._lib import Context
    c = Context(0)
    yield c
    c.close()


@pytest.mark.gpu
@pytest.mark.parametrize('name,kind', CASES)
def test_gpu_render_matches_reference_and_oracle(ctx, golden_dir, name, kind):
    import torch
    d = load(golden_dir, name)
    flat = pa.flat_scene_from_npz(d)
    ctx.upload(flat)
    cam, film, smp, tile = descs(d)
    integ = pa.integrator_desc(kind, 5, pretest=True)
    n = len(d['samples'])
    acc = torch.zeros((film.y_cnt, film.x_cnt, 4), dtype=torch.float64, device='cuda')
    L = torch.zeros((n, 3), dtype=torch.float64, device='cuda')
    sg, so = A.pt_stats(), A.pt_stats()
    ctx.render(cam, film, smp, integ, tile, acc, table=torch.from_numpy(d['samples']).cuda(), sample_L=L, stats=sg)
    torch.cuda.synchronize()
    acc, L = acc.cpu().numpy(), L.cpu().numpy()
    assert rel_err(L, d['L']) < 1e-4
    np.testing.assert_allclose(acc[..., 0:3], d['film_Lxyz'], rtol=1e-4, atol=1e-9)
    np.testing.assert_allclose(acc[..., 3], d['film_wsum'], rtol=1e-4, atol=1e-9)
    oacc, oL = O.OracleScene(flat).render(cam, film, smp, integ, tile, table=d['samples'], stats=so)
    assert rel_err(L, oL) < RTOL
    np.testing.assert_allclose(acc, oacc, rtol=RTOL, atol=1e-12)
    a, b = sg.as_dict(), so.as_dict()
    for k in ('camera_samples', 'rays_closest', 'rays_any', 'guard_nan', 'guard_neg', 'guard_inf'):
        assert a[k] == b[k], (k, a, b)


@pytest.mark.gpu
def test_gpu_logged_rays_bitwise(ctx, golden_dir):
    import torch
    d = load(golden_dir, CASES[1][0])
    flat = pa.flat_scene_from_npz(d)
    ctx.upload(flat)
    osc = O.OracleScene(flat)
    log, res = d['log_rays'], d['log_rres']
    rays = rays_from_log(log)
    for any_hit in (False, True):
        sel = np.nonzero(log[:, 1].astype(int) == int(any_hit))[0]
        rt = torch.from_numpy(np.ascontiguousarray(rays[:, sel])).cuda()
        sg, so = A.pt_stats(), A.pt_stats()
        prim, t, b1, b2 = ctx.intersect(rt, any_hit=any_hit, stats=sg)
        torch.cuda.synchronize()
        op, ot, ob1, ob2 = osc.intersect(rays[:, sel], any_hit=any_hit, stats=so)
        assert np.array_equal(prim.cpu().numpy(), op)
        assert np.array_equal(prim.cpu().numpy(), res[sel, 0 if any_hit else 1].astype(np.int32))
        if not any_hit:
            assert np.array_equal(t.cpu().numpy(), ot) and np.array_equal(b1.cpu().numpy(), ob1)
            assert np.array_equal(b2.cpu().numpy(), ob2)
        assert sg.node_visits == so.node_visits and sg.tri_tests == so.tri_tests
