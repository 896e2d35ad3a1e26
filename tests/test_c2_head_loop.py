"""C2 (BASELINE.json configs[1]): examples/head_loop.py — the head after 3 levels of Loop subdivision
(177 666 triangles, 352 417 BVH nodes), DirectLightingIntegrator, 512x512, 4x4 spp.

tests/golden/c2_head_loop.npz was written by running the reference (tools/oracle/gen_fixtures.py:gen_c2):
SHA-256 digests of the subdivided head mesh and of the reference's flattened tree (too big to commit), 300
reference-traced rays and the reference's render of a 12x10-pixel window across the head's silhouette.
CPU part: native Loop subdivision of the control mesh and the native builder reproduce mesh and tree bit for bit; oracle vs reference.  GPU part: the HIP
path on that tree vs reference and oracle, and the whole C2 render through size-independent
properties."""
import hashlib
import os

import numpy as np
import pytest

import pytracer_amd as pa
from pytracer_amd import _abi as A, accel
from pytracer_amd.flatten import FlatScene, lobes_from_params, _MAT_KINDS
import oracle_lib as O

RTOL = 1e-9


def rel_err(a, ref):
    scale = np.maximum(np.abs(ref).max(axis=-1, keepdims=True), 1e-12)
    return (np.abs(a - ref) / scale).max()


def rays_from_log(log):
    r = np.empty((8, len(log)))
    r[0:6] = log[:, 2:8].T
    r[6] = log[:, 8]
    r[7] = log[:, 9]
    return r


@pytest.fixture(scope='module')
def c2(golden_dir):
    from pytracer_amd import workloads
    c2_npz, loop_npz = os.path.join(golden_dir, 'c2_head_loop.npz'), os.path.join(golden_dir, 'loop_subdiv.npz')
    d = dict(np.load(c2_npz))
    flat = workloads.c2_head_loop(c2_npz, loop_npz)       # native Loop subdivision + native BVH build
    meshes, _ = workloads.c2_meshes(c2_npz, loop_npz)
    d['_mesh0'] = meshes[0]
    return d, flat, None


def digest(a):
    return np.frombuffer(hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest(), np.uint8)


def test_native_subdivision_reproduces_the_reference_mesh(c2):
    d, _, _ = c2
    P, idx, _ = d['_mesh0']
    assert len(P) == int(d['mesh0_nverts']) == 89078 and len(idx) == 3 * int(d['mesh0_ntris'])
    assert np.array_equal(digest(P), d['sha256_mesh0_p'])
    assert np.array_equal(digest(idx.astype(np.int32)), d['sha256_mesh0_idx'])


def test_native_builder_reproduces_the_reference_tree(c2):
    d, flat, depth = c2
    assert flat.n_nodes == int(d['n_nodes']) == 352417 and flat.n_tris == int(d['n_tris']) == 177666
    assert np.array_equal(digest(flat.node_bounds), d['sha256_node_bounds'])
    assert np.array_equal(digest(flat.node_meta), d['sha256_node_meta'])
    assert np.array_equal(digest(flat.tri_verts), d['sha256_tri_verts'])      # = the reference's primitive order
    assert np.array_equal(digest(flat.tri_mat), d['sha256_tri_mat'])
    assert np.array_equal(digest(flat.tri_flip), d['sha256_tri_flip'])


def test_oracle_on_reference_rays_and_window(c2):
    d, flat, _ = c2
    osc = O.OracleScene(flat)
    rays = np.ascontiguousarray(d['bvh_rays'].T)
    prim, t, _, _ = osc.intersect(rays, any_hit=False)
    assert np.array_equal(prim, d['bvh_res'][:, 1].astype(np.int32))
    np.testing.assert_allclose(t, d['bvh_res'][:, 2], rtol=RTOL)
    occ, _, _, _ = osc.intersect(rays, any_hit=True)
    assert np.array_equal(occ, d['bvh_res'][:, 3].astype(np.int32))
    log, res = d['log_rays'], d['log_rres']
    lr = rays_from_log(log)
    for any_hit in (0, 1):
        sel = np.nonzero(log[:, 1].astype(int) == any_hit)[0]
        p, tt, _, _ = osc.intersect(lr[:, sel], any_hit=bool(any_hit))
        assert np.array_equal(p, res[sel, 0 if any_hit else 1].astype(np.int32))
    cam, film = pa.camera_film_from_npz(d)
    sub = d['sub']
    smp = pa.sampler_desc(sub[0], sub[1], sub[2], sub[3], 4, 4, True, 0)
    tile = pa.tile_desc(sub[0], sub[1], sub[2], sub[3], 0, 16)
    acc, L = osc.render(cam, film, smp, pa.integrator_desc(A.PT_INTEG_DIRECT, 5, True), tile, table=d['samples'])
    assert (d['L'].sum(axis=1) > 0).mean() > .2            # the window is partly lit
    assert rel_err(L, d['L']) < 1e-8
    np.testing.assert_allclose(acc[..., 0:3], d['film_Lxyz'], rtol=1e-8, atol=1e-10)


@pytest.fixture(scope='module')
def ctx():
    import torch
    assert torch.cuda.is_available(), 'these tests need the MI355X'
    from pytracer_amd._lib import Context
    c = Context(0)
    yield c
    c.close()


@pytest.mark.gpu
def test_gpu_rays_and_window(ctx, c2):
    import torch
    d, flat, _ = c2
    ctx.upload(flat)
    osc = O.OracleScene(flat)
    lr = rays_from_log(d['log_rays'])
    allr = np.concatenate([np.ascontiguousarray(d['bvh_rays'].T), lr], axis=1)
    rt = torch.from_numpy(allr).cuda()
    for any_hit in (False, True):
        sg, so = A.pt_stats(), A.pt_stats()
        prim, t, b1, b2 = ctx.intersect(rt, any_hit=any_hit, stats=sg)
        torch.cuda.synchronize()
        op, ot, ob1, ob2 = osc.intersect(allr, any_hit=any_hit, stats=so)
        assert np.array_equal(prim.cpu().numpy(), op)
        assert np.array_equal(prim.cpu().numpy()[:300], d['bvh_res'][:, 3 if any_hit else 1].astype(np.int32))
        if not any_hit:
            assert np.array_equal(t.cpu().numpy(), ot) and np.array_equal(b1.cpu().numpy(), ob1)
        assert sg.as_dict() == so.as_dict()
    cam, film = pa.camera_film_from_npz(d)
    sub = d['sub']
    smp = pa.sampler_desc(sub[0], sub[1], sub[2], sub[3], 4, 4, True, 0)
    tile = pa.tile_desc(sub[0], sub[1], sub[2], sub[3], 0, 16)
    integ = pa.integrator_desc(A.PT_INTEG_DIRECT, 5, True)
    n = len(d['samples'])
    acc = torch.zeros((film.y_cnt, film.x_cnt, 4), dtype=torch.float64, device='cuda')
    L = torch.zeros((n, 3), dtype=torch.float64, device='cuda')
    ctx.render(cam, film, smp, integ, tile, acc, table=torch.from_numpy(d['samples']).cuda(), sample_L=L)
    torch.cuda.synchronize()
    assert rel_err(L.cpu().numpy(), d['L']) < 1e-4
    np.testing.assert_allclose(acc.cpu().numpy()[..., 0:3], d['film_Lxyz'], rtol=1e-4, atol=1e-9)
    oacc, oL = osc.render(cam, film, smp, integ, tile, table=d['samples'])
    assert rel_err(L.cpu().numpy(), oL) < RTOL


@pytest.mark.gpu
def test_gpu_full_c2_render_properties(ctx, c2):
    """The whole configs[1] job: 512x512, 16 spp, Direct.  No oracle at this size (minutes of CPU), so:
    census identities, any-hit == closest-hit on the camera rays, shard additivity, and the oracle on
    one 24x24 window of the same production sample stream."""
    import torch
    from pytracer_amd import host
    d, flat, _ = c2
    ctx.upload(flat)
    res = 512
    cam, film = host.head_mesh_camera_film(res, res)
    smp = pa.sampler_desc(0, res, 0, res, 4, 4, True, 5)
    integ = pa.integrator_desc(A.PT_INTEG_DIRECT, 5, False)

    def render(tile):
        acc = torch.zeros((film.y_cnt, film.x_cnt, 4), dtype=torch.float64, device='cuda')
        n = (tile.x1 - tile.x0) * (tile.y1 - tile.y0) * (tile.s1 - tile.s0)
        L = torch.zeros((n, 3), dtype=torch.float64, device='cuda')
        st = A.pt_stats()
        ctx.render(cam, film, smp, integ, tile, acc, sample_L=L, stats=st)
        torch.cuda.synchronize()
        return acc.cpu().numpy(), L.cpu().numpy(), st

    full, L, st = render(pa.tile_desc(0, res, 0, res, 0, 16))
    assert st.camera_samples == res * res * 16 and st.rays_closest >= st.camera_samples
    assert st.guard_nan == 0 and st.guard_neg == 0
    assert np.isfinite(full).all() and (full[..., 3] != 0).all()
    # every camera sample hits the backdrop or the head; 2 shadow rays per lit vertex at most
    assert st.rays_any <= 2 * st.rays_closest
    # sample-index shards (the multi-GPU split) add up to the whole
    a0, _, _ = render(pa.tile_desc(0, res, 0, res, 0, 7))
    a1, _, _ = render(pa.tile_desc(0, res, 0, res, 7, 16))
    np.testing.assert_allclose(a0 + a1, full, rtol=1e-12, atol=1e-12)
    # one window against the oracle on the same in-kernel sample stream
    wt = pa.tile_desc(100, 124, 150, 174, 0, 16)
    _, Lw, _ = render(wt)
    oacc, oL = O.OracleScene(flat).render(cam, film, smp, integ, wt)
    assert rel_err(Lw, oL) < RTOL
    assert (oL.sum(axis=1) > 0).mean() > .1
