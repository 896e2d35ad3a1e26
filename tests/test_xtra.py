"""§8(f) rank 2, pinned by the reference (tools/oracle/gen_fixtures.py:gen_xtra → tests/golden/xtra_*.npz):
WhittedIntegrator (integrator/surface/whitted.py:16-59), MetalMaterial (Microfacet + FresnelConductor,
material.py:242-258), OrthoCamera (camera/projective.py:65-127) and the thin-lens branches of the
perspective and orthographic cameras.  (EnvironmentCamera: the reference itself raises on every attempt
with this scene, see the generator.)"""
import numpy as np
import pytest

import pytracer_amd as pa
from pytracer_amd import _abi as A
import oracle_lib as O
from test_c4mini import load, rays_from_log, rel_err, descs

RTOL = 1e-9
CASES = [('xtra_whitted.npz', A.PT_INTEG_WHITTED), ('xtra_metal_direct.npz', A.PT_INTEG_DIRECT),
         ('xtra_metal_path.npz', A.PT_INTEG_PATH), ('xtra_lens_direct.npz', A.PT_INTEG_DIRECT),
         ('xtra_ortho_direct.npz', A.PT_INTEG_DIRECT), ('xtra_ortholens_whitted.npz', A.PT_INTEG_WHITTED)]


def test_fixtures_cover_what_they_claim(golden_dir):
    flat = pa.flat_scene_from_npz(load(golden_dir, 'xtra_whitted.npz'))
    assert any(l[0] == A.PT_LOBE_MICROFACET_BLINN_COND for m in flat.material_lobes for l in m)
    assert int(load(golden_dir, 'xtra_ortho_direct.npz')['cam_kind']) == A.PT_CAM_ORTHO
    assert float(load(golden_dir, 'xtra_lens_direct.npz')['cam_lens_rad']) > 0
    d = load(golden_dir, 'xtra_ortholens_whitted.npz')
    assert int(d['cam_kind']) == A.PT_CAM_ORTHO and float(d['cam_lens_rad']) > 0 and d['samples'].shape[1] == 5


@pytest.mark.parametrize('name,kind', CASES)
def test_oracle_matches_reference(golden_dir, name, kind):
    d = load(golden_dir, name)
    flat = pa.flat_scene_from_npz(d)
    osc = O.OracleScene(flat)
    cam, film, smp, tile = descs(d)
    np.testing.assert_allclose(O.camera_rays(cam, d['samples']), d['rays'][:, 0:6], rtol=1e-12, atol=1e-14)
    assert d['samples'].shape[1] == flat.table_dims(kind)
    log, res = d['log_rays'], d['log_rres']
    rays = rays_from_log(log)
    kinds = log[:, 1].astype(int)
    for any_hit in (0, 1):
        sel = np.nonzero(kinds == any_hit)[0]
        prim, t, _, _ = osc.intersect(rays[:, sel], any_hit=bool(any_hit))
        assert np.array_equal(prim, res[sel, 0 if any_hit else 1].astype(np.int32))
    st = A.pt_stats()
    acc, L = osc.render(cam, film, smp, pa.integrator_desc(kind, 5, pretest=True), tile, table=d['samples'], stats=st)
    assert (d['L'].sum(axis=1) > 0).mean() > .3
    assert rel_err(L, d['L']) < 1e-8
    np.testing.assert_allclose(acc[..., 0:3], d['film_Lxyz'], rtol=1e-8, atol=1e-8 * np.abs(d['film_Lxyz']).max())
    np.testing.assert_allclose(acc[..., 3], d['film_wsum'], rtol=1e-10, atol=1e-12)


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
def test_gpu_matches_reference_and_oracle(ctx, golden_dir, name, kind):
    import torch
    d = load(golden_dir, name)
    flat = pa.flat_scene_from_npz(d)
    ctx.upload(flat)
    cam, film, smp, tile = descs(d)
    integ = pa.integrator_desc(kind, 5, pretest=True)
    acc = torch.zeros((film.y_cnt, film.x_cnt, 4), dtype=torch.float64, device='cuda')
    L = torch.zeros((len(d['samples']), 3), dtype=torch.float64, device='cuda')
    sg, so = A.pt_stats(), A.pt_stats()
    ctx.render(cam, film, smp, integ, tile, acc, table=torch.from_numpy(d['samples']).cuda(), sample_L=L, stats=sg)
    torch.cuda.synchronize()
    acc, L = acc.cpu().numpy(), L.cpu().numpy()
    assert rel_err(L, d['L']) < 1e-4
    np.testing.assert_allclose(acc[..., 0:3], d['film_Lxyz'], rtol=1e-4, atol=1e-4 * np.abs(d['film_Lxyz']).max())
    oacc, oL = O.OracleScene(flat).render(cam, film, smp, integ, tile, table=d['samples'], stats=so)
    assert rel_err(L, oL) < RTOL
    a, b = sg.as_dict(), so.as_dict()
    for k in ('camera_samples', 'rays_closest', 'rays_any', 'guard_nan', 'guard_neg', 'guard_inf'):
        assert a[k] == b[k], (k, a, b)


@pytest.mark.gpu
def test_gpu_whitted_production_stream(ctx, golden_dir):
    """Whitted without a table (in-kernel sample stream) against the oracle on the same stream."""
    import torch
    d = load(golden_dir, 'xtra_whitted.npz')
    flat = pa.flat_scene_from_npz(d)
    ctx.upload(flat)
    cam, film, _, _ = descs(d)
    smp = pa.sampler_desc(0, 24, 0, 24, 2, 2, True, 9)
    tile = pa.tile_desc(0, 24, 0, 24, 0, 4)
    integ = pa.integrator_desc(A.PT_INTEG_WHITTED, 5, pretest=False)
    acc = torch.zeros((film.y_cnt, film.x_cnt, 4), dtype=torch.float64, device='cuda')
    L = torch.zeros((24 * 24 * 4, 3), dtype=torch.float64, device='cuda')
    ctx.render(cam, film, smp, integ, tile, acc, sample_L=L)
    torch.cuda.synchronize()
    oacc, oL = O.OracleScene(flat).render(cam, film, smp, integ, tile)
    assert rel_err(L.cpu().numpy(), oL) < RTOL
