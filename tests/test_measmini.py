"""§8(f) rank 3, measured BRDF — pinned by the reference (tools/oracle/gen_fixtures.py:gen_meas →
tests/golden/measmini_*.npz): two MeasuredMaterials (material/material.py:403-524) over synthetic irregular isotropic
tables (600 samples, and 40 samples whose query balls have to grow; the reference ships no .brdf file, the generator
writes them in the format its loader reads), whose IrIsotropicBRDF (reflection/bdf/measured.py:39-66) is evaluated
directly (f, pdf, sample_f for 300 direction pairs, non-unit and other-hemisphere ones included) and under
DirectLightingIntegrator, PathIntegrator and WhittedIntegrator.  The sample values are the RGB the reference made of the
measured spectra (Spectrum = RGBSpectrum, spectral/spectrum.py:610)."""
import numpy as np
import pytest

import pytracer_amd as pa
from pytracer_amd import _abi as A
from pytracer_amd import flatten as F
import oracle_lib as O
from test_c4mini import load, rel_err, descs

RTOL = 1e-9
CASES = [('measmini_direct_24x24x1.npz', A.PT_INTEG_DIRECT), ('measmini_path_24x24x4.npz', A.PT_INTEG_PATH),
         ('measmini_whitted_24x24x1.npz', A.PT_INTEG_WHITTED)]


def test_scene_has_measured_lobes(golden_dir):
    d = load(golden_dir, CASES[0][0])
    flat = pa.flat_scene_from_npz(d)
    meas = [(i, m.lobes[0]) for i, m in enumerate(flat._materials) if m.n_lobes == 1 and
            m.lobes[0].type == A.PT_LOBE_MEASURED_IRISO]
    # materials are numbered by first appearance in BVH.primitives: the sparse table comes first here
    assert [(l.a, l.b) for _, l in meas] == [(0., 40.), (40., 600.)]
    desc = flat.desc()
    assert desc.n_brdf_samples == 640 and desc.brdf_p and desc.brdf_v
    np.testing.assert_array_equal(flat.brdf_p[40:], d['mat%d_brdf_p' % meas[1][0]])
    np.testing.assert_array_equal(flat.brdf_v[:40], d['mat%d_brdf_v' % meas[0][0]])
    # brdf_remap's range: x = sin sin in [0,1], y = dphi/pi in [0,1], z = cos cos (upper hemisphere measurements)
    assert flat.brdf_p.min() >= 0. and flat.brdf_p.max() <= 1.


def _make(name, kw):
    o = type(name, (object,), {})()
    for k, v in kw.items():
        setattr(o, k, v)
    return o


def test_flatten_reads_reference_shaped_measured_material():
    smp = np.array([_make('IrIsotropicBRDFSample', dict(p=np.array([.1 * k, .2, .3]), v=np.array([.5, .6, .7 + k])))
                    for k in range(4)], dtype=object)
    mat = _make('MeasuredMaterial', dict(bump_map=None, regular_data=None, theta_phi_tree=object(), theta_phi_samples=smp))
    kind, params = F.material_params(mat)
    assert kind == 'measured' and params['brdf_p'].shape == (4, 3)
    lobes = F.lobes_from_params(kind, **params)
    tables = []
    m = F._pack_material(lobes, [], tables)
    m2 = F._pack_material(lobes, [], tables)
    assert (m.lobes[0].type, m.lobes[0].a, m.lobes[0].b) == (A.PT_LOBE_MEASURED_IRISO, 0., 4.)
    assert (m2.lobes[0].a, m2.lobes[0].b) == (4., 4.)
    # the second MeasuredMaterial of one file has no samples in the reference (misspelt attribute, material.py:434)
    mat.theta_phi_samples = None
    with pytest.raises(ValueError):
        F.material_params(mat)
    mat.theta_phi_tree = None
    with pytest.raises(ValueError):
        F.material_params(mat)
    mat.regular_data = np.zeros(3)
    with pytest.raises(NotImplementedError):
        F.material_params(mat)


def test_oracle_lobe_matches_reference(golden_dir):
    d = load(golden_dir, CASES[0][0])
    flat = pa.flat_scene_from_npz(d)
    osc = O.OracleScene(flat)
    meas = [i for i, m in enumerate(flat._materials) if m.n_lobes == 1 and m.lobes[0].type == A.PT_LOBE_MEASURED_IRISO]
    ev = d['brdf_eval']
    assert len(ev) == 300
    for row in ev:
        mi, wo, wi, u1, u2 = int(row[0]), row[1:4], row[4:7], row[7], row[8]
        want = row[9:20]           # f[3], pdf, sample pdf, sample wi[3], sample f[3]
        got = osc.lobe_eval(meas[mi], 0, wo, wi, u1, u2)
        np.testing.assert_allclose(got, want, rtol=1e-10, atol=1e-300)
    # arbitrary (non-unit) directions can land so far from every sample that all weights underflow: the reference
    # returns 0/0 = NaN there (assert_allclose above compares NaN with NaN as equal), and so does the restatement
    fin = np.isfinite(ev[:, 9:12]).all(1)
    assert fin.sum() > 250 and (ev[fin, 9:12] > 0).all()


@pytest.mark.parametrize('name,kind', CASES)
def test_oracle_matches_reference(golden_dir, name, kind):
    d = load(golden_dir, name)
    flat = pa.flat_scene_from_npz(d)
    osc = O.OracleScene(flat)
    cam, film, smp, tile = descs(d)
    st = A.pt_stats()
    acc, L = osc.render(cam, film, smp, pa.integrator_desc(kind, 5, pretest=True), tile, table=d['samples'], stats=st)
    assert rel_err(L, d['L']) < 1e-8
    np.testing.assert_allclose(acc[..., 0:3], d['film_Lxyz'], rtol=1e-8, atol=1e-10)
    kinds = d['log_rays'][:, 1].astype(int)
    assert st.rays_closest == int((kinds == 0).sum()) and st.rays_any == int((kinds == 1).sum())
    # the tables matter: with the measured values halved the image changes
    half = dict(d)
    for k in [k for k in d if k.endswith('_brdf_v')]:
        half[k] = d[k] * .5
    _, L2 = O.OracleScene(pa.flat_scene_from_npz(half)).render(cam, film, smp, pa.integrator_desc(kind, 5, True), tile,
                                                               table=d['samples'])
    assert rel_err(L2, d['L']) > 1e-2


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
    np.testing.assert_allclose(acc[..., 0:3], d['film_Lxyz'], rtol=1e-4, atol=1e-9)
    oacc, oL = O.OracleScene(flat).render(cam, film, smp, integ, tile, table=d['samples'], stats=so)
    assert rel_err(L, oL) < RTOL
    a, b = sg.as_dict(), so.as_dict()
    for k in ('camera_samples', 'rays_closest', 'rays_any', 'guard_nan', 'guard_neg', 'guard_inf'):
        assert a[k] == b[k], (k, a, b)
    # production streams, a larger render: the grid queries of the device against the table scans of the oracle
    smp2 = pa.sampler_desc(0, 24, 0, 24, 3, 3, True, 23)
    tile2 = pa.tile_desc(0, 24, 0, 24, 0, 9)
    acc2 = torch.zeros((film.y_cnt, film.x_cnt, 4), dtype=torch.float64, device='cuda')
    ctx.render(cam, film, smp2, pa.integrator_desc(kind, 5, False), tile2, acc2)
    o2, _ = O.OracleScene(flat).render(cam, film, smp2, pa.integrator_desc(kind, 5, False), tile2, want_L=False)
    a2 = acc2.cpu().numpy()
    assert np.abs(a2 - o2).max() <= 1e-9 * np.abs(o2).max()


@pytest.mark.gpu
def test_gpu_rejects_bad_sample_ranges(ctx, golden_dir):
    d = load(golden_dir, CASES[0][0])
    flat = pa.flat_scene_from_npz(d)
    for m in flat._materials:
        if m.n_lobes and m.lobes[0].type == A.PT_LOBE_MEASURED_IRISO:
            m.lobes[0].b = 1e6          # beyond n_brdf_samples
    with pytest.raises(Exception):
        ctx.upload(flat)
    ctx.upload(pa.flat_scene_from_npz(d))


@pytest.mark.gpu
def test_gpu_c5_workload_matches_oracle(ctx, golden_dir):
    """bench.py --workload c5 in small: the head mesh with a 1000-sample measured table (8x8x8 grid on the device) and a
    checkerboard backdrop, production sample streams, 16x8 strata like the full configuration: device == oracle."""
    import os
    import torch
    from pytracer_amd import host, workloads
    flat = workloads.c5_head_measured(os.path.join(golden_dir, 'head_mesh_scene.npz'))
    assert len(flat.brdf_p) == 1000 and len(flat.textures) == 1
    ctx.upload(flat)
    cam, film = host.head_mesh_camera_film(48, 48)
    smp = pa.sampler_desc(0, 48, 0, 48, 16, 8, True, 3)
    tile = pa.tile_desc(8, 40, 8, 40, 0, 128)
    integ = pa.integrator_desc(A.PT_INTEG_PATH, 5, False)
    acc = torch.zeros((film.y_cnt, film.x_cnt, 4), dtype=torch.float64, device='cuda')
    sg, so = A.pt_stats(), A.pt_stats()
    ctx.render(cam, film, smp, integ, tile, acc, stats=sg)
    torch.cuda.synchronize()
    oacc, _ = O.OracleScene(flat).render(cam, film, smp, integ, tile, stats=so, want_L=False)
    a = acc.cpu().numpy()
    # 1e-6 rather than the 1e-9 of the analytic lobes: brdf_remap takes the azimuth difference of wo and wi, which is
    # ill-conditioned where a direction is nearly parallel to the normal (x, y of the local vector are rounding
    # residue of the frame dot products), and the Gaussian weights exp(-100 d^2) amplify it.  Measured on this render:
    # 3 of 131 072 samples differ by more than 1e-9 relative (largest 6.4e-8, a 9e4-radiance outlier), the rest agree
    # to 1e-11; the film differs by 3.7e-8 of its maximum.
    assert np.abs(a - oacc).max() <= 1e-6 * np.abs(oacc).max()
    # (node visits and triangle tests of bounce rays may differ in the last digits: their directions carry the device
    # libm's sincos / pow, one ulp from glibc's)
    x, y = sg.as_dict(), so.as_dict()
    for k in ('camera_samples', 'rays_closest', 'rays_any', 'guard_nan', 'guard_neg', 'guard_inf'):
        assert x[k] == y[k], (k, x, y)


@pytest.mark.gpu
@pytest.mark.parametrize('n_samples', [2, 3, 5000])
def test_gpu_table_sizes(ctx, golden_dir, n_samples):
    """Tables the LDS copy cannot hold (5000 samples = 240 KB: the walk runs from HBM) and tables of 2 or 3 samples
    (2: every ball grows past 1.5 and takes both samples, measured.py:56; 3: the first ball that reaches all three):
    device == oracle on a small C5-like render."""
    import os
    import torch
    from pytracer_amd import host, workloads
    flat = workloads.c5_head_measured(os.path.join(golden_dir, 'head_mesh_scene.npz'), n_samples=n_samples, seed=11)
    assert len(flat.brdf_p) == n_samples
    ctx.upload(flat)
    cam, film = host.head_mesh_camera_film(32, 32)
    smp = pa.sampler_desc(0, 32, 0, 32, 2, 2, True, 5)
    tile = pa.tile_desc(0, 32, 0, 32, 0, 4)
    integ = pa.integrator_desc(A.PT_INTEG_PATH, 5, False)
    acc = torch.zeros((film.y_cnt, film.x_cnt, 4), dtype=torch.float64, device='cuda')
    sg, so = A.pt_stats(), A.pt_stats()
    ctx.render(cam, film, smp, integ, tile, acc, stats=sg)
    torch.cuda.synchronize()
    oacc, _ = O.OracleScene(flat).render(cam, film, smp, integ, tile, stats=so, want_L=False)
    a = acc.cpu().numpy()
    assert np.isfinite(a).all() and np.abs(oacc).max() > 0
    assert np.abs(a - oacc).max() <= 1e-6 * np.abs(oacc).max()       # tolerance: see test_gpu_c5_workload_matches_oracle
    x, y = sg.as_dict(), so.as_dict()
    for k in ('camera_samples', 'rays_closest', 'rays_any', 'guard_nan', 'guard_neg', 'guard_inf'):
        assert x[k] == y[k], (k, x, y)
