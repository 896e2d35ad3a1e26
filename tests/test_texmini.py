"""§8(f) rank 3, textures — pinned by the reference (tools/oracle/gen_fixtures.py:gen_tex → tests/golden/texmini_*.npz):
Chekcerboard2DTexture (default closed-form antialiasing over a UV mapping; point-sampled over a planar mapping with a
MixTexture child), Checkboard3DTexture through IdentityMapping3D (with a ScaleTexture child), BilerpTexture through
SphericalMapping2D and UVTexture, as Matte Kd / Plastic Kd and Ks, under DirectLightingIntegrator and PathIntegrator.
Materials evaluate textures on a copy of the shading geometry with zero differentials (material.py:128,172), so the
closed-form checkerboard equals point sampling — the reference renders below are the proof."""
import numpy as np
import pytest

import pytracer_amd as pa
from pytracer_amd import _abi as A
from pytracer_amd import flatten as F
import oracle_lib as O
from test_c4mini import load, rays_from_log, rel_err, descs

RTOL = 1e-9
CASES = [('texmini_direct_24x24x1.npz', A.PT_INTEG_DIRECT), ('texmini_path_24x24x4.npz', A.PT_INTEG_PATH),
         ('tex2mini_direct_24x24x1.npz', A.PT_INTEG_DIRECT), ('tex2mini_path_24x24x4.npz', A.PT_INTEG_PATH),
         ('tex2mini_whitted_24x24x1.npz', A.PT_INTEG_WHITTED),
         # meshes with uv coordinates (26 of them degenerate: coordinate_system fallback, triangle.py:219-221)
         ('uvmini_direct_24x24x1.npz', A.PT_INTEG_DIRECT), ('uvmini_path_24x24x4.npz', A.PT_INTEG_PATH)]


def test_scene_has_textured_lobes(golden_dir):
    flat = pa.flat_scene_from_npz(load(golden_dir, CASES[0][0]))
    kinds = sorted((t['kind'], t['mapping']) for t in flat.textures)
    assert kinds == sorted([(A.PT_TEX_CHECKER2D, A.PT_MAP_UV), (A.PT_TEX_CHECKER3D, A.PT_MAP_IDENTITY3D),
                            (A.PT_TEX_BILERP, A.PT_MAP_SPHERICAL), (A.PT_TEX_UV, A.PT_MAP_UV),
                            (A.PT_TEX_CHECKER2D, A.PT_MAP_PLANAR)])
    d = flat.desc()
    assert d.n_textures == 5
    tex = [[m.lobes[k].tex for k in range(m.n_lobes)] for m in flat._materials]
    assert sorted(x for t in tex for x in t) == [1, 2, 3, 4, 5]
    # folded children: ScaleTexture(const, const) and MixTexture(const, const, const)
    c3 = [t for t in flat.textures if t['kind'] == A.PT_TEX_CHECKER3D][0]
    np.testing.assert_allclose(c3['v'][1], [.45, .45, .1])
    pl = [t for t in flat.textures if t['mapping'] == A.PT_MAP_PLANAR][0]
    np.testing.assert_allclose(pl['v'][0], .75 * np.full(3, .9) + .25 * np.full(3, .5))


def test_unsupported_texture_uses_raise():
    rec = dict(kind=A.PT_TEX_UV, mapping=A.PT_MAP_UV, m=np.zeros(16), v=np.zeros((4, 3)))
    uber = dict(Kd=np.zeros(3), Ks=np.zeros(3), Kr=np.zeros(3), Kt=np.zeros(3), roughness=.1, opacity=np.ones(3), eta=1.5)
    with pytest.raises(NotImplementedError):     # a varying opacity also decides the first lobe and scales the others
        F.lobes_from_params('uber', **dict(uber, opacity=rec))
    with pytest.raises(NotImplementedError):
        F.lobes_from_params('metal', eta=rec, k=np.ones(3), roughness=.1)
    lobes = F.lobes_from_params('plastic', Kd=rec, Ks=np.full(3, .2), roughness=.1)
    assert lobes[0][-1] == {'tex': rec, 'drop': False} and not isinstance(lobes[1][-1], dict)
    # Uber / Mirror / Glass: constant-black colours drop the lobe on the host, textured ones carry the per-hit test
    lobes = F.lobes_from_params('uber', **dict(uber, Kr=rec, opacity=np.full(3, .5)))
    assert [l[0] for l in lobes] == [A.PT_LOBE_SPEC_TRANSMIT, A.PT_LOBE_SPEC_REFLECT]
    assert lobes[1][-1] == {'tex': rec, 'drop': True} and np.array_equal(lobes[1][1], np.full(3, .5))
    m = F._pack_material(lobes, [])
    assert m.lobes[1].tex == (1 | A.PT_LOBE_DROP_IF_BLACK) and m.lobes[0].tex == 0
    assert F.lobes_from_params('mirror', Kr=np.zeros(3)) == []
    assert F.lobes_from_params('mirror', Kr=rec)[0][-1]['drop'] is True


def _make(name, kw):
    o = type(name, (object,), {})()
    for k, v in kw.items():
        setattr(o, k, v)
    return o


def test_texture_record_reads_reference_shaped_objects(golden_dir):
    """flatten reads textures by class name and public attributes, like everything else: objects shaped like the
    reference's (its class names, typos included) give the records the fixture holds."""
    const = lambda v: _make('ConstantTexture', dict(value=np.asarray(v, float)))
    cb = _make('Chekcerboard2DTexture', dict(mapping=_make('UVMapping2D', dict(su=6., sv=5., du=.25, dv=.1)),
                                             tex1=const([.8, .2, .2]), tex2=const([.1, .1, .7])))
    flat = pa.flat_scene_from_npz(load(golden_dir, CASES[0][0]))
    want = [t for t in flat.textures if (t['kind'], t['mapping']) == (A.PT_TEX_CHECKER2D, A.PT_MAP_UV)][0]
    got = F.texture_record(cb)
    assert got['kind'] == want['kind'] and got['mapping'] == want['mapping']
    np.testing.assert_array_equal(got['m'], want['m'])
    np.testing.assert_array_equal(got['v'], want['v'])
    mat = _make('MatteMaterial', dict(Kd=cb, sigma=const(0.), bump_map=None))
    kind, params = F.material_params(mat)
    assert kind == 'matte' and isinstance(params['Kd'], dict)
    scale = _make('ScaleTexture', dict(tex1=const([.9, .9, .2]), tex2=_make('ConstantTexture', dict(value=.5))))
    np.testing.assert_allclose(F._tex(scale), [.45, .45, .1])
    with pytest.raises(NotImplementedError):      # CylindricalMapping2D raises AttributeError in the reference itself
        F.texture_record(_make('UVTexture', dict(mapping=_make('CylindricalMapping2D', dict(w2t=None)))))
    with pytest.raises(NotImplementedError):      # a noise texture
        F.texture_record(_make('FBmTexture', dict()))
    with pytest.raises(NotImplementedError):      # roughness must be constant
        F.material_params(_make('PlasticMaterial', dict(Kd=const([.1, .1, .1]), Ks=const([.1, .1, .1]), roughness=cb,
                                                        bump_map=None)))


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
    if kind == A.PT_INTEG_PATH or not name.startswith('tex2mini'):
        assert st.rays_closest == int((kinds == 0).sum()) and st.rays_any == int((kinds == 1).sum())
    else:
        # Direct / Whitted on glass: the reference also traces specular_transmit subtrees whose radiance it then
        # drops (Q5, see test_specmini.py); those rays are not traced here
        assert len(d['L']) <= st.rays_closest <= int((kinds == 0).sum())
    # the textures matter: with every texture replaced by its first child the image changes
    plain = dict(d)
    for k in [k for k in d if '_tex_' in k]:
        del plain[k]
    for i in range(int(d['n_materials'])):
        for p in ('Kd', 'Ks', 'Kr', 'Kt'):
            if 'mat%d_%s_tex_kind' % (i, p) in d:
                plain['mat%d_%s' % (i, p)] = np.full(3, .5)
    acc2, L2 = O.OracleScene(pa.flat_scene_from_npz(plain)).render(cam, film, smp, pa.integrator_desc(kind, 5, True), tile,
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
    # production streams on the textured scene, a larger render: device == oracle
    smp2 = pa.sampler_desc(0, 24, 0, 24, 3, 3, True, 17)
    tile2 = pa.tile_desc(0, 24, 0, 24, 0, 9)
    acc2 = torch.zeros((film.y_cnt, film.x_cnt, 4), dtype=torch.float64, device='cuda')
    ctx.render(cam, film, smp2, pa.integrator_desc(kind, 5, False), tile2, acc2)
    o2, _ = O.OracleScene(flat).render(cam, film, smp2, pa.integrator_desc(kind, 5, False), tile2, want_L=False)
    a2 = acc2.cpu().numpy()
    assert np.abs(a2 - o2).max() <= 1e-9 * np.abs(o2).max()


def test_mesh_uvs_change_the_frames(golden_dir):
    """The uv fixture without its tri_uv array renders differently: the uvs reach dpdu/dpdv (BSDF frame) and the
    texture lookups.  And flatten reads mesh.uvs by index-list slot like Triangle.get_uvs (triangle.py:129-131)."""
    d = load(golden_dir, 'uvmini_direct_24x24x1.npz')
    assert d['tri_uv'].shape == (len(d['tri_verts']), 6)
    flat = pa.flat_scene_from_npz(d)
    assert flat.desc().tri_uv
    cam, film, smp, tile = descs(d)
    integ = pa.integrator_desc(A.PT_INTEG_DIRECT, 5, True)
    _, L = O.OracleScene(flat).render(cam, film, smp, integ, tile, table=d['samples'])
    plain = {k: v for k, v in d.items() if k != 'tri_uv'}
    _, L2 = O.OracleScene(pa.flat_scene_from_npz(plain)).render(cam, film, smp, integ, tile, table=d['samples'])
    assert rel_err(L, d['L']) < 1e-8 and rel_err(L2, d['L']) > 1e-3
    # stand-in mesh: two triangles, uvs per index slot
    mesh = _make('TriangleMesh', dict(uvs=[0., 0., 1., 0., 1., 1., .2, .3, .4, .5, .6, .9], n=None, s=None, alphaTexture=None,
                                      p=[np.zeros(3), np.array([1., 0, 0]), np.array([0., 1, 0]), np.array([0., 0, 1])],
                                      vertexIndex=[0, 1, 2, 1, 2, 3]))
    mat = _make('MatteMaterial', dict(Kd=_make('ConstantTexture', dict(value=np.full(3, .5))),
                                      sigma=_make('ConstantTexture', dict(value=0.)), bump_map=None))
    prims = [_make('GeometricPrimitive', dict(shape=_make('Triangle', dict(mesh=mesh, v=3 * k, ro=False,
                                                                             transform_swaps_handedness=False)),
                                              material=mat, areaLight=None)) for k in range(2)]
    nodes = [_make('Node', dict(bounds=_make('BBox', dict(pMin=np.zeros(3), pMax=np.ones(3))), offset=0, n_prim=2, axis=0))]
    scene = _make('Scene', dict(aggregate=_make('BVH', dict(nodes=nodes, primitives=prims)), lights=[]))
    flat = F.flatten_scene(scene)
    np.testing.assert_array_equal(flat.tri_uv, [[0., 0., 1., 0., 1., 1.], [.2, .3, .4, .5, .6, .9]])
    mesh.uvs = mesh.uvs[:8]
    with pytest.raises(ValueError):
        F.flatten_scene(scene)
