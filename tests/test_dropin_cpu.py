"""Host logic of the drop-in layer, on CPU: flattening the REAL reference objects (when the
reference is present in this container), flattening reference-shaped stand-ins, and the
multi-GPU shard + reduce logic rehearsed with world_size-2 gloo (the oracle plays the device)."""
import os
import sys

import numpy as np
import pytest

import pytracer_amd as pa
from pytracer_amd import _abi as A
from pytracer_amd.renderer import shard_tile
import standin

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = '/root/reference'


def test_flatten_standin_equals_fixture(golden_dir):
    d = dict(np.load(os.path.join(golden_dir, 'head_mesh_scene.npz')))
    a = pa.flat_scene_from_npz(d)
    b = pa.flatten_scene(standin.build_scene(d))
    for k in ('node_bounds', 'node_meta', 'tri_verts', 'tri_mat', 'tri_flip'):
        assert np.array_equal(getattr(a, k), getattr(b, k)), k
    assert repr(a.material_lobes) == repr(b.material_lobes)
    assert bytes(a._lights) == bytes(b._lights)


@pytest.mark.skipif(not os.path.isdir(REF), reason='reference not present (GPU box)')
def test_flatten_real_reference_scene_equals_fixture(golden_dir):
    """Runs the reference's own examples/head_mesh.py scene through flatten_scene (subprocess: the
    import shim patches numpy/collections globally)."""
    import subprocess
    code = r'''
import sys, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r)
import _shim
import examples.head_mesh as model
import pytracer_amd as pa
from pytracer_amd import flatten as F
flat = pa.flatten_scene(model.scene)
d = dict(np.load(%r))
ref = pa.flat_scene_from_npz(d)
for k in ('node_bounds', 'node_meta', 'tri_verts', 'tri_mat', 'tri_flip'):
    assert np.array_equal(getattr(flat, k), getattr(ref, k)), k
assert repr(flat.material_lobes) == repr(ref.material_lobes)
assert bytes(flat._lights) == bytes(ref._lights)
cam = model.configure_camera(x_res=128, y_res=128, file_name='/tmp/_x.png')
c = F.camera_desc_from_object(cam)
assert np.array_equal(np.array(c.r2c).reshape(4, 4), d['cam_r2c'])
f = F.film_desc_from_object(cam.film)
assert np.array_equal(np.array(f.table).reshape(16, 16), d['filt_table'])
from pytracer.sampler import StratifiedSampler
from pytracer.integrator import PathIntegrator
s = F.sampler_desc_from_object(StratifiedSampler(0, 128, 0, 128, 2, 2, True, 0., 0.))
assert (s.x_end, s.xs, s.ys, s.jitter) == (128, 2, 2, 1)
assert F.integrator_desc_from_object(PathIntegrator(max_depth=5)).kind == 1
print('OK')
''' % (os.path.join(ROOT, 'tools', 'oracle'), ROOT, os.path.join(golden_dir, 'head_mesh_scene.npz'))
    out = subprocess.run([sys.executable, '-W', 'ignore', '-c', code], capture_output=True, text=True, timeout=300,
                         env=dict(os.environ, PYTHONDONTWRITEBYTECODE='1'))
    assert out.returncode == 0 and 'OK' in out.stdout, out.stderr[-2000:]


@pytest.mark.skipif(not os.path.isdir(REF), reason='reference not present (GPU box)')
def test_flatten_real_measured_material(golden_dir, tmp_path):
    """The reference's own MeasuredMaterial, loaded from a .brdf file written here (the fixture generator's synthetic
    format), goes through flatten's material walk; its IrIsotropicBRDF.f equals the oracle's lobe on the flattened
    table; and a second MeasuredMaterial of the same file (the reference's cache stores its samples under a misspelt
    name, material.py:434) is refused."""
    import subprocess
    code = r'''
import sys, os, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r); sys.path.insert(0, %r)
import _shim
import gen_fixtures as G
import pytracer.geometry as geo
from pytracer.material import MeasuredMaterial
from pytracer.spectral.spectrum import SampledSpectrum
from pytracer.reflection import IrIsotropicBRDF
import pytracer_amd as pa
from pytracer_amd import flatten as F, _abi as A
import oracle_lib as O
SampledSpectrum.init()
path = G.write_synthetic_brdf(os.path.join(%r, 'dropin_test.brdf'), 120, 3)
mat = MeasuredMaterial(path)
kind, params = F.material_params(mat)
assert kind == 'measured' and params['brdf_p'].shape == (120, 3)
assert np.array_equal(params['brdf_p'], np.asarray(mat.theta_phi_tree.data))
d = dict(np.load(%r))
for k in [k for k in d if k.startswith('mat0_')]:
    del d[k]
d['mat0_kind'] = np.int32(6); d['mat0_brdf_p'] = params['brdf_p']; d['mat0_brdf_v'] = params['brdf_v']
osc = O.OracleScene(pa.flat_scene_from_npz(d))
b = IrIsotropicBRDF(mat.theta_phi_tree, mat.theta_phi_samples)
rng = np.random.default_rng(1)
for _ in range(40):
    wo, wi = rng.normal(size=3), rng.normal(size=3)
    wo /= np.linalg.norm(wo); wi /= np.linalg.norm(wi); wo[2] = abs(wo[2]); wi[2] = abs(wi[2])
    want = np.array(b.f(geo.Vector(*wo), geo.Vector(*wi)))
    got = osc.lobe_eval(0, 0, wo, wi, .3, .7)[:3]
    assert np.allclose(got, want, rtol=1e-10, atol=0), (got, want)
again = MeasuredMaterial(path)
try:
    F.material_params(again)
    raise SystemExit('second MeasuredMaterial of one file was accepted')
except ValueError:
    pass
print('OK')
''' % (os.path.join(ROOT, 'tools', 'oracle'), ROOT, os.path.join(ROOT, 'tests'), str(tmp_path),
       os.path.join(golden_dir, 'head_mesh_scene.npz'))
    out = subprocess.run([sys.executable, '-W', 'ignore', '-c', code], capture_output=True, text=True, timeout=300,
                         env=dict(os.environ, PYTHONDONTWRITEBYTECODE='1'))
    assert out.returncode == 0 and 'OK' in out.stdout, out.stderr[-2000:]


@pytest.mark.skipif(not os.path.isdir(REF), reason='reference not present (GPU box)')
def test_textured_fixture_inputs_are_reference_derived(golden_dir):
    """The material / texture INPUTS of the textured fixtures are what the fixture generator reads off the live
    reference objects by itself (tools/oracle/gen_fixtures.py texture_dump: no product code on that side), they
    equal the committed fixture files, and the product's flattener (flatten.texture_record) gives the same records
    for the same objects."""
    import subprocess
    code = r'''
import sys, os, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r)
import gen_fixtures as G
assert 'pytracer_amd' not in sys.modules, 'the fixture generator imports the product'
dumps = {}
for variant, fname in (('tex', 'texmini_path_24x24x4.npz'), ('tex2', 'tex2mini_path_24x24x4.npz'), ('uv', 'uvmini_path_24x24x4.npz'),
                       ('wrink', 'wrinkmini_path_24x24x4.npz')):
    model = G.C4Mini(variant)
    d = G.dump_scene(model.scene)
    assert 'pytracer_amd' not in sys.modules
    fx = dict(np.load(os.path.join(%r, fname)))
    keys = sorted(k for k in d if k.startswith('mat') or k == 'n_materials')
    assert keys == sorted(k for k in fx if k.startswith('mat') or k == 'n_materials'), (variant, keys)
    assert any('_tex_' in k for k in keys)
    for k in keys:
        assert np.array_equal(np.asarray(d[k]), fx[k]), (variant, k, d[k], fx[k])
    dumps[variant] = model
from pytracer_amd import flatten as F
n = 0
for variant, model in dumps.items():
    for pr in model.scene.aggregate.primitives:
        for name in ('Kd', 'Ks', 'Kr', 'Kt'):
            t = getattr(pr.material, name, None)
            if t is None or G.is_surface_constant(t):
                continue
            a, b = G.texture_dump(t), F.texture_record(t)
            assert a['kind'] == b['kind'] and a['mapping'] == b['mapping']
            assert np.array_equal(a['m'], b['m']) and np.array_equal(a['v'], b['v'])
            n += 1
        for name in ('sigma', 'roughness'):        # float parameters: WrinkledTexture, alone or under a ScaleTexture
            t = getattr(pr.material, name, None)
            if t is None or G.is_surface_constant(t):
                continue
            a, b = G.texture_dump(t, is_float=True), F.texture_record(t, clip=False)
            assert a['kind'] == b['kind'] == 4 and a['mapping'] == b['mapping']
            assert np.array_equal(a['m'], b['m']) and np.array_equal(a['v'], b['v'])
            n += 1
assert n > 0
print('OK')
''' % (os.path.join(ROOT, 'tools', 'oracle'), ROOT, golden_dir)
    out = subprocess.run([sys.executable, '-W', 'ignore', '-c', code], capture_output=True, text=True, timeout=300,
                         env=dict(os.environ, PYTHONDONTWRITEBYTECODE='1'))
    assert out.returncode == 0 and 'OK' in out.stdout, out.stderr[-2000:]


def test_shard_tile_partitions_exactly():
    smp = pa.sampler_desc(3, 67, 5, 50, 4, 2)
    for world in (1, 2, 3, 8):
        cover = np.zeros(8, int)
        for r in range(world):
            t = shard_tile(smp, r, world, 'samples')
            assert (t.x0, t.x1, t.y0, t.y1) == (3, 67, 5, 50)
            cover[t.s0:t.s1] += 1
        assert (cover == 1).all()
        rows = np.zeros(50, int)
        for r in range(world):
            t = shard_tile(smp, r, world, 'rows')
            assert (t.s0, t.s1) == (0, 8)
            rows[t.y0:t.y1] += 1
        assert (rows[5:50] == 1).all() and rows[:5].sum() == 0
        # interleaved rows: every row of the window exactly once, all samples, row counts within one of each other
        rows = np.zeros(50, int)
        counts = []
        for r in range(world):
            t = shard_tile(smp, r, world, 'interleaved')
            assert (t.x0, t.x1, t.s0, t.s1) == (3, 67, 0, 8)
            ys = [t.y0 + (t.y_phase if t.y_step > 1 else 0) + k * max(t.y_step, 1) for k in range(t.rows())]
            assert all(t.y0 <= y < t.y1 for y in ys)
            rows[ys] += 1
            counts.append(len(ys))
            assert t.n_samples() == len(ys) * 64 * 8
        assert (rows[5:50] == 1).all() and rows[:5].sum() == 0 and max(counts) - min(counts) <= 1
    # spp < world falls back to row bands
    smp1 = pa.sampler_desc(0, 16, 0, 16, 1, 1)
    t = shard_tile(smp1, 1, 2, 'samples')
    assert (t.y0, t.y1, t.s0, t.s1) == (8, 16, 0, 1)


def _gloo_worker(rank, world, port, golden_dir, mode, q):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import oracle_lib as O
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    d = dict(np.load(os.path.join(golden_dir, 'head_mesh_scene.npz')))
    flat = pa.flat_scene_from_npz(d)
    osc = O.OracleScene(flat)
    cam, film = pa.camera_film_from_npz(d)
    smp = pa.sampler_desc(40, 72, 40, 72, 2, 2, True, 5)
    integ = pa.integrator_desc(A.PT_INTEG_PATH, 5, False)
    tile = shard_tile(smp, rank, world, mode)
    acc, _ = osc.render(cam, film, smp, integ, tile, want_L=False)     # the oracle stands in for the device
    t = torch.from_numpy(acc)
    dist.reduce(t, dst=0)
    if rank == 0:
        whole, _ = osc.render(cam, film, smp, integ, pa.tile_desc(40, 72, 40, 72, 0, 4), want_L=False)
        q.put(float(np.abs(t.numpy() - whole).max() / np.abs(whole).max()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('mode', ['samples', 'rows', 'interleaved'])
def test_two_rank_shard_and_reduce_gloo(golden_dir, mode):
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + {'samples': 0, 'rows': 7, 'interleaved': 13}[mode]
    procs = [ctx.Process(target=_gloo_worker, args=(r, 2, port, golden_dir, mode, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    assert q.get(timeout=5) < 1e-12
