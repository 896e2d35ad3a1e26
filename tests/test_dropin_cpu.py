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


@pytest.mark.parametrize('mode', ['samples', 'rows'])
def test_two_rank_shard_and_reduce_gloo(golden_dir, mode):
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + (7 if mode == 'rows' else 0)
    procs = [ctx.Process(target=_gloo_worker, args=(r, 2, port, golden_dir, mode, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    assert q.get(timeout=5) < 1e-12
