"""§8(f) rank 1: native Loop subdivision (pt_loop_subdiv, host code) against the reference's own
LoopSubdiv.refine outputs (tests/golden/loop_subdiv.npz, tools/oracle/gen_fixtures.py:gen_loop):
vertex positions, vertex order and triangle indices bit for bit, at 1-3 levels, for meshes with
boundary rules, valence-4 and valence-5 extraordinary vertices; the 3-level head (89 078 vertices) is
pinned by digest in tests/test_c2_head_loop.py."""
import os

import numpy as np
import pytest

from pytracer_amd import accel


@pytest.fixture(scope='module')
def fx(golden_dir):
    return dict(np.load(os.path.join(golden_dir, 'loop_subdiv.npz')))


@pytest.mark.parametrize('name', ['octa', 'grid', 'bipyramid'])
@pytest.mark.parametrize('levels', [1, 2, 3])
def test_equals_reference(fx, name, levels):
    P, idx = accel.loop_subdivide(fx[name + '_P'], fx[name + '_idx'], levels)
    assert np.array_equal(idx, fx['%s_l%d_idx' % (name, levels)])
    assert np.array_equal(P, fx['%s_l%d_P' % (name, levels)])


def test_counts_and_level_zero(fx):
    P0, i0 = fx['grid_P'], fx['grid_idx']
    P, idx = accel.loop_subdivide(P0, i0, 0)              # no split: only the push to the limit surface
    assert np.array_equal(idx, i0) and P.shape == P0.shape and not np.array_equal(P, P0)
    for lv in (1, 2):
        P, idx = accel.loop_subdivide(P0, i0, lv)
        assert len(idx) == len(i0) * 4 ** lv and idx.max() == len(P) - 1 and idx.min() == 0
    # head control mesh: every level multiplies the faces by 4 (Euler: V' = V + E)
    Ph, ih = accel.loop_subdivide(fx['head_P'], fx['head_idx'], 1)
    assert len(ih) == 4 * len(fx['head_idx'])


def test_fails_where_the_reference_fails(fx):
    # a closed mesh: LoopSubdiv.refine raises UnboundLocalError in its tangent pass (loopsubdiv.py:409)
    octa = [0, 2, 4, 2, 1, 4, 1, 3, 4, 3, 0, 4, 2, 0, 5, 1, 2, 5, 3, 1, 5, 0, 3, 5]
    with pytest.raises(ValueError, match='UnboundLocalError'):
        accel.loop_subdivide(fx['octa_P'], octa, 1)
    # a vertex no face uses: AttributeError on startFace (loopsubdiv.py:235)
    with pytest.raises(ValueError, match='no face'):
        accel.loop_subdivide(np.concatenate([fx['grid_P'], [[9., 9., 9.]]]), fx['grid_idx'], 1)
    with pytest.raises(ValueError, match='out of range'):
        accel.loop_subdivide(fx['grid_P'], [0, 1, 99], 1)
    with pytest.raises(ValueError):
        accel.loop_subdivide(fx['grid_P'], [0, 1], 1)


@pytest.mark.skipif(not os.path.isdir('/root/reference'), reason='reference not present (GPU box)')
def test_closed_mesh_really_fails_in_the_reference(fx):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r'''
import sys
sys.path.insert(0, %r)
import _shim
import pytracer.geometry as geo
from pytracer.shape import create_loop_subdiv
from pytracer.transform import Transform
t = Transform.translate(geo.Vector(0., 0., 0.))
P = [1,0,0, -1,0,0, 0,1,0, 0,-1,0, 0,0,1, 0,0,-1]
idx = [0,2,4, 2,1,4, 1,3,4, 3,0,4, 2,0,5, 1,2,5, 3,1,5, 0,3,5]
try:
    create_loop_subdiv(t, t.inverse(), False, {'indices': idx, 'P': [float(x) for x in P], 'nlevels': 1}).refine()
except UnboundLocalError:
    print('UNBOUND')
''' % os.path.join(root, 'tools', 'oracle')
    out = subprocess.run([sys.executable, '-W', 'ignore', '-c', code], capture_output=True, text=True, timeout=300,
                         env=dict(os.environ, PYTHONDONTWRITEBYTECODE='1'))
    assert 'UNBOUND' in out.stdout, (out.stdout + out.stderr)[-1500:]


@pytest.mark.skipif(not os.path.isdir('/root/reference'), reason='reference not present (GPU box)')
def test_dropin_create_loop_subdiv_equals_reference_refine(fx):
    """accel.create_loop_subdiv + accel.BVH in a reference scene script: same TriangleMesh, same tree."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r'''
import sys, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r)
import _shim
import pytracer.geometry as geo
from pytracer.shape import create_loop_subdiv
from pytracer.aggregate import BVH, GeometricPrimitive
from pytracer.transform import Transform
from pytracer.material import MatteMaterial
from pytracer.texture import ConstantTexture
from pytracer.spectral import Spectrum
from pytracer_amd import accel
d = np.load(%r)
params = {'indices': [int(k) for k in d['grid_idx']], 'P': [float(x) for x in d['grid_P'].ravel()], 'nlevels': 2}
t = Transform.translate(geo.Vector(1., -2., 3.))
ref = create_loop_subdiv(t, t.inverse(), False, params).refine()[0]
mine = accel.create_loop_subdiv(t, t.inverse(), False, params)
assert type(mine).__name__ == 'TriangleMesh' and mine.ntris == ref.ntris and mine.nverts == ref.nverts
assert np.array_equal(np.array(mine.vertexIndex), np.array(ref.vertexIndex))
assert np.array_equal(np.array([np.array(p) for p in mine.p]), np.array([np.array(p) for p in ref.p]))
mat = MatteMaterial(ConstantTexture(Spectrum([.5, .5, .5])), ConstantTexture(0.))
a = BVH([GeometricPrimitive(create_loop_subdiv(t, t.inverse(), False, params), mat)])
b = accel.BVH([GeometricPrimitive(mine, mat)])
assert len(a.nodes) == len(b.nodes)
for x, y in zip(a.nodes, b.nodes):
    assert (int(x.offset), int(x.n_prim), int(x.axis)) == (y.offset, y.n_prim, y.axis)
    assert np.array_equal(np.array(x.bounds.pMin), y.bounds.pMin)
assert [p.shape.v for p in a.primitives] == [p.shape.v for p in b.primitives]
print('OK')
''' % (os.path.join(root, 'tools', 'oracle'), root, os.path.join(root, 'tests', 'golden', 'loop_subdiv.npz'))
    out = subprocess.run([sys.executable, '-W', 'ignore', '-c', code], capture_output=True, text=True, timeout=300,
                         env=dict(os.environ, PYTHONDONTWRITEBYTECODE='1'))
    assert out.returncode == 0 and 'OK' in out.stdout, (out.stdout + out.stderr)[-2000:]
