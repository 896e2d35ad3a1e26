"""can the reference run with Spectrum = SampledSpectrum if the alias of spectral/spectrum.py:610 is
swapped in process (no reference file edited)?  Build container only.  Outcome (recorded in DESIGN.md section 9): every
module then sees SampledSpectrum, SampledSpectrum.init() works, but SamplerRenderer.li creates `Spectrum(0.)`
(renderer/renderer.py:63) and SampledSpectrum.__new__ accepts only None or an iterable (spectrum.py:285-290): TypeError
at the first sample.  The shipped example scenes fail earlier (`Spectrum(10.)`, examples/head_mesh.py:29)."""
import sys, collections, collections.abc, traceback
for n in ('Iterable', 'Sequence', 'MutableSequence', 'Mapping', 'Callable'):
    setattr(collections, n, getattr(collections.abc, n))
import numpy as np
if not hasattr(np, 'float'): np.float = float
if not hasattr(np, 'int'): np.int = int
sys.path.insert(0, '/root/reference'); sys.dont_write_bytecode = True
import pytracer, pytracer.spectral.spectrum as sp, pytracer.spectral as spk
sp.SampledSpectrum.init()
sp.Spectrum = spk.Spectrum = pytracer.Spectrum = sp.SampledSpectrum
import pytracer.reflection
sys.modules['pytracer.reflection.bdf.bdf'].UINT = lambda v: np.uint32(int(v) & 0xFFFFFFFF)
import pytracer.renderer.renderer as _rr
_rr.util.progress_reporter = lambda *a, **k: None
from pytracer.geometry import *
from pytracer.transform import Transform
from pytracer.shape import Sphere, create_triangle_mesh
from pytracer.light import DiffuseAreaLight
from pytracer.texture import ConstantTexture
from pytracer.material import MatteMaterial, UberMaterial
from pytracer.aggregate import GeometricPrimitive, BVH
from pytracer.scene import Scene
S = sp.SampledSpectrum
def const(v): return S(np.full(30, float(v)))
try:
    main_trans = Transform.translate(Vector(0., 0., -60))
    main_shape = Sphere(main_trans, main_trans.inverse(), False, 3, -3, 3, 360.)
    light = DiffuseAreaLight(main_trans, const(10.), 2, main_shape)
    back_param = {'indices': [0, 1, 2, 2, 0, 3], 'P': [-20, 0, -10, 20, 0, -10, 20, 30, -10, -20, 30, -10]}
    back_mat = MatteMaterial(ConstantTexture(const(.1)), ConstantTexture(0.))
    bt = Transform.translate(Vector(0., -5., 0.))
    back = GeometricPrimitive(create_triangle_mesh(bt, bt.inverse(), False, back_param), back_mat)
    tri_param = {'indices': [0, 1, 2], 'P': [-5, 5, -14, 5, 5, -14, 0, 15, -16]}
    um = UberMaterial(ConstantTexture(const(.25)), ConstantTexture(const(.25)), ConstantTexture(const(0.)), ConstantTexture(const(.25)),
                      ConstantTexture(0.1), ConstantTexture(const(1.)), ConstantTexture(1.5))
    tt = Transform.translate(Vector(-3, 6., -15))
    tri = GeometricPrimitive(create_triangle_mesh(tt, tt.inverse(), False, tri_param), um)
    scene = Scene(BVH([tri, back]), [light], None)
    import examples.head_mesh  # noqa -- will fail; only configure_camera wanted
except Exception:
    traceback.print_exc()
try:
    from pytracer.camera import PerspectiveCamera
    from pytracer.film import ImageFilm
    from pytracer.filter import LanczosSincFilter
    from pytracer.transform import AnimatedTransform
    from pytracer.sampler import StratifiedSampler
    from pytracer.integrator import DirectLightingIntegrator, PathIntegrator
    from pytracer.renderer import SamplerRenderer
    import inspect, examples
    for integ in (DirectLightingIntegrator(), PathIntegrator(max_depth=5)):
        np.random.seed(0)
        film = ImageFilm(16, 16, LanczosSincFilter(4., 4., 3.), [0., 1., 0., 1.], '/tmp/try_sampled_spectrum.png')
        c2w = Transform.translate(Vector(-13, -1, -35))
        cam = PerspectiveCamera(AnimatedTransform(c2w, 0., c2w, 0.), [0., 1., 0., 1.], 0., 0., 0., 1e6, 90., film)
        smp = StratifiedSampler(0, 16, 0, 16, 1, 1, True, 0., 0.)
        ren = SamplerRenderer(smp, cam, integ, None)
        ren.render(scene)
        px = cam.film.pixels[8][8]
        print(type(integ).__name__, 'rendered; pixel Lxyz', px.Lxyz, px.weight_sum)
except Exception:
    traceback.print_exc()
