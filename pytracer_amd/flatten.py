"""Host-side flattening of a pyTracer-shaped object graph into the arrays the C ABI takes.

Only PUBLIC attributes of the reference classes are read:
``scene.aggregate.{nodes,primitives}``, ``prim.shape.{mesh,v,ro,transform_swaps_handedness}``,
``prim.material``, ``scene.lights``, ``camera.{r2c,c2w,dxCam,dyCam,lens_rad,focal_dist,film}``,
``film.*``, ``sampler.*``.  The same structure can be loaded from the ``tests/golden`` fixture
format (``flat_scene_from_npz``), which is how the GPU box — where the reference does not
exist — gets its scenes.

Materials are evaluated at flatten time into the ordered lobe list ``Material.get_bsdf`` would push
(reference pytracer/material/material.py:122-141, 166-189, 208-222, 280-299, 329-364).  Parameters are
ConstantTextures (or Scale/Mix of constants, folded here); the spectrum parameters whose lobe is pushed
of Matte (Kd), Plastic (Kd, Ks), Uber (Kd, Ks, Kr, Kt), Mirror (Kr) and Glass (Kr, Kt) may also be a
checkerboard / bilerp / UV texture (``texture_record``), evaluated per hit on the device — including the
``is_black`` tests with which Uber / Mirror / Glass decide per hit whether a lobe exists.  The FLOAT parameters
sigma of Matte and roughness of Plastic / Uber may be a float texture as well (``WrinkledTexture`` — the one noise
texture the reference can evaluate — or a checkerboard / bilerp of float constants): the lobe then carries a
parameter texture (``PT_LOBE_PARAM_TEX``) and the device derives A, B / the Blinn exponent per hit.
"""
import ctypes as C

import numpy as np

from . import _abi as A

EPS = 1e-5   # pytracer/__init__.py


# --------------------------------------------------------------------------------------
# materials → lobes
# --------------------------------------------------------------------------------------
def _clip(v):
    """util.clip / Spectrum.clip: clip to [0, inf) (utility/utility.py:110-116)."""
    return np.clip(np.asarray(v, np.float64), 0., np.inf)


def _spec(v):
    v = np.asarray(v, np.float64)
    return np.full(3, float(v)) if v.ndim == 0 else v.astype(np.float64).copy()


def _is_black(s):
    """CoefficientSpectrum.is_black (spectral/spectrum.py:182-187)."""
    return bool((np.asarray(s) < EPS).all())


def _blinn_exponent(rough):
    """Blinn.__init__ clips the exponent at 10000 (reflection/bdf/torrance.py:66-67)."""
    return float(np.clip(1. / rough, -np.inf, 10000.))


def _with_param(ex, ptex):
    """the lobe's extras tuple with a parameter-texture record merged in"""
    if ptex is None:
        return ex
    d = dict(ex[0]) if ex else {}
    d['ptex'] = ptex
    return (d,)


def _colour(v, scale=None, drop=False):
    """(R, extras, exists) of a spectrum parameter.  A constant is clipped (and scaled) here and `exists` is the
    material's own is_black test when `drop`.  A texture record (dict, see texture_record) leaves R = scale (or 1)
    and rides along as the lobe's last tuple element {'tex': record, 'drop': bool}: the colour, and with `drop`
    the lobe's existence, are decided per hit on the device."""
    if isinstance(v, dict):
        return (np.ones(3) if scale is None else np.asarray(scale, np.float64)), ({'tex': v, 'drop': bool(drop)},), True
    r = _clip(_spec(v))
    if scale is not None:
        r = np.asarray(scale, np.float64) * r
    return r, (), (not drop) or (not _is_black(r))


def lobes_from_params(kind, **p):
    """Ordered lobe list [(type, R[3], a, b, c)] for one material record.

    kind: 'matte' | 'uber' | 'plastic' | 'mirror' | 'glass' | 'metal' | 'measured'
    (a conductor lobe carries two more entries: eta[3], k[3]; a textured lobe ends with {'tex': record}, a
    measured lobe with {'brdf': (p[n,3], v[n,3])})
    """
    out = []
    textured_ok = {'matte': ('Kd', 'sigma'), 'plastic': ('Kd', 'Ks', 'roughness'),
                   'uber': ('Kd', 'Ks', 'Kr', 'Kt', 'roughness'), 'mirror': ('Kr',), 'glass': ('Kr', 'Kt')}
    for name, v in p.items():
        if isinstance(v, dict) and name not in textured_ok.get(kind, ()):
            raise NotImplementedError('a non-constant texture is supported for the colours %s of %s only, not for %s'
                                      % ('/'.join(textured_ok.get(kind, ())) or '(none)', kind, name))
    if kind == 'matte':                         # material.py:122-141
        r, ex, _ = _colour(p['Kd'])
        if isinstance(p['sigma'], dict):        # a float texture: Lambertian or Oren-Nayar, and A, B, decided per hit
            out.append((A.PT_LOBE_OREN_NAYAR, r, 0., 0., 0.) + _with_param(ex, p['sigma']))
            return out
        sigma = float(np.clip(p['sigma'], 0., 90.))
        if sigma == 0.:
            out.append((A.PT_LOBE_LAMBERT, r, 0., 0., 0.) + ex)
        else:                                   # OrenNayar.__init__, orennayar.py:36-42
            sig = np.deg2rad(sigma)
            sig_sq = sig * sig
            a = 1. - (sig_sq / (2. * (sig_sq + 0.33)))
            b = .45 * sig_sq / (sig_sq + .09)
            out.append((A.PT_LOBE_OREN_NAYAR, r, float(a), float(b), 0.) + ex)
    elif kind == 'uber':                        # material.py:329-364
        op = _clip(_spec(p['opacity']))
        if not np.array_equal(op, [1., 1., 1.]):
            out.append((A.PT_LOBE_SPEC_TRANSMIT, 1. - op, 1., 1., 0.))
        e = float(p['eta'])
        kd, ex, on = _colour(p['Kd'], op, True)
        if on:
            out.append((A.PT_LOBE_LAMBERT, kd, 0., 0., 0.) + ex)
        ks, ex, on = _colour(p['Ks'], op, True)
        if on:
            if isinstance(p['roughness'], dict):
                out.append((A.PT_LOBE_MICROFACET_BLINN, ks, e, 1., 0.) + _with_param(ex, p['roughness']))
            else:
                out.append((A.PT_LOBE_MICROFACET_BLINN, ks, e, 1., _blinn_exponent(float(p['roughness']))) + ex)
        kr, ex, on = _colour(p['Kr'], op, True)
        if on:
            out.append((A.PT_LOBE_SPEC_REFLECT, kr, e, 1., 0.) + ex)
        kt, ex, on = _colour(p['Kt'], op, True)
        if on:
            out.append((A.PT_LOBE_SPEC_TRANSMIT, kt, e, 1., 0.) + ex)
    elif kind == 'plastic':                     # material.py:166-189
        kd, exd, _ = _colour(p['Kd'])
        ks, exs, _ = _colour(p['Ks'])
        out.append((A.PT_LOBE_LAMBERT, kd, 0., 0., 0.) + exd)
        if isinstance(p['roughness'], dict):
            out.append((A.PT_LOBE_MICROFACET_BLINN, ks, 1.5, 1., 0.) + _with_param(exs, p['roughness']))
        else:
            out.append((A.PT_LOBE_MICROFACET_BLINN, ks, 1.5, 1., _blinn_exponent(float(p['roughness']))) + exs)
    elif kind == 'mirror':                      # material.py:208-222
        r, ex, on = _colour(p['Kr'], None, True)
        if on:
            out.append((A.PT_LOBE_SPEC_REFLECT, r, 1., 1., 1.) + ex)      # c != 0: FresnelFullReflect
    elif kind == 'glass':                       # material.py:280-299
        ior = float(p['index'])
        r, ex, on = _colour(p['Kr'], None, True)
        if on:
            out.append((A.PT_LOBE_SPEC_REFLECT, r, 1., ior, 0.) + ex)
        t, ex, on = _colour(p['Kt'], None, True)
        if on:
            out.append((A.PT_LOBE_SPEC_TRANSMIT, t, 1., ior, 0.) + ex)
    elif kind == 'metal':                       # material.py:242-258: Microfacet(Spectrum(1.), FresnelConductor, Blinn)
        out.append((A.PT_LOBE_MICROFACET_BLINN_COND, np.ones(3), 0., 0., _blinn_exponent(float(p['roughness'])),
                    _spec(p['eta']), _spec(p['k'])))
    elif kind == 'measured':                    # material.py:504-524: one IrIsotropicBRDF over the sample table
        bp = np.ascontiguousarray(p['brdf_p'], np.float64).reshape(-1, 3)
        bv = np.ascontiguousarray(p['brdf_v'], np.float64).reshape(-1, 3)
        if len(bp) != len(bv):
            raise ValueError('measured BRDF: %d sample positions but %d values' % (len(bp), len(bv)))
        out.append((A.PT_LOBE_MEASURED_IRISO, np.ones(3), 0., 0., 0., {'brdf': (bp, bv)}))
    else:
        raise NotImplementedError('material kind %r is outside the supported hot path' % (kind,))
    if len(out) > A.PT_MAX_LOBES:
        raise ValueError('more than %d lobes' % A.PT_MAX_LOBES)
    return out


def _const(t):
    """Value of a texture that is constant over the surface: ConstantTexture (texture/__init__.py:41-46),
    ScaleTexture (:49-61) and MixTexture (:64-82) of such; None otherwise."""
    nm = type(t).__name__
    if nm == 'ConstantTexture':
        return t.value
    if nm == 'ScaleTexture':
        a, b = _const(t.tex1), _const(t.tex2)
        return None if a is None or b is None else np.asarray(a, np.float64) * np.asarray(b, np.float64)
    if nm == 'MixTexture':
        a, b, m = _const(t.tex1), _const(t.tex2), _const(t.amt)
        if a is None or b is None or m is None:
            return None
        return (1. - m) * np.asarray(a, np.float64) + m * np.asarray(b, np.float64)
    return None


def _mapping_record(mp):
    nm = type(mp).__name__
    m = np.zeros(16)
    if nm == 'UVMapping2D':
        m[0:4] = [mp.su, mp.sv, mp.du, mp.dv]
        return A.PT_MAP_UV, m
    if nm == 'PlannarMapping2D':
        m[0:3], m[3:6], m[6], m[7] = np.asarray(mp.vs, np.float64), np.asarray(mp.vt, np.float64), mp.ds, mp.dt
        return A.PT_MAP_PLANAR, m
    if nm == 'SphericalMapping2D':
        return A.PT_MAP_SPHERICAL, np.asarray(mp.w2t.m, np.float64).ravel().copy()
    if nm == 'IdentityMapping3D':
        return A.PT_MAP_IDENTITY3D, np.asarray(mp.w2t.m, np.float64).ravel().copy()
    # CylindricalMapping2D calls self.arctan2 and raises AttributeError in the reference (texturemap:107)
    raise NotImplementedError('texture mapping %s is outside the supported hot path' % nm)


def texture_record(t, clip=True):
    """{'kind', 'mapping', 'm'[16], 'v'[4][3]} of a texture over constant children (include/pytracer_hip.h
    pt_texture).  Spectrum textures (clip=True): children are clipped here — util.clip of the selected child
    (material.py:131) is the selection of the clipped children.  Float textures (clip=False: sigma, roughness) are
    not clipped by the materials; a float child c is stored as (c, c, c) and the device reads the first component."""
    nm = type(t).__name__
    v = np.zeros((4, 3))

    def child(x):
        c = _const(x)
        if c is None:
            raise NotImplementedError('textures nested in %s must be constant' % nm)
        if not clip and np.ndim(c) != 0:
            raise NotImplementedError('a float parameter needs a float texture: %s has a spectrum child' % nm)
        return _clip(_spec(c)) if clip else _spec(c)
    scale = 1.
    if not clip and nm == 'ScaleTexture':       # ScaleTexture(constant, WrinkledTexture): tex1(dg) * tex2(dg), :49-61
        kids = [t.tex1, t.tex2]
        consts = [_const(k) for k in kids]
        var = [k for k, c in zip(kids, consts) if c is None]
        if len(var) != 1 or type(var[0]).__name__ != 'WrinkledTexture' or np.ndim([c for c in consts if c is not None][0]) != 0:
            raise NotImplementedError('a varying ScaleTexture must be a float constant times a WrinkledTexture')
        scale = float([c for c in consts if c is not None][0])
        t = var[0]
        nm = type(t).__name__
    if not clip and nm in ('BilerpTexture', 'UVTexture'):
        raise NotImplementedError('%s is a spectrum texture: not usable as a float parameter' % nm)
    if nm == 'WrinkledTexture':                 # texture/__init__.py:252-266: turbulence over Perlin noise, a float
        if clip:
            raise NotImplementedError('WrinkledTexture is a float texture: sigma / roughness, not a colour')
        mk, m = _mapping_record(t.mapping)
        if mk != A.PT_MAP_IDENTITY3D:
            raise NotImplementedError('WrinkledTexture needs a 3D mapping')
        if not 0. <= float(t.octaves) <= 64.:
            raise NotImplementedError('WrinkledTexture with %r octaves' % (t.octaves,))
        v[0, 0], v[0, 1], v[1, 0] = float(t.omega), float(t.octaves), scale
        return dict(kind=int(A.PT_TEX_WRINKLED), mapping=int(mk), m=m, v=v)
    if nm == 'Chekcerboard2DTexture':
        kind = A.PT_TEX_CHECKER2D           # 'closedform' == 'none': materials zero the differentials
        v[0], v[1] = child(t.tex1), child(t.tex2)
    elif nm == 'Checkboard3DTexture':
        kind = A.PT_TEX_CHECKER3D
        v[0], v[1] = child(t.tex1), child(t.tex2)
    elif nm == 'BilerpTexture':
        kind = A.PT_TEX_BILERP
        v[:] = [_spec(t.v00), _spec(t.v01), _spec(t.v10), _spec(t.v11)]
    elif nm == 'UVTexture':
        kind = A.PT_TEX_UV
    else:
        raise NotImplementedError('texture %s is outside the supported hot path' % nm)
    mk, m = _mapping_record(t.mapping)
    if (kind == A.PT_TEX_CHECKER3D) != (mk == A.PT_MAP_IDENTITY3D):
        raise NotImplementedError('%s needs a %s mapping' % (nm, '3D' if kind == A.PT_TEX_CHECKER3D else '2D'))
    return dict(kind=int(kind), mapping=int(mk), m=m, v=v)


def _tex(t, spectrum=False, float_tex=False):
    """Evaluate a surface-constant texture; with spectrum=True (float_tex=True) a supported varying spectrum (float)
    texture gives its record."""
    c = _const(t)
    if c is not None:
        return c
    if spectrum:
        return texture_record(t)
    if float_tex:
        return texture_record(t, clip=False)
    raise NotImplementedError('only constant textures are supported for this parameter, got %s' % type(t).__name__)


def material_params(mat):
    """(kind, params) of a reference Material instance, read by attribute."""
    nm = type(mat).__name__
    if getattr(mat, 'bump_map', None) is not None:
        raise NotImplementedError('bump mapping is outside the supported hot path')
    if nm == 'MatteMaterial':
        return 'matte', dict(Kd=_tex(mat.Kd, True), sigma=_tex(mat.sigma, float_tex=True))
    if nm == 'UberMaterial':
        return 'uber', dict(Kd=_tex(mat.Kd, True), Ks=_tex(mat.Ks, True), Kr=_tex(mat.Kr, True), Kt=_tex(mat.Kt, True),
                            roughness=_tex(mat.roughness, float_tex=True), opacity=_tex(mat.opacity), eta=_tex(mat.eta))
    if nm == 'PlasticMaterial':
        return 'plastic', dict(Kd=_tex(mat.Kd, True), Ks=_tex(mat.Ks, True), roughness=_tex(mat.roughness, float_tex=True))
    if nm == 'MirrorMaterial':
        return 'mirror', dict(Kr=_tex(mat.Kr, True))
    if nm == 'GlassMaterial':
        return 'glass', dict(Kr=_tex(mat.Kr, True), Kt=_tex(mat.Kt, True), index=_tex(mat.index))
    if nm == 'MetalMaterial':
        return 'metal', dict(eta=_tex(mat.eta), k=_tex(mat.k), roughness=_tex(mat.roughness))
    if nm == 'MeasuredMaterial':                # material.py:403-524
        if getattr(mat, 'regular_data', None) is not None:
            raise NotImplementedError('regular half-angle (MERL) data: the reference cannot load it (material.py:495)')
        smp = getattr(mat, 'theta_phi_samples', None)
        if getattr(mat, 'theta_phi_tree', None) is None:
            raise ValueError('MeasuredMaterial without data: its BSDF has no lobe')
        if smp is None:
            # second MeasuredMaterial of one file: __init__ stores the cached samples under a misspelt name
            # (material.py:434), so IrIsotropicBRDF.data is None and the first f() raises TypeError
            raise ValueError('MeasuredMaterial built from the IrIsotropicData cache has no samples '
                             '(material.py:434); the reference fails at the first hit')
        return 'measured', dict(brdf_p=np.array([np.asarray(x.p, np.float64) for x in smp]).reshape(-1, 3),
                                brdf_v=np.array([np.asarray(x.v, np.float64) for x in smp]).reshape(-1, 3))
    raise NotImplementedError('material %s is outside the supported hot path' % nm)


def _pack_material(lobes, textures=None, brdfs=None):
    """pt_material of a lobe list; textured lobes register their record in `textures` (a list), measured lobes
    their (p, v) sample table in `brdfs` (a list; the lobe gets first sample and count)."""
    m = A.pt_material()
    m.n_lobes = len(lobes)
    for i, lobe in enumerate(lobes):
        extras = lobe[-1] if isinstance(lobe[-1], dict) else None
        if extras is not None:
            lobe = lobe[:-1]
        tp, R, a, b, c = lobe[:5]
        m.lobes[i].type = tp
        m.lobes[i].R[:] = [float(x) for x in R]
        m.lobes[i].a, m.lobes[i].b, m.lobes[i].c = float(a), float(b), float(c)
        if len(lobe) > 5:
            m.lobes[i].eta[:] = [float(x) for x in lobe[5]]
            m.lobes[i].k[:] = [float(x) for x in lobe[6]]
        if extras is not None and 'brdf' in extras:
            if brdfs is None:
                raise ValueError('measured lobe without a sample table list')
            m.lobes[i].a = float(sum(len(t[0]) for t in brdfs))
            m.lobes[i].b = float(len(extras['brdf'][0]))
            brdfs.append(extras['brdf'])
        elif extras is not None:
            if textures is None:
                raise ValueError('textured lobe without a texture table')
            if 'tex' in extras:
                textures.append(extras['tex'])
                m.lobes[i].tex = len(textures) | (A.PT_LOBE_DROP_IF_BLACK if extras.get('drop') else 0)
            if 'ptex' in extras:                  # the lobe's float parameter (PT_LOBE_PARAM_TEX)
                textures.append(extras['ptex'])
                if len(textures) > 0x7FF:
                    raise ValueError('more than 2047 textures before a parameter texture')
                m.lobes[i].tex |= len(textures) << A.PT_LOBE_PARAM_TEX_SHIFT
    return m


def _pack_texture(rec):
    t = A.pt_texture()
    t.kind, t.mapping = int(rec['kind']), int(rec['mapping'])
    t.m[:] = [float(x) for x in np.asarray(rec['m'], np.float64).ravel()]
    v = np.asarray(rec['v'], np.float64).reshape(4, 3)
    for i in range(4):
        t.v[i][:] = [float(x) for x in v[i]]
    return t


# --------------------------------------------------------------------------------------
# lights
# --------------------------------------------------------------------------------------
def _rows12(m):
    return [float(x) for x in np.asarray(m, np.float64)[:3, :].ravel()]


def _pack_light(rec):
    lt = A.pt_light()
    kind = rec['kind']
    lt.kind = kind
    lt.ns = max(1, int(rec.get('ns', 1)))
    eye = np.eye(4)
    lt.w2o[:] = _rows12(eye)
    lt.o2w[:] = _rows12(eye)
    if kind == A.PT_LIGHT_DIFFUSE_SPHERE:
        r = float(rec['radius'])
        full = (abs(rec['zmin'] + r) == 0. and abs(rec['zmax'] - r) == 0. and
                abs(rec['phimax'] - 2. * np.pi) < 1e-15)
        if not full:
            raise NotImplementedError('partial spheres are outside the supported hot path')
        lt.ro = int(rec.get('ro', 0))
        lt.emit[:] = [float(x) for x in _spec(rec['emit'])]
        o2w = np.asarray(rec['o2w'], np.float64)
        lt.pos[:] = [float(o2w[i, 3]) for i in range(3)]      # o2w(Point(0,0,0)); affine
        lt.radius = r
        lt.w2o[:] = _rows12(rec['w2o'])
        lt.o2w[:] = _rows12(o2w)
    elif kind == A.PT_LIGHT_SPOT:
        lt.emit[:] = [float(x) for x in _spec(rec['intensity'])]
        lt.pos[:] = [float(x) for x in rec['pos']]
        lt.w2o[:] = _rows12(rec['w2l'])
        lt.cos_width = float(rec['cos_width'])
        lt.cos_falloff = float(rec['cos_falloff'])
    elif kind == A.PT_LIGHT_POINT:
        lt.emit[:] = [float(x) for x in _spec(rec['intensity'])]
        lt.pos[:] = [float(x) for x in rec['pos']]
    else:
        raise NotImplementedError('light kind %r' % (kind,))
    return lt


def light_record(light):
    """Plain-dict record of a reference Light instance, read by attribute."""
    nm = type(light).__name__
    if nm == 'DiffuseAreaLight':
        shapes = light.shape_set.shapes
        if len(shapes) != 1 or type(shapes[0]).__name__ != 'Sphere':
            raise NotImplementedError('DiffuseAreaLight over anything but one Sphere is outside the supported hot path')
        sp = shapes[0]
        return dict(kind=A.PT_LIGHT_DIFFUSE_SPHERE, emit=np.array(light.emit), ns=light.ns, radius=sp.radius,
                    o2w=np.array(sp.o2w.m), w2o=np.array(sp.w2o.m), ro=bool(sp.ro), zmin=sp.zmin, zmax=sp.zmax,
                    phimax=sp.phiMax)
    if nm == 'SpotLight':
        return dict(kind=A.PT_LIGHT_SPOT, pos=np.array(light.pos), intensity=np.array(light.intensity), ns=light.ns,
                    w2l=np.array(light.w2l.m), cos_width=light.cos_width, cos_falloff=light.cos_falloff)
    if nm == 'PointLight':
        return dict(kind=A.PT_LIGHT_POINT, pos=np.array(light.pos), intensity=np.array(light.intensity), ns=light.ns)
    raise NotImplementedError('light %s is outside the supported hot path' % nm)


# --------------------------------------------------------------------------------------
# the flat scene
# --------------------------------------------------------------------------------------
class FlatScene(object):
    """Arrays of one scene in the layout pt_scene_desc points at (all host memory)."""

    def __init__(self, node_bounds, node_meta, tri_verts, tri_mat, tri_flip, material_lobes, light_records, tri_uv=None):
        self.node_bounds = np.ascontiguousarray(node_bounds, np.float64).reshape(-1, 6)
        self.node_meta = np.ascontiguousarray(node_meta, np.int32).reshape(-1, 3)
        self.tri_verts = np.ascontiguousarray(tri_verts, np.float64).reshape(-1, 9)
        self.tri_mat = np.ascontiguousarray(tri_mat, np.int32).ravel()
        self.tri_flip = np.ascontiguousarray(tri_flip, np.int32).ravel()
        # corner (u,v) per triangle as Triangle.get_uvs returns them, or None for the defaults (triangle.py:125-131)
        self.tri_uv = None if tri_uv is None else np.ascontiguousarray(tri_uv, np.float64).reshape(-1, 6)
        self.material_lobes = list(material_lobes)
        self.light_records = list(light_records)
        n, p = len(self.node_bounds), len(self.tri_verts)
        if len(self.node_meta) != n or len(self.tri_mat) != p or len(self.tri_flip) != p or \
                (self.tri_uv is not None and len(self.tri_uv) != p):
            raise ValueError('inconsistent scene array lengths')
        if p and (self.tri_mat.min() < 0 or self.tri_mat.max() >= max(1, len(self.material_lobes))):
            raise ValueError('material index out of range')
        self._check_tree()
        self.textures = []                     # records referenced by textured lobes, in lobe order
        brdfs = []                             # (p, v) tables of measured lobes, in lobe order
        self._materials = (A.pt_material * max(1, len(self.material_lobes)))(
            *[_pack_material(l, self.textures, brdfs) for l in self.material_lobes])
        self.brdf_p = np.ascontiguousarray(np.concatenate([t[0] for t in brdfs]) if brdfs else np.zeros((0, 3)))
        self.brdf_v = np.ascontiguousarray(np.concatenate([t[1] for t in brdfs]) if brdfs else np.zeros((0, 3)))
        self._textures = (A.pt_texture * max(1, len(self.textures)))(*[_pack_texture(t) for t in self.textures])
        self._lights = (A.pt_light * max(1, len(self.light_records)))(*[_pack_light(r) for r in self.light_records])

    def _check_tree(self):
        """Every index the kernels will dereference is validated here, on the host."""
        n, p = len(self.node_bounds), len(self.tri_verts)
        m = self.node_meta
        if n == 0:
            return
        leaf = m[:, 1] > 0
        if (m[:, 1] < 0).any():
            raise ValueError('negative n_prim')
        lo = m[leaf, 0]
        if len(lo) and (lo.min() < 0 or (lo + m[leaf, 1]).max() > p):
            raise ValueError('leaf primitive range out of bounds')
        inner = ~leaf
        idx = np.nonzero(inner)[0]
        if len(idx):
            if (m[inner, 0] <= idx).any() or (m[inner, 0] >= n).any() or (idx + 1 >= n).any():
                raise ValueError('interior child index out of bounds')
            if (m[inner, 2] < 0).any() or (m[inner, 2] > 2).any():
                raise ValueError('split axis out of range')

    @property
    def n_nodes(self):
        return len(self.node_bounds)

    @property
    def n_tris(self):
        return len(self.tri_verts)

    @property
    def n_lights(self):
        return len(self.light_records)

    def light_ns(self):
        return [max(1, int(r.get('ns', 1))) for r in self.light_records]

    def desc(self):
        d = A.pt_scene_desc()
        d.n_nodes, d.n_tris = self.n_nodes, self.n_tris
        d.n_materials, d.n_lights = len(self.material_lobes), len(self.light_records)
        d.node_bounds = self.node_bounds.ctypes.data
        d.node_meta = self.node_meta.ctypes.data
        d.tri_verts = self.tri_verts.ctypes.data
        d.tri_mat = self.tri_mat.ctypes.data
        d.tri_flip = self.tri_flip.ctypes.data
        d.materials = C.cast(self._materials, C.POINTER(A.pt_material))
        d.lights = C.cast(self._lights, C.POINTER(A.pt_light))
        d.n_textures = len(self.textures)
        d.textures = C.cast(self._textures, C.POINTER(A.pt_texture))
        d.tri_uv = self.tri_uv.ctypes.data if self.tri_uv is not None and len(self.tri_uv) else None
        d.n_brdf_samples = len(self.brdf_p)
        d.brdf_p = self.brdf_p.ctypes.data if len(self.brdf_p) else None
        d.brdf_v = self.brdf_v.ctypes.data if len(self.brdf_v) else None
        return d

    def table_dims(self, integ_kind):
        """Length of one sample-table row (sampler/sample.py:36-76 with the requests of
        direct.py:46-62 or path.py:34-42)."""
        if integ_kind == A.PT_INTEG_PATH:
            return 35
        if integ_kind == A.PT_INTEG_WHITTED:
            return 5                            # WhittedIntegrator requests no samples (whitted.py)
        return 5 + 6 * sum(self.light_ns())


def flatten_scene(scene):
    """FlatScene from a reference-shaped ``Scene`` whose aggregate is a ``BVH`` of triangles."""
    agg = scene.aggregate
    nodes = agg.nodes or []
    prims = agg.primitives
    if getattr(agg, 'flat_nodes', None) is not None:       # pytracer_amd.accel.BVH keeps the arrays
        nb, nm = agg.flat_nodes
    else:
        nb = np.empty((len(nodes), 6), np.float64)
        nm = np.empty((len(nodes), 3), np.int32)
        for i, nd in enumerate(nodes):
            nb[i, 0:3] = nd.bounds.pMin
            nb[i, 3:6] = nd.bounds.pMax
            nm[i] = (nd.offset, nd.n_prim, nd.axis)
    tv = np.empty((len(prims), 9), np.float64)
    tm = np.empty(len(prims), np.int32)
    tf = np.empty(len(prims), np.int32)
    mats, lobes = [], []
    uv = None
    for i, pr in enumerate(prims):
        sh = pr.shape
        if type(sh).__name__ != 'Triangle':
            raise NotImplementedError('only triangles may be in the BVH (got %s)' % type(sh).__name__)
        m = sh.mesh
        # shading geometry is the geometric one unless the mesh has BOTH N and S (triangle.py:325):
        # LoopSubdiv meshes carry N only and shade like plain meshes
        if (m.n is not None and m.s is not None) or m.alphaTexture is not None:
            raise NotImplementedError('meshes with N+S shading frames / alpha are outside the supported hot path')
        if m.uvs is not None:
            # Triangle.get_uvs reads mesh.uvs by the triangle's position in the INDEX list, not through
            # vertexIndex (triangle.py:129-131): corner k of the triangle at slot v has uvs[2(v+k)], uvs[2(v+k)+1]
            if 2 * (sh.v + 2) + 1 >= len(m.uvs):
                raise ValueError('mesh.uvs is too short for triangle slot %d: the reference raises IndexError at '
                                 'the first hit of this triangle (triangle.py:129-131)' % sh.v)
            if uv is None:
                uv = np.tile(np.array([0., 0., 1., 0., 1., 1.]), (len(prims), 1))
            uv[i] = [m.uvs[2 * (sh.v + k) + c] for k in range(3) for c in range(2)]
        if getattr(pr, 'areaLight', None) is not None:
            raise NotImplementedError('emissive primitives are outside the supported hot path')
        for k in range(3):
            tv[i, 3 * k:3 * k + 3] = m.p[m.vertexIndex[sh.v + k]]
        tf[i] = int(bool(sh.ro) ^ bool(sh.transform_swaps_handedness))
        mid = next((j for j, mm in enumerate(mats) if mm is pr.material), None)
        if mid is None:
            mats.append(pr.material)
            kind, params = material_params(pr.material)
            lobes.append(lobes_from_params(kind, **params))
            mid = len(mats) - 1
        tm[i] = mid
    return FlatScene(nb, nm, tv, tm, tf, lobes, [light_record(l) for l in scene.lights], tri_uv=uv)


_MAT_KINDS = {0: 'matte', 1: 'uber', 2: 'plastic', 3: 'mirror', 4: 'glass', 5: 'metal', 6: 'measured'}


def flat_scene_from_npz(d):
    """FlatScene from the fixture format written by tools/oracle/gen_fixtures.py:dump_scene."""
    lobes = []
    for i in range(int(d['n_materials'])):
        pre = 'mat%d_' % i
        params = {k[len(pre):]: d[k] for k in d.keys() if k.startswith(pre) and k != pre + 'kind'}
        # a textured parameter P is stored as P_tex_kind / P_tex_mapping / P_tex_m / P_tex_v
        for name in [k[:-len('_tex_kind')] for k in list(params) if k.endswith('_tex_kind')]:
            params[name] = dict(kind=int(params.pop(name + '_tex_kind')), mapping=int(params.pop(name + '_tex_mapping')),
                                m=params.pop(name + '_tex_m'), v=params.pop(name + '_tex_v'))
        lobes.append(lobes_from_params(_MAT_KINDS[int(d[pre + 'kind'])], **params))
    lights = []
    for i in range(int(d['n_lights'])):
        pre = 'light%d_' % i
        lights.append({k[len(pre):]: (d[k].item() if d[k].ndim == 0 else d[k]) for k in d.keys() if k.startswith(pre)})
    return FlatScene(d['node_bounds'], d['node_meta'], d['tri_verts'], d['tri_mat'], d['tri_flip'], lobes, lights,
                     tri_uv=d['tri_uv'] if 'tri_uv' in d else None)


# --------------------------------------------------------------------------------------
# camera / film / sampler / integrator descriptors
# --------------------------------------------------------------------------------------
def camera_desc(r2c, c2w, dx, dy, lens_rad=0., focal_dist=0., kind=A.PT_CAM_PERSPECTIVE):
    c = A.pt_camera()
    c.kind = int(kind)
    c.r2c[:] = [float(x) for x in np.asarray(r2c, np.float64).ravel()]
    c.c2w[:] = [float(x) for x in np.asarray(c2w, np.float64).ravel()]
    c.dx[:] = [float(x) for x in dx]
    c.dy[:] = [float(x) for x in dy]
    c.lens_rad, c.focal_dist = float(lens_rad), float(focal_dist)
    return c


def camera_desc_from_object(camera):
    kinds = {'PerspectiveCamera': A.PT_CAM_PERSPECTIVE, 'OrthoCamera': A.PT_CAM_ORTHO}
    if type(camera).__name__ not in kinds:
        raise NotImplementedError('only PerspectiveCamera and OrthoCamera are on the supported hot path '
                                  '(the reference itself cannot render with EnvironmentCamera)')
    c2w = camera.c2w
    if getattr(c2w, 'animated', False):
        raise NotImplementedError('animated camera transforms are outside the supported hot path')
    return camera_desc(camera.r2c.m, c2w.startTransform.m, camera.dxCam, camera.dyCam,
                       camera.lens_rad, camera.focal_dist, kinds[type(camera).__name__])


def film_desc(x_res, y_res, win, xw, yw, table):
    f = A.pt_film_desc()
    f.x_res, f.y_res = int(x_res), int(y_res)
    f.x_start, f.x_cnt, f.y_start, f.y_cnt = [int(v) for v in win]
    f.xw, f.yw = float(xw), float(yw)
    t = np.asarray(table, np.float64)
    if t.shape != (16, 16):
        raise ValueError('filter table must be 16x16 (FILTER_TABLE_SIZE, pytracer/__init__.py)')
    f.table[:] = [float(x) for x in t.ravel()]
    return f


def film_desc_from_object(film):
    return film_desc(film.xResolution, film.yResolution,
                     (film.xPixel_start, film.xPixel_cnt, film.yPixel_start, film.yPixel_cnt),
                     film.filter.xw, film.filter.yw, film.filter_table)


def camera_film_from_npz(d):
    cam = camera_desc(d['cam_r2c'], d['cam_c2w'], d['cam_dx'], d['cam_dy'], float(d['cam_lens_rad']),
                      float(d['cam_focal']), int(d['cam_kind']) if 'cam_kind' in d else A.PT_CAM_PERSPECTIVE)
    film = film_desc(d['film_res'][0], d['film_res'][1], d['film_win'], d['filt_w'][0], d['filt_w'][1],
                     d['filt_table'])
    return cam, film


def sampler_desc(x_start, x_end, y_start, y_end, xs, ys, jitter=True, seed=0):
    s = A.pt_sampler_desc()
    s.x_start, s.x_end, s.y_start, s.y_end = int(x_start), int(x_end), int(y_start), int(y_end)
    s.xs, s.ys = max(int(xs), 1), max(int(ys), 1)
    s.jitter = int(bool(jitter))
    s.seed = int(seed) & 0xFFFFFFFF
    return s


def sampler_desc_from_object(sampler, seed=0):
    if type(sampler).__name__ != 'StratifiedSampler':
        raise NotImplementedError('only StratifiedSampler is on the supported hot path')
    if sampler.s_open != 0. or sampler.s_close != 0.:
        raise NotImplementedError('shutter intervals (motion blur) are outside the supported hot path')
    return sampler_desc(sampler.xPixel_start, sampler.xPixel_end, sampler.yPixel_start, sampler.yPixel_end,
                        sampler.xSamples, sampler.ySamples, sampler.jitter, seed)


def integrator_desc(kind, max_depth=5, pretest=False, shade_f32=False):
    i = A.pt_integrator_desc()
    i.kind, i.max_depth, i.pretest = int(kind), int(max_depth), int(bool(pretest))
    i.shade_f32 = int(bool(shade_f32))          # production streams only (include/pytracer_hip.h)
    return i


def integrator_desc_from_object(integ, pretest=False, shade_f32=False):
    nm = type(integ).__name__
    if nm == 'DirectLightingIntegrator':
        return integrator_desc(A.PT_INTEG_DIRECT, integ.max_depth, pretest, shade_f32)
    if nm == 'PathIntegrator':
        return integrator_desc(A.PT_INTEG_PATH, integ.max_depth, pretest, shade_f32)
    if nm == 'WhittedIntegrator':
        return integrator_desc(A.PT_INTEG_WHITTED, integ.max_depth, pretest, shade_f32)
    raise NotImplementedError('integrator %s is outside the supported hot path' % nm)


def tile_desc(x0, x1, y0, y1, s0, s1, y_step=1, y_phase=0):
    t = A.pt_tile()
    t.x0, t.x1, t.y0, t.y1, t.s0, t.s1 = int(x0), int(x1), int(y0), int(y1), int(s0), int(s1)
    t.y_step, t.y_phase = int(y_step), int(y_phase)
    if t.y_step < 0 or t.y_phase < 0 or (t.y_phase and t.y_phase >= max(t.y_step, 1)):
        raise ValueError('tile row interleave: need 0 <= y_phase < y_step')
    return t
