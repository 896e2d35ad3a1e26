"""Reference-SHAPED stand-in objects (same class names and public attributes as the reference's
Scene / BVH / GeometricPrimitive / Triangle / TriangleMesh / materials / lights / camera / film /
sampler / integrators), built from a tests/golden scene fixture.  They let the drop-in layer
(pytracer_amd.flatten, GpuSamplerRenderer) be exercised where the reference is not importable
(the GPU box).  No behaviour is implemented here — only the attributes the flattener reads."""
import numpy as np


class _NS(object):
    def __init__(self, **kw):
        self.__dict__.update(kw)


def _cls(name, **kw):
    return type(name, (_NS,), {})(**kw)


class ConstantTexture(object):
    def __init__(self, value):
        self.value = value

    def __call__(self, dg):
        return self.value


def build_scene(d):
    """Scene stand-in from the dump_scene fixture format."""
    mats = []
    for i in range(int(d['n_materials'])):
        pre = 'mat%d_' % i
        kind = int(d[pre + 'kind'])
        if kind == 0:
            mats.append(_cls('MatteMaterial', Kd=ConstantTexture(d[pre + 'Kd']),
                             sigma=ConstantTexture(float(d[pre + 'sigma'])), bump_map=None))
        elif kind == 2:
            mats.append(_cls('PlasticMaterial', Kd=ConstantTexture(d[pre + 'Kd']), Ks=ConstantTexture(d[pre + 'Ks']),
                             roughness=ConstantTexture(float(d[pre + 'roughness'])), bump_map=None))
        elif kind == 6:      # a loaded MeasuredMaterial: samples with .p / .v and the tree over them (material.py:463-485)
            smp = np.array([_cls('IrIsotropicBRDFSample', p=p_, v=v_) for p_, v_ in zip(d[pre + 'brdf_p'], d[pre + 'brdf_v'])],
                           dtype=object)
            mats.append(_cls('MeasuredMaterial', bump_map=None, regular_data=None, theta_phi_samples=smp,
                             theta_phi_tree=_cls('cKDTree', data=d[pre + 'brdf_p'])))
        else:
            mats.append(_cls('UberMaterial', bump_map=None,
                             **{k: ConstantTexture(d[pre + k] if d[pre + k].ndim else float(d[pre + k]))
                                for k in ('Kd', 'Ks', 'Kr', 'Kt', 'roughness', 'opacity', 'eta')}))
    # one mesh holding every triangle's vertices (3 per triangle, unshared)
    tv = d['tri_verts']
    P = [tv[i, 3 * k:3 * k + 3] for i in range(len(tv)) for k in range(3)]
    mesh = _cls('TriangleMesh', p=P, vertexIndex=list(range(len(P))), uvs=None, n=None, s=None, alphaTexture=None)
    prims = []
    for i in range(len(tv)):
        flip = bool(d['tri_flip'][i])
        sh = _cls('Triangle', mesh=mesh, v=3 * i, ro=flip, transform_swaps_handedness=False)
        prims.append(_cls('GeometricPrimitive', shape=sh, material=mats[int(d['tri_mat'][i])], areaLight=None))
    nodes = []
    for b, m in zip(d['node_bounds'], d['node_meta']):
        nodes.append(_cls('_LinearNode', bounds=_cls('BBox', pMin=b[0:3], pMax=b[3:6]), offset=int(m[0]),
                          n_prim=int(m[1]), axis=int(m[2])))
    lights = []
    for i in range(int(d['n_lights'])):
        pre = 'light%d_' % i
        kind = int(d[pre + 'kind'])
        if kind == 0:
            sp = _cls('Sphere', radius=float(d[pre + 'radius']), o2w=_cls('Transform', m=d[pre + 'o2w']),
                      w2o=_cls('Transform', m=d[pre + 'w2o']), ro=bool(d[pre + 'ro']), zmin=float(d[pre + 'zmin']),
                      zmax=float(d[pre + 'zmax']), phiMax=float(d[pre + 'phimax']))
            lights.append(_cls('DiffuseAreaLight', emit=d[pre + 'emit'], ns=int(d[pre + 'ns']),
                               shape_set=_cls('ShapeSet', shapes=[sp])))
        elif kind == 1:
            lights.append(_cls('SpotLight', pos=d[pre + 'pos'], intensity=d[pre + 'intensity'], ns=int(d[pre + 'ns']),
                               w2l=_cls('Transform', m=d[pre + 'w2l']), cos_width=float(d[pre + 'cos_width']),
                               cos_falloff=float(d[pre + 'cos_falloff'])))
        else:
            lights.append(_cls('PointLight', pos=d[pre + 'pos'], intensity=d[pre + 'intensity'], ns=int(d[pre + 'ns'])))
    agg = _cls('BVH', nodes=nodes, primitives=prims)
    return _cls('Scene', aggregate=agg, lights=lights)


class _Pixel(object):
    def __init__(self):
        self.Lxyz = np.zeros(3)
        self.weight_sum = 0.


class ImageFilm(object):
    def __init__(self, d):
        self.xResolution, self.yResolution = int(d['film_res'][0]), int(d['film_res'][1])
        self.xPixel_start, self.xPixel_cnt, self.yPixel_start, self.yPixel_cnt = [int(v) for v in d['film_win']]
        self.filter = _cls('LanczosSincFilter', xw=float(d['filt_w'][0]), yw=float(d['filt_w'][1]))
        self.filter_table = d['filt_table']
        self.pixels = np.empty([self.xPixel_cnt, self.yPixel_cnt], dtype=object)
        for i in range(self.xPixel_cnt):
            for j in range(self.yPixel_cnt):
                self.pixels[i][j] = _Pixel()
        self.written = 0

    def write_image(self, splat_scale=1.):
        self.written += 1


def build_camera(d):
    film = ImageFilm(d)
    return _cls('PerspectiveCamera', r2c=_cls('Transform', m=d['cam_r2c']),
                c2w=_cls('AnimatedTransform', startTransform=_cls('Transform', m=d['cam_c2w']), animated=False),
                dxCam=d['cam_dx'], dyCam=d['cam_dy'], lens_rad=float(d['cam_lens_rad']),
                focal_dist=float(d['cam_focal']), film=film)


def build_sampler(xs, xe, ys, ye, nx, ny, jitter=True):
    return _cls('StratifiedSampler', xPixel_start=xs, xPixel_end=xe, yPixel_start=ys, yPixel_end=ye, xSamples=nx,
                ySamples=ny, jitter=jitter, s_open=0., s_close=0., spp=nx * ny)


def build_integrator(kind, max_depth=5):
    return _cls('PathIntegrator' if kind == 'path' else 'DirectLightingIntegrator', max_depth=max_depth)
