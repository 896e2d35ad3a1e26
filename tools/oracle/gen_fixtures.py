#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by running the *reference*
pyTracer (imported read-only from /root/reference through tools/oracle/_shim.py).

Runs ONLY in the build container.  The outputs are plain .npz data files (inputs and
expected outputs), committed; neither this script nor the fixtures contain reference
source.  See SURVEY.md §8(c) F1..F8 for what each fixture pins.

    PYTHONDONTWRITEBYTECODE=1 python3 -W ignore tools/oracle/gen_fixtures.py [which ...]

which ∈ {scene, direct, path, film, bvh, spot}; default: all.
"""
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _shim  # noqa: F401,E402
import numpy as np  # noqa: E402

import pytracer.geometry as geo  # noqa: E402
from pytracer import Spectrum, EPS  # noqa: E402
from pytracer.aggregate import Intersection, BVH, GeometricPrimitive  # noqa: E402
from pytracer.sampler import StratifiedSampler, Sample  # noqa: E402
from pytracer.integrator import DirectLightingIntegrator, PathIntegrator, WhittedIntegrator  # noqa: E402
from pytracer.renderer import SamplerRenderer  # noqa: E402
from pytracer.reflection import BSDF, BDFType  # noqa: E402
import pytracer.light.light as light_mod  # noqa: E402
import pytracer.integrator.utility as iutil  # noqa: E402

OUT = os.path.normpath(os.path.join(HERE, '..', '..', 'tests', 'golden'))
os.makedirs(OUT, exist_ok=True)

MASK64 = (1 << 64) - 1


def hash_uniform(sample_id: int, k: int) -> float:
    """Counter-based uniform in [0,1): splitmix64 finaliser of (sample_id, k), top 53 bits.
    Restated identically in oracle/pt_oracle.c and csrc (pt_hash_uniform)."""
    z = ((sample_id & 0xFFFFFFFF) << 32 | (k & 0xFFFFFFFF)) & MASK64
    z = (z + 0x9E3779B97F4A7C15) & MASK64
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & MASK64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & MASK64
    z = z ^ (z >> 31)
    return (z >> 11) * (1.0 / 9007199254740992.0)


class CounterRng:
    def __init__(self):
        self.sid = 0
        self.k = 0

    def reset(self, sid):
        self.sid = sid
        self.k = 0

    def __call__(self, *a):
        assert not a
        v = hash_uniform(self.sid, self.k)
        self.k += 1
        return v


# ------------------------------------------------------------------------------------
# scene flattening (by public attributes only; same walk as pytracer_amd/flatten.py)
# ------------------------------------------------------------------------------------
LOBE_LAMBERT, LOBE_OREN_NAYAR, LOBE_MICROFACET_BLINN, LOBE_SPEC_REFLECT, LOBE_SPEC_TRANS = range(5)


def tex_value(t):
    return t.value if hasattr(t, 'value') else t(None)


# Texture / mapping codes of include/pytracer_hip.h (pt_texture.kind / .mapping), restated here so that the fixture
# INPUTS do not pass through the product's flattener (tests/test_dropin_cpu.py compares the two on live objects).
TEX_KIND = {'Chekcerboard2DTexture': 0, 'Checkboard3DTexture': 1, 'BilerpTexture': 2, 'UVTexture': 3, 'WrinkledTexture': 4}
MAP_KIND = {'UVMapping2D': 0, 'PlannarMapping2D': 1, 'SphericalMapping2D': 2, 'IdentityMapping3D': 3}
SURFACE_CONSTANT = ('ConstantTexture', 'ScaleTexture', 'MixTexture')


def is_surface_constant(t):
    nm = type(t).__name__
    if nm == 'ConstantTexture':
        return True
    if nm in ('ScaleTexture', 'MixTexture'):
        kids = [t.tex1, t.tex2] + ([t.amt] if nm == 'MixTexture' else [])
        return all(is_surface_constant(k) for k in kids)
    return False


def as_rgb(v):
    v = np.asarray(v, np.float64)
    return np.full(3, float(v)) if v.ndim == 0 else np.array(v, np.float64)


def texture_dump(t, is_float=False):
    """{'kind', 'mapping', 'm'[16], 'v'[4][3]} of a spatially varying texture, read off the reference object:
    the values of constant children are the REFERENCE's own evaluation (t(None)), clipped as the materials clip what
    they evaluate (material.py:131; float parameters — sigma, roughness — are not clipped there and not here);
    the mapping's 16 doubles are its attributes in the order pt_texture documents."""
    nm = type(t).__name__
    v = np.zeros((4, 3))
    scale = 1.
    if is_float and nm == 'ScaleTexture':      # ScaleTexture(constant, WrinkledTexture): tex1(dg) * tex2(dg)
        kids = [t.tex1, t.tex2]
        const = [k for k in kids if is_surface_constant(k)]
        var = [k for k in kids if not is_surface_constant(k)]
        assert len(const) == 1 and len(var) == 1 and type(var[0]).__name__ == 'WrinkledTexture'
        scale = float(const[0](None))
        t = var[0]
        nm = type(t).__name__

    def child(x):
        assert is_surface_constant(x), 'children of %s must be constant over the surface' % nm
        if is_float:
            return as_rgb(float(x(None)))
        return np.clip(as_rgb(x(None)), 0., np.inf)
    if nm == 'WrinkledTexture':
        assert is_float
        v[0, 0], v[0, 1], v[1, 0] = float(t.omega), float(t.octaves), scale
    elif nm in ('Chekcerboard2DTexture', 'Checkboard3DTexture'):
        v[0], v[1] = child(t.tex1), child(t.tex2)
    elif nm == 'BilerpTexture':
        for k, c in enumerate((t.v00, t.v01, t.v10, t.v11)):
            v[k] = as_rgb(c)
    elif nm != 'UVTexture':
        raise NotImplementedError(nm)
    mp = t.mapping
    mn = type(mp).__name__
    m = np.zeros(16)
    if mn == 'UVMapping2D':
        m[:4] = (mp.su, mp.sv, mp.du, mp.dv)
    elif mn == 'PlannarMapping2D':
        m[0:3], m[3:6], m[6], m[7] = np.array(mp.vs, np.float64), np.array(mp.vt, np.float64), mp.ds, mp.dt
    elif mn in ('SphericalMapping2D', 'IdentityMapping3D'):
        m[:] = np.array(mp.w2t.m, np.float64).reshape(16)
    else:
        raise NotImplementedError(mn)
    return dict(kind=TEX_KIND[nm], mapping=MAP_KIND[mn], m=m, v=v)


def spec_param(t):
    """A spectrum parameter: the value of a texture that is constant over the surface (the reference's own
    evaluation of it), or the record of a varying one."""
    if type(t).__name__ == 'ConstantTexture':
        return np.array(t.value)
    if is_surface_constant(t):
        return np.array(t(None))
    return texture_dump(t)


def float_param(t):
    """A float parameter (sigma, roughness): the reference's own value of a constant texture, or the record of a
    varying float texture."""
    if is_surface_constant(t):
        return float(t(None))
    return texture_dump(t, is_float=True)


def dump_scene(scene, camera=None):
    agg = scene.aggregate
    N = len(agg.nodes)
    bounds = np.empty((N, 6), np.float64)
    meta = np.empty((N, 3), np.int32)
    for i, nd in enumerate(agg.nodes):
        bounds[i, 0:3] = nd.bounds.pMin
        bounds[i, 3:6] = nd.bounds.pMax
        meta[i] = (nd.offset, nd.n_prim, nd.axis)
    P = len(agg.primitives)
    verts = np.empty((P, 9), np.float64)
    flip = np.empty(P, np.int32)
    matid = np.empty(P, np.int32)
    tri_src = np.empty((P, 2), np.int32)   # (mesh ordinal, triangle number in mesh)
    mats = []
    meshes = []
    uvs, has_uv = [], False
    for i, pr in enumerate(agg.primitives):
        sh = pr.shape
        m = sh.mesh
        # shading geometry == geometric unless the mesh has BOTH N and S (triangle.py:325)
        assert (m.n is None or m.s is None) and m.alphaTexture is None
        if m.uvs is not None:
            has_uv = True
        uvs.append(np.array(sh.get_uvs(), np.float64).ravel())      # the reference's own per-triangle (u,v) corners
        for k in range(3):
            verts[i, 3 * k:3 * k + 3] = m.p[m.vertexIndex[sh.v + k]]
        flip[i] = int(bool(sh.ro) ^ bool(sh.transform_swaps_handedness))
        if pr.material not in mats:
            mats.append(pr.material)
        matid[i] = mats.index(pr.material)
        if m not in meshes:
            meshes.append(m)
        tri_src[i] = (meshes.index(m), sh.v // 3)
        assert pr.areaLight is None
    d = dict(node_bounds=bounds, node_meta=meta, tri_verts=verts, tri_flip=flip, tri_mat=matid,
             tri_src=tri_src)
    if has_uv:
        d['tri_uv'] = np.array(uvs)
    # raw material parameters (ConstantTexture values)
    mrec = []
    for mt in mats:
        nm = type(mt).__name__
        if nm == 'MatteMaterial':
            mrec.append(dict(kind=0, Kd=spec_param(mt.Kd), sigma=float_param(mt.sigma)))
        elif nm == 'UberMaterial':
            mrec.append(dict(kind=1, Kd=spec_param(mt.Kd), Ks=spec_param(mt.Ks),
                             Kr=spec_param(mt.Kr), Kt=spec_param(mt.Kt),
                             roughness=float_param(mt.roughness), opacity=np.array(mt.opacity(None)),
                             eta=float(mt.eta(None))))
        elif nm == 'MetalMaterial':
            mrec.append(dict(kind=5, eta=np.array(mt.eta(None)), k=np.array(mt.k(None)),
                             roughness=float(mt.roughness(None))))
        elif nm == 'MirrorMaterial':
            mrec.append(dict(kind=3, Kr=spec_param(mt.Kr)))
        elif nm == 'GlassMaterial':
            mrec.append(dict(kind=4, Kr=spec_param(mt.Kr), Kt=spec_param(mt.Kt), index=float(mt.index(None))))
        elif nm == 'PlasticMaterial':
            mrec.append(dict(kind=2, Kd=spec_param(mt.Kd), Ks=spec_param(mt.Ks),
                             roughness=float_param(mt.roughness)))
        elif nm == 'MeasuredMaterial':
            mrec.append(dict(kind=6, brdf_p=np.array([np.array(x.p) for x in mt.theta_phi_samples]),
                             brdf_v=np.array([np.array(x.v) for x in mt.theta_phi_samples])))
        else:
            raise NotImplementedError(nm)
    d['n_materials'] = np.int32(len(mrec))
    for i, r in enumerate(mrec):
        for k, v in r.items():
            if isinstance(v, dict):            # a non-constant texture: its record (pytracer_amd.flatten.texture_record)
                for kk, vv in v.items():
                    d['mat%d_%s_tex_%s' % (i, k, kk)] = np.asarray(vv)
            else:
                d['mat%d_%s' % (i, k)] = np.asarray(v)
    # lights
    d['n_lights'] = np.int32(len(scene.lights))
    for i, lt in enumerate(scene.lights):
        nm = type(lt).__name__
        if nm == 'DiffuseAreaLight':
            assert len(lt.shape_set.shapes) == 1
            sp = lt.shape_set.shapes[0]
            assert type(sp).__name__ == 'Sphere'
            d['light%d_kind' % i] = np.int32(0)
            d['light%d_emit' % i] = np.array(lt.emit)
            d['light%d_ns' % i] = np.int32(lt.ns)
            d['light%d_radius' % i] = np.float64(sp.radius)
            d['light%d_o2w' % i] = np.array(sp.o2w.m)
            d['light%d_w2o' % i] = np.array(sp.w2o.m)
            d['light%d_ro' % i] = np.int32(bool(sp.ro))
            d['light%d_zmin' % i] = np.float64(sp.zmin)
            d['light%d_zmax' % i] = np.float64(sp.zmax)
            d['light%d_phimax' % i] = np.float64(sp.phiMax)
            d['light%d_area' % i] = np.float64(sp.area())
        elif nm == 'SpotLight':
            d['light%d_kind' % i] = np.int32(1)
            d['light%d_pos' % i] = np.array(lt.pos)
            d['light%d_intensity' % i] = np.array(lt.intensity)
            d['light%d_ns' % i] = np.int32(lt.ns)
            d['light%d_w2l' % i] = np.array(lt.w2l.m)
            d['light%d_cos_width' % i] = np.float64(lt.cos_width)
            d['light%d_cos_falloff' % i] = np.float64(lt.cos_falloff)
        elif nm == 'PointLight':
            d['light%d_kind' % i] = np.int32(2)
            d['light%d_pos' % i] = np.array(lt.pos)
            d['light%d_intensity' % i] = np.array(lt.intensity)
            d['light%d_ns' % i] = np.int32(lt.ns)
        else:
            raise NotImplementedError(nm)
    if camera is not None:
        d.update(dump_camera(camera))
    return d


def dump_camera(camera):
    film = camera.film
    kind = {'PerspectiveCamera': 0, 'OrthoCamera': 1, 'EnvironmentCamera': 2}[type(camera).__name__]
    if kind == 2:
        return dict(cam_kind=np.int32(2), cam_r2c=np.eye(4), cam_c2w=np.array(camera.c2w.startTransform.m),
                    cam_dx=np.zeros(3), cam_dy=np.zeros(3), cam_lens_rad=np.float64(0.), cam_focal=np.float64(0.),
                    film_res=np.array([film.xResolution, film.yResolution], np.int32),
                    film_win=np.array([film.xPixel_start, film.xPixel_cnt, film.yPixel_start, film.yPixel_cnt],
                                      np.int32),
                    filt_w=np.array([film.filter.xw, film.filter.yw], np.float64),
                    filt_table=np.array(film.filter_table, np.float64))
    return dict(cam_kind=np.int32(kind), cam_r2c=np.array(camera.r2c.m), cam_c2w=np.array(camera.c2w.startTransform.m),
                cam_dx=np.array(camera.dxCam), cam_dy=np.array(camera.dyCam),
                cam_lens_rad=np.float64(camera.lens_rad), cam_focal=np.float64(camera.focal_dist),
                film_res=np.array([film.xResolution, film.yResolution], np.int32),
                film_win=np.array([film.xPixel_start, film.xPixel_cnt, film.yPixel_start, film.yPixel_cnt],
                                  np.int32),
                filt_w=np.array([film.filter.xw, film.filter.yw], np.float64),
                filt_table=np.array(film.filter_table, np.float64))


def flat_sample(s):
    v = [s.imageX, s.imageY, s.lens_u, s.lens_v, s.time]
    for a in s.oneD:
        v.extend(np.asarray(a).ravel())
    for a in s.twoD:
        v.extend(np.asarray(a).ravel())
    return np.array(v, np.float64)


# ------------------------------------------------------------------------------------
# instrumented render: mirrors SamplerRendererTask.__call__ (renderer/renderer.py:132-198)
# with an injected counter rng and logging wrappers.
# ------------------------------------------------------------------------------------
class Logs:
    def __init__(self):
        self.rays = []      # (sid, kind[0 closest,1 any], o3, d3, mint, maxt)
        self.rres = []      # (hit, prim, t)
        self.hitgeo = []    # for closest hits: p3 nn3 dpdu3 dpdv3 u v rEps  (else zeros)
        self.ed = []        # estimate_direct: sid, light idx, p3 n3 wo3 rEps ul3 ub3 -> Ld3
        self.sl = []        # sample_l: p3, pEps, u3 -> Li3, wi3, pdf, ray o3 d3 mint maxt
        self.sf = []        # path sample_f(ALL): sid, wo3, u3 -> pdf, wi3, type, f3
        self.sid = -1


def instrumented_render(model, x_res, y_res, xs, ys, integrator, seed, jitter=True, sub=None):
    camera = model.configure_camera(x_res=x_res, y_res=y_res, file_name='/tmp/_fixture.png')
    scene = model.scene
    agg = scene.aggregate
    if sub is None:
        sub = (0, x_res, 0, y_res)
    sampler = StratifiedSampler(sub[0], sub[1], sub[2], sub[3], xs, ys, jitter, 0., 0.)
    renderer = SamplerRenderer(sampler, camera, integrator, None)
    np.random.seed(seed)
    logs = Logs()
    prim_index = {id(p): i for i, p in enumerate(agg.primitives)}
    rng = CounterRng()

    o_int, o_intp = type(scene).intersect, type(scene).intersect_p
    o_ed = iutil.estimate_direct
    sl_classes = (light_mod.DiffuseAreaLight, light_mod.SpotLight, light_mod.PointLight)
    o_sls = [c.sample_l for c in sl_classes]
    o_sf = BSDF.sample_f

    def w_int(self, ray, isect):
        rec = (logs.sid, 0) + tuple(ray.o) + tuple(ray.d) + (ray.mint, ray.maxt)
        hit = o_int(self, ray, isect)
        logs.rays.append(rec)
        if hit:
            dg = isect.dg
            logs.rres.append((1, prim_index[id(isect.primitive)], ray.maxt))
            logs.hitgeo.append(tuple(dg.p) + tuple(dg.nn) + tuple(dg.dpdu) + tuple(dg.dpdv) +
                               (dg.u, dg.v, isect.rEps))
        else:
            logs.rres.append((0, -1, ray.maxt))
            logs.hitgeo.append((0.,) * 15)
        return hit

    def w_intp(self, ray):
        rec = (logs.sid, 1) + tuple(ray.o) + tuple(ray.d) + (ray.mint, ray.maxt)
        hit = o_intp(self, ray)
        logs.rays.append(rec)
        logs.rres.append((int(hit), -1, ray.maxt))
        logs.hitgeo.append((0.,) * 15)
        return hit

    def w_ed(scene_, renderer_, light, p, n, wo, r_eps, time_, bsdf, light_smp, bsdf_smp, flags, rng_=None):
        Ld = o_ed(scene_, renderer_, light, p, n, wo, r_eps, time_, bsdf, light_smp, bsdf_smp, flags, rng_)
        logs.ed.append((logs.sid, scene_.lights.index(light)) + tuple(p) + tuple(n) + tuple(wo) + (r_eps,) +
                       (light_smp.u_pos[0], light_smp.u_pos[1], light_smp.u_com,
                        bsdf_smp.u_dir[0], bsdf_smp.u_dir[1], bsdf_smp.u_com) + tuple(Ld))
        return Ld

    def make_w_sl(o_sl):
        def w_sl(self, p, pEps, ls, time_):
            r = o_sl(self, p, pEps, ls, time_)
            Li, wi, pdf, vis = r
            logs.sl.append(tuple(p) + (pEps, ls.u_pos[0], ls.u_pos[1], ls.u_com) + tuple(Li) + tuple(wi) + (pdf,) +
                           tuple(vis.ray.o) + tuple(vis.ray.d) + (vis.ray.mint, vis.ray.maxt) +
                           (scene.lights.index(self),))
            return r
        return w_sl

    def w_sf(self, wo_w, smp, flags):
        r = o_sf(self, wo_w, smp, flags)
        if not isinstance(flags, BDFType):      # path.py:107 passes the bare int BDFType.ALL
            pdf, wi, tp, f = r
            wi_ = tuple(wi) if wi is not None else (0., 0., 0.)
            tpv = int(tp.v) if tp is not None else 0
            logs.sf.append((logs.sid,) + tuple(wo_w) + (smp.u_dir[0], smp.u_dir[1], smp.u_com, pdf) + wi_ +
                           (tpv,) + tuple(f))
        return r

    type(scene).intersect = w_int
    type(scene).intersect_p = w_intp
    iutil.estimate_direct = w_ed
    for c, o in zip(sl_classes, o_sls):
        c.sample_l = make_w_sl(o)
    BSDF.sample_f = w_sf
    try:
        if integrator is not None:
            integrator.preprocess(scene, camera, renderer)
        sample0 = Sample(sampler, integrator, None, scene)
        spp = sampler.maximum_sample_cnt()
        samples = sample0.duplicate(spp)
        S, R, Lout, Hit = [], [], [], []
        sid = 0
        while sampler.generate(samples):
            for sample in samples:
                logs.sid = sid
                rng.reset(sid)
                wt, ray = camera.generate_ray_differential(sample)
                ray.scale_differential(1. / np.sqrt(sampler.spp))
                S.append(flat_sample(sample))
                R.append(np.concatenate([ray.o, ray.d, ray.rxOrigin, ray.rxDirection, ray.ryOrigin,
                                         ray.ryDirection]))
                hit = scene.intersect_p(ray)
                if hit and wt > 0.:
                    isect = Intersection()
                    L, T = renderer.li(scene, ray, sample, isect, rng)
                    L = L * wt
                else:
                    L = Spectrum(0.)
                if L.has_nans():
                    L = Spectrum(0.)
                elif L.y() < -EPS:
                    L = Spectrum(0.)
                elif L.y() == np.inf:
                    L = Spectrum(0.)
                Lout.append(np.array(L))
                Hit.append(int(hit))
                camera.film.add_sample(sample, L)
                sid += 1
    finally:
        type(scene).intersect = o_int
        type(scene).intersect_p = o_intp
        iutil.estimate_direct = o_ed
        for c, o in zip(sl_classes, o_sls):
            c.sample_l = o
        BSDF.sample_f = o_sf
    film = camera.film
    W, H = film.xPixel_cnt, film.yPixel_cnt
    Lxyz = np.zeros((H, W, 3))
    wsum = np.zeros((H, W))
    for x in range(W):
        for y in range(H):
            Lxyz[y, x] = film.pixels[x][y].Lxyz
            wsum[y, x] = film.pixels[x][y].weight_sum
    img8 = film.write_image()
    out = dict(samples=np.array(S), rays=np.array(R), L=np.array(Lout), pretest=np.array(Hit, np.int32),
               film_Lxyz=Lxyz, film_wsum=wsum, img8=np.asarray(img8),
               n1D=np.array(sample0.n1D, np.int32), n2D=np.array(sample0.n2D, np.int32),
               spp=np.array([xs, ys], np.int32), sub=np.array(sub, np.int32),
               log_rays=np.array(logs.rays, np.float64).reshape(-1, 10),
               log_rres=np.array(logs.rres, np.float64).reshape(-1, 3),
               log_hitgeo=np.array(logs.hitgeo, np.float64).reshape(-1, 15),
               log_ed=np.array(logs.ed, np.float64).reshape(-1, 21),
               log_sl=np.array(logs.sl, np.float64).reshape(-1, 23)[:, :22],
               log_sl_light=np.array(logs.sl, np.float64).reshape(-1, 23)[:, 22].astype(np.int32),
               log_sf=np.array(logs.sf, np.float64).reshape(-1, 15))
    out.update(dump_camera(camera))
    return out


def gen_scene():
    import examples.head_mesh as model
    cam = model.configure_camera(x_res=128, y_res=128, file_name='/tmp/_fixture.png')
    d = dump_scene(model.scene, cam)
    np.savez_compressed(os.path.join(OUT, 'head_mesh_scene.npz'), **d)
    print('scene: nodes', d['node_bounds'].shape, 'tris', d['tri_verts'].shape)


def gen_direct():
    import examples.head_mesh as model
    t = time.time()
    d = instrumented_render(model, 40, 40, 1, 1, DirectLightingIntegrator(), seed=0)
    np.savez_compressed(os.path.join(OUT, 'head_direct_40x40x1.npz'), **d)
    print('direct 40x40x1: %.1fs, rays logged %d, sumL %.6f' % (time.time() - t, len(d['log_rays']), d['L'].sum()))
    t = time.time()
    d = instrumented_render(model, 24, 24, 2, 2, DirectLightingIntegrator(), seed=1)
    np.savez_compressed(os.path.join(OUT, 'head_direct_24x24x4.npz'), **d)
    print('direct 24x24x4: %.1fs, rays logged %d, sumL %.6f' % (time.time() - t, len(d['log_rays']), d['L'].sum()))


def gen_path():
    import examples.head_mesh as model
    t = time.time()
    d = instrumented_render(model, 32, 32, 2, 2, PathIntegrator(max_depth=5), seed=2)
    np.savez_compressed(os.path.join(OUT, 'head_path_32x32x4.npz'), **d)
    print('path 32x32x4: %.1fs, rays logged %d, sumL %.6f' % (time.time() - t, len(d['log_rays']), d['L'].sum()))


def gen_c1_window():
    """A window of the README quick-start (C1: 128^2, 1 spp, Direct): pixels 40..88 x 30..70,
    unjittered samples so the stream is reproducible without np.random."""
    import examples.head_mesh as model
    t = time.time()
    d = instrumented_render(model, 128, 128, 1, 1, DirectLightingIntegrator(), seed=0, jitter=False,
                            sub=(44, 84, 36, 76))
    np.savez_compressed(os.path.join(OUT, 'head_c1_window.npz'), **d)
    print('c1 window: %.1fs sumL %.6f' % (time.time() - t, d['L'].sum()))


def soup(n, seed, dup=0):
    rng = np.random.default_rng(seed)
    c = rng.uniform(-10, 10, (n, 3))
    e1 = rng.normal(0, 0.6, (n, 3))
    e2 = rng.normal(0, 0.6, (n, 3))
    P = np.stack([c, c + e1, c + e2], 1).reshape(-1, 3)
    if dup:
        # exact duplicates -> identical centroids -> exercises the equal-centroid leaf branch
        P = np.concatenate([P, P[:3 * dup]], 0)
    idx = np.arange(len(P), dtype=np.int64)
    return P, idx


def gen_bvh():
    from pytracer.shape import create_triangle_mesh
    from pytracer.transform import Transform
    from pytracer.material import MatteMaterial
    from pytracer.texture import ConstantTexture
    from pytracer.scene import Scene
    for name, n, seed, dup in (('soup300', 300, 11, 0), ('soup2000', 2000, 12, 0), ('soup500dup', 500, 13, 7)):
        P, idx = soup(n, seed, dup)
        tr = Transform.translate(geo.Vector(0., 0., 0.))
        mesh = create_triangle_mesh(tr, tr.inverse(), False, {'indices': list(idx), 'P': list(P.ravel())})
        mat = MatteMaterial(ConstantTexture(Spectrum([.5, .5, .5])), ConstantTexture(0.))
        t = time.time()
        bvh = BVH([GeometricPrimitive(mesh, mat)])
        sc = Scene(bvh, [], None)
        d = dump_scene(sc)
        d['in_P'] = P
        d['in_idx'] = idx
        # a ray batch against this soup (closest + any)
        rng = np.random.default_rng(seed + 100)
        nr = 400
        o = rng.uniform(-12, 12, (nr, 3))
        tgt = rng.uniform(-8, 8, (nr, 3))
        dvec = tgt - o
        rays = np.zeros((nr, 8))
        rays[:, 0:3] = o
        rays[:, 3:6] = dvec
        rays[:, 6] = 0.
        rays[:, 7] = np.where(rng.uniform(size=nr) < .5, np.inf, 1.0)
        # a few axis-parallel rays (zero direction components -> inf inv_dir)
        rays[:8, 3:6] = np.eye(3)[np.arange(8) % 3] * np.where(np.arange(8) < 4, 1., -1.)[:, None]
        prim_index = {id(p): i for i, p in enumerate(bvh.primitives)}
        res = np.zeros((nr, 4))
        for i in range(nr):
            r = geo.Ray(geo.Point(*rays[i, 0:3]), geo.Vector(*rays[i, 3:6]), rays[i, 6], rays[i, 7])
            isect = Intersection()
            hit = bvh.intersect(r, isect)
            r2 = geo.Ray(geo.Point(*rays[i, 0:3]), geo.Vector(*rays[i, 3:6]), rays[i, 6], rays[i, 7])
            anyh = bvh.intersect_p(r2)
            res[i] = (int(hit), prim_index[id(isect.primitive)] if hit else -1, r.maxt, int(anyh))
        d['rays'] = rays
        d['res'] = res
        np.savez_compressed(os.path.join(OUT, 'bvh_%s.npz' % name), **d)
        print('bvh %s: %.1fs nodes %d hits %d' % (name, time.time() - t, len(bvh.nodes), int(res[:, 0].sum())))


def write_synthetic_brdf(path, n_samples, seed, n_wl=30):
    """A synthetic irregularly sampled isotropic BRDF in the text format MeasuredMaterial reads
    (material/material.py:443-479): '#' comments, the number of wavelengths, the wavelengths, then per sample
    theta_i phi_i theta_o phi_o and one value per wavelength.  The reference ships no such file."""
    rng = np.random.default_rng(seed)
    wls = np.linspace(400., 700., n_wl)
    with open(path, 'w') as f:
        f.write('# synthetic irregular isotropic BRDF, %d samples, seed %d\n%d\n' % (n_samples, seed, n_wl))
        f.write(' '.join('%.17g' % w for w in wls) + '\n')
        for _ in range(n_samples):
            thi, tho = np.arccos(rng.uniform(0., 1., 2))
            phi, pho = rng.uniform(0., 2. * np.pi, 2)
            base = rng.uniform(.002, .04)
            tint = 1. + .6 * np.sin(wls / 50. + rng.uniform(0., 6.))
            gloss = .03 * max(0., np.cos(thi - tho)) ** 8
            v = base * tint + gloss
            f.write('%.17g %.17g %.17g %.17g ' % (thi, phi, tho, pho) + ' '.join('%.17g' % x for x in v) + '\n')
    return path


class C4Mini(object):
    """C4 in miniature: a triangle soup with alternating Matte / Plastic materials in
    front of a matte wall, lit by a SpotLight, a PointLight and a sphere DiffuseAreaLight; camera at
    (0,0,-30) looking +z, fov 60."""

    def __init__(self, variant='lights'):
        from pytracer.shape import Sphere, create_triangle_mesh
        from pytracer.transform import Transform
        from pytracer.material import (MatteMaterial, PlasticMaterial, MirrorMaterial, GlassMaterial, UberMaterial,
                                       MetalMaterial)
        from pytracer.texture import ConstantTexture
        from pytracer.light import SpotLight, PointLight, DiffuseAreaLight
        from pytracer.scene import Scene
        rng = np.random.default_rng(4242)
        n = 180
        c = np.concatenate([rng.uniform(-9, 9, (n, 2)), rng.uniform(-5, 5, (n, 1))], axis=1)
        e1 = rng.normal(0, 1.6, (n, 3))
        e2 = rng.normal(0, 1.6, (n, 3))
        tri = np.stack([c, c + e1, c + e2], 1)
        ident = Transform.translate(geo.Vector(0., 0., 0.))
        self.in_tris = []
        prims = []
        mats = [MatteMaterial(ConstantTexture(Spectrum([.6, .3, .2])), ConstantTexture(0.)),
                PlasticMaterial(ConstantTexture(Spectrum([.2, .4, .6])), ConstantTexture(Spectrum([.3, .3, .3])),
                                ConstantTexture(0.05)),
                MatteMaterial(ConstantTexture(Spectrum([.5, .5, .5])), ConstantTexture(20.))]    # Oren-Nayar wall
        nmesh = 2
        if variant == 'xtra':
            c3 = lambda v: ConstantTexture(Spectrum(v))
            mats = [MetalMaterial(c3([.2, .9, 1.1]), c3([3.9, 2.4, 2.2]), ConstantTexture(.08)),
                    MirrorMaterial(c3([.9, .8, .7])),
                    GlassMaterial(c3([.8, .8, .8]), c3([.7, .9, .7]), ConstantTexture(1.5)),
                    MatteMaterial(c3([.5, .5, .5]), ConstantTexture(0.))]
            nmesh = 3
        if variant == 'tex':
            # §8(f) rank 3, textures: checkerboards (2D closed-form and point-sampled, 3D), bilerp and UV textures
            # through UV, planar, spherical and 3D identity mappings, in Matte Kd / Plastic Kd and Ks
            from pytracer.texture.texture import (Chekcerboard2DTexture, Checkboard3DTexture, BilerpTexture, UVTexture,
                                                  ScaleTexture, MixTexture)
            from pytracer.texture.texturemap import UVMapping2D, PlannarMapping2D, SphericalMapping2D, IdentityMapping3D
            c3 = lambda v: ConstantTexture(Spectrum(v))
            sp = lambda v: Spectrum(v)
            mats = [MatteMaterial(Chekcerboard2DTexture(UVMapping2D(6., 5., .25, .1), c3([.8, .2, .2]), c3([.1, .1, .7])),
                                  ConstantTexture(0.)),
                    PlasticMaterial(Checkboard3DTexture(IdentityMapping3D(Transform.scale(.5, .4, .3)), c3([.2, .6, .3]),
                                                        ScaleTexture(c3([.9, .9, .2]), ConstantTexture(.5))),
                                    BilerpTexture(SphericalMapping2D(Transform.translate(geo.Vector(1., -2., 3.))),
                                                  sp([.1, .1, .1]), sp([.6, .2, .2]), sp([.2, .6, .2]), sp([.3, .3, .7])),
                                    ConstantTexture(0.05)),
                    MatteMaterial(UVTexture(UVMapping2D(3., 2., .5, .5)), ConstantTexture(15.)),
                    MatteMaterial(Chekcerboard2DTexture(PlannarMapping2D(geo.Vector(.25, 0., 0.), geo.Vector(0., .2, 0.), .3, .7),
                                                        MixTexture(c3([.9, .9, .9]), c3([.5, .5, .5]), ConstantTexture(.25)),
                                                        c3([.15, .15, .15]), 'none'),
                                  ConstantTexture(0.))]
            nmesh = 3
        if variant == 'tex2':
            # textured colours of the materials that decide PER HIT which lobes exist (`if not k.is_black()`):
            # a mirror whose Kr checkerboard has black checks (no lobe there), glass with a textured Kt,
            # an Uber material with opacity < 1 and checkerboards in Kd (one check black) and Kr
            from pytracer.texture.texture import Chekcerboard2DTexture, Checkboard3DTexture, UVTexture
            from pytracer.texture.texturemap import UVMapping2D, PlannarMapping2D, IdentityMapping3D
            c3 = lambda v: ConstantTexture(Spectrum(v))
            mats = [MirrorMaterial(Chekcerboard2DTexture(UVMapping2D(4., 4., 0., 0.), c3([.9, .8, .7]), c3([0., 0., 0.]))),
                    GlassMaterial(c3([.8, .8, .8]),
                                  Checkboard3DTexture(IdentityMapping3D(Transform.scale(.4, .4, .4)), c3([.7, .9, .7]),
                                                      c3([0., 0., 0.])), ConstantTexture(1.5)),
                    UberMaterial(Chekcerboard2DTexture(UVMapping2D(3., 3., .5, 0.), c3([.6, .4, .2]), c3([0., 0., 0.]), 'none'),
                                 c3([.2, .2, .2]),
                                 Chekcerboard2DTexture(PlannarMapping2D(geo.Vector(.5, 0., 0.), geo.Vector(0., .5, 0.), 0., 0.),
                                                       c3([.5, .5, .5]), c3([0., 0., 0.])),
                                 c3([0., 0., 0.]), ConstantTexture(.2), c3([.8, .8, .8]), ConstantTexture(1.33)),
                    MatteMaterial(c3([.5, .5, .5]), ConstantTexture(0.))]
            nmesh = 3
        if variant == 'wrink':
            # the one noise texture the reference can evaluate (FBm / Windy / Marble raise UnboundLocalError, Dots
            # TypeError): WrinkledTexture, a FLOAT texture, as Matte sigma and as Plastic / Uber roughness
            from pytracer.texture.texture import WrinkledTexture, ScaleTexture, Chekcerboard2DTexture
            from pytracer.texture.texturemap import IdentityMapping3D, UVMapping2D
            c3 = lambda v: ConstantTexture(Spectrum(v))
            self.wrinkled = [WrinkledTexture(3, .5, IdentityMapping3D(Transform.scale(.7, .9, 1.1))),
                             WrinkledTexture(4, .6, IdentityMapping3D(Transform.translate(geo.Vector(3.3, -1.7, .4)))),
                             WrinkledTexture(2, .45, IdentityMapping3D(Transform.scale(1.7, 1.7, 1.7)))]
            mats = [MatteMaterial(c3([.6, .3, .2]), ScaleTexture(ConstantTexture(40.), self.wrinkled[0])),
                    PlasticMaterial(Chekcerboard2DTexture(UVMapping2D(4., 4., 0., 0.), c3([.2, .4, .6]), c3([.5, .2, .2])),
                                    c3([.3, .3, .3]), ScaleTexture(self.wrinkled[1], ConstantTexture(.2))),
                    UberMaterial(c3([.3, .2, .2]), c3([.25, .25, .25]), c3([0., 0., 0.]), c3([0., 0., 0.]),
                                 self.wrinkled[2], c3([1., 1., 1.]), ConstantTexture(1.33)),
                    MatteMaterial(c3([.5, .5, .5]), self.wrinkled[0])]
            nmesh = 3
        if variant == 'spec':
            # specular chains: a mirror, glass, and an Uber material with a specular-reflection component
            c3 = lambda v: ConstantTexture(Spectrum(v))
            mats = [MirrorMaterial(c3([.9, .8, .7])),
                    GlassMaterial(c3([.8, .8, .8]), c3([.7, .9, .7]), ConstantTexture(1.5)),
                    UberMaterial(c3([.3, .2, .2]), c3([.2, .2, .2]), c3([.5, .5, .5]), c3([0., 0., 0.]),
                                 ConstantTexture(.2), c3([1., 1., 1.]), ConstantTexture(1.33)),
                    MatteMaterial(c3([.5, .5, .5]), ConstantTexture(0.))]
            nmesh = 3
        if variant == 'uv':
            # meshes WITH uv coordinates (dpdu/dpdv, hence the BSDF frame, and the texture lookups follow them);
            # every 7th triangle has a degenerate parametrisation (det == 0: coordinate_system fallback)
            from pytracer.texture.texture import Chekcerboard2DTexture, UVTexture
            from pytracer.texture.texturemap import UVMapping2D
            c3 = lambda v: ConstantTexture(Spectrum(v))
            mats = [MatteMaterial(Chekcerboard2DTexture(UVMapping2D(5., 5., 0., 0.), c3([.8, .3, .2]), c3([.2, .3, .8])),
                                  ConstantTexture(25.)),
                    PlasticMaterial(UVTexture(UVMapping2D(2., 2., .25, .25)), c3([.3, .3, .3]), ConstantTexture(0.05)),
                    MatteMaterial(c3([.5, .5, .5]), ConstantTexture(20.))]
        if variant == 'meas':
            # §8(f) rank 3, measured BRDF: two MeasuredMaterials over synthetic irregular isotropic tables
            # (a dense one and a sparse one whose balls have to grow), Plastic, and a matte wall
            from pytracer.material import MeasuredMaterial
            from pytracer.spectral.spectrum import SampledSpectrum
            SampledSpectrum.init()                  # CIE tables for Spectrum.from_sampled (material.py:473)
            c3 = lambda v: ConstantTexture(Spectrum(v))
            mats = [MeasuredMaterial(write_synthetic_brdf('/tmp/_fixture_dense.brdf', 600, 7)),
                    MeasuredMaterial(write_synthetic_brdf('/tmp/_fixture_sparse.brdf', 40, 8)),
                    PlasticMaterial(c3([.2, .4, .6]), c3([.3, .3, .3]), ConstantTexture(0.05)),
                    MatteMaterial(c3([.5, .5, .5]), ConstantTexture(0.))]
            nmesh = 3
        for k in range(nmesh):
            P = tri[k::nmesh].reshape(-1, 3)
            mp = {'indices': list(range(len(P))), 'P': list(P.ravel())}
            if variant == 'uv':
                uvr = np.random.default_rng(99 + k).uniform(0., 1., (len(P) // 3, 3, 2))
                uvr[::7, 1] = uvr[::7, 0]
                uvr[::7, 2] = uvr[::7, 0]
                mp['uv'] = list(uvr.ravel())
            mesh = create_triangle_mesh(ident, ident.inverse(), False, mp)
            prims.append(GeometricPrimitive(mesh, mats[k]))
            self.in_tris.append(P.reshape(-1, 9))
        wall = {'indices': [0, 1, 2, 2, 0, 3], 'P': [-30, -30, 9, 30, -30, 9, 30, 30, 9, -30, 30, 9]}
        prims.append(GeometricPrimitive(create_triangle_mesh(ident, ident.inverse(), False, wall), mats[nmesh]))
        w = np.array(wall['P'], float).reshape(-1, 3)
        self.in_tris.append(w[np.array(wall['indices'])].reshape(-1, 9))
        # +z of the light frame towards the origin (Transform.look_at mixes up rows and columns, so compose)
        spot_l2w = Transform.translate(geo.Vector(0., 15., -25.)) * Transform.rotate_x(np.rad2deg(np.arctan2(15., 25.)))
        st = Transform.translate(geo.Vector(-10., 10., -20.))
        sph = Sphere(st, st.inverse(), False, 3, -3, 3, 360.)
        lights = [SpotLight(spot_l2w, Spectrum(200.), 30., 25.),
                  PointLight(Transform.translate(geo.Vector(8., -6., -12.)), Spectrum(150.)),
                  DiffuseAreaLight(st, Spectrum(10.), 2, sph)]
        if variant in ('spec', 'xtra', 'tex', 'tex2', 'uv', 'meas', 'wrink'):
            lights = lights[1:]
        self.scene = Scene(BVH(prims), lights, None)

    camera_kind = 'perspective'

    def configure_camera(self, x_res=50, y_res=50, file_name='tmp.png'):
        from pytracer.transform import Transform, AnimatedTransform
        from pytracer.filter import LanczosSincFilter
        from pytracer.film import ImageFilm
        from pytracer.camera import PerspectiveCamera, OrthoCamera, EnvironmentCamera
        fil = LanczosSincFilter(4., 4., 3.)
        film = ImageFilm(x_res, y_res, fil, [0., 1., 0., 1.], file_name)
        c_trans = Transform.translate(geo.Vector(0., 0., -30.))
        at = AnimatedTransform(c_trans, 0., c_trans, 0.)
        k = self.camera_kind
        if k == 'perspective':
            return PerspectiveCamera(at, [0., 1., 0., 1.], 0., 0., 0., 1e100, 60., film)
        if k == 'perspective_lens':       # thin lens: camera/projective.py:196-210
            return PerspectiveCamera(at, [-1., 1., -1., 1.], 0., 0., .8, 30., 60., film)
        if k == 'ortho':
            return OrthoCamera(at, [-12., 12., -12., 12.], 0., 0., 0., 1e100, film)
        if k == 'ortho_lens':
            return OrthoCamera(at, [-12., 12., -12., 12.], 0., 0., .6, 30., film)
        if k == 'environment':
            e_trans = Transform.translate(geo.Vector(0., 0., -8.))
            return EnvironmentCamera(AnimatedTransform(e_trans, 0., e_trans, 0.), 0., 0., film)
        raise ValueError(k)


def gen_c4mini():
    model = C4Mini()
    cam = model.configure_camera(x_res=24, y_res=24, file_name='/tmp/_fixture.png')
    sc = dump_scene(model.scene, cam)
    for k, t in enumerate(model.in_tris):
        sc['in_tris%d' % k] = t
    for name, integ, xs in (('direct', DirectLightingIntegrator(), 1), ('path', PathIntegrator(max_depth=5), 2)):
        t = time.time()
        d = instrumented_render(model, 24, 24, xs, xs, integ, seed=7)
        d.update(sc)
        np.savez_compressed(os.path.join(OUT, 'c4mini_%s_24x24x%d.npz' % (name, xs * xs)), **d)
        print('c4mini %s: %.1fs, rays logged %d, sumL %.6f, zero-L %d of %d, spot/point/sphere sample_l %s' % (
            name, time.time() - t, len(d['log_rays']), d['L'].sum(), int((d['L'].sum(1) == 0).sum()), len(d['L']),
            np.bincount(d['log_sl_light'], minlength=3)))


def gen_c2():
    """C2 (BASELINE.json configs[1]): examples/head_loop.py — the head mesh after 3 levels of Loop
    subdivision (177 666 triangles), 512x512, 4x4 spp, DirectLightingIntegrator.  Importing the example
    runs the reference's subdivision and its Python BVH build (minutes).  The fixture holds
    digests of the subdivided mesh (the builder's INPUT), the backdrop mesh, digests of the reference's tree (352 k nodes are too big to
    commit), reference-traced rays, and a reference render of a small window."""
    import hashlib
    t = time.time()
    import examples.head_loop as model
    print('c2: scene built by the reference in %.1fs' % (time.time() - t), flush=True)
    bvh = model.scene.aggregate
    sc = dump_scene(model.scene)
    d = {}
    for k in ('node_bounds', 'node_meta', 'tri_verts', 'tri_mat', 'tri_flip'):
        d['sha256_' + k] = np.frombuffer(hashlib.sha256(np.ascontiguousarray(sc[k]).tobytes()).digest(), np.uint8)
    d['n_nodes'] = np.int64(len(sc['node_bounds']))
    d['n_tris'] = np.int64(len(sc['tri_verts']))
    for k, v in sc.items():
        if k.startswith('mat') or k.startswith('light') or k in ('n_materials', 'n_lights'):
            d[k] = v
    # the meshes as the reference refined them, in the order of BVH([head, back])
    meshes = []
    for pr in bvh.primitives:
        if pr.shape.mesh not in meshes:
            meshes.append(pr.shape.mesh)
    order = sorted(range(len(meshes)), key=lambda i: -meshes[i].ntris)      # head first, backdrop second
    for j, i in enumerate(order):
        m = meshes[i]
        mp = np.array([np.array(p) for p in m.p], np.float64)
        mi = np.array(m.vertexIndex, np.int32)
        if j == 0:
            # the subdivided head (89 078 vertices): digests only — pt_loop_subdiv reproduces it bit for
            # bit from the control mesh in loop_subdiv.npz (tests/test_loop_subdiv_cpu.py)
            d['sha256_mesh0_p'] = np.frombuffer(hashlib.sha256(mp.tobytes()).digest(), np.uint8)
            d['sha256_mesh0_idx'] = np.frombuffer(hashlib.sha256(mi.tobytes()).digest(), np.uint8)
            d['mesh0_nverts'] = np.int64(len(mp))
            d['mesh0_ntris'] = np.int64(len(mi) // 3)
            d['mesh0_o2w_translate'] = np.array([-3., 6., -15.])      # examples/head_loop.py:59
        else:
            d['mesh%d_p' % j] = mp
            d['mesh%d_idx' % j] = mi
        d['mesh%d_mat' % j] = np.int32(sc['tri_mat'][[k for k, pr in enumerate(bvh.primitives) if pr.shape.mesh is m][0]])
    # reference-traced rays
    rng = np.random.default_rng(2024)
    nr = 300
    o = np.array([-13., -1., -35.]) + rng.normal(0, .5, (nr, 3))
    tgt = np.array([-3., 6., -15.]) + rng.uniform(-6, 6, (nr, 3))
    rays = np.zeros((nr, 8))
    rays[:, 0:3] = o
    rays[:, 3:6] = tgt - o
    rays[:, 7] = np.where(rng.uniform(size=nr) < .7, np.inf, 1.0)
    prim_index = {id(p): i for i, p in enumerate(bvh.primitives)}
    res = np.zeros((nr, 4))
    for i in range(nr):
        r = geo.Ray(geo.Point(*rays[i, 0:3]), geo.Vector(*rays[i, 3:6]), rays[i, 6], rays[i, 7])
        isect = Intersection()
        hit = bvh.intersect(r, isect)
        r2 = geo.Ray(geo.Point(*rays[i, 0:3]), geo.Vector(*rays[i, 3:6]), rays[i, 6], rays[i, 7])
        res[i] = (int(hit), prim_index[id(isect.primitive)] if hit else -1, r.maxt, int(bvh.intersect_p(r2)))
    d['bvh_rays'] = rays
    d['bvh_res'] = res
    print('c2: %d nodes, %d tris, ray hits %d, %.1fs' % (d['n_nodes'], d['n_tris'], int(res[:, 0].sum()), time.time() - t),
          flush=True)
    # a window of the C2 render itself
    w = instrumented_render(model, 512, 512, 4, 4, DirectLightingIntegrator(), seed=3, sub=(106, 118, 158, 168))
    for k, v in w.items():
        if not k.startswith('log_hitgeo'):
            d[k] = v
    np.savez_compressed(os.path.join(OUT, 'c2_head_loop.npz'), **d)
    print('c2: window rendered, rays logged %d, sumL %.6f, total %.1fs' % (len(w['log_rays']), w['L'].sum(),
                                                                          time.time() - t))


def gen_specmini():
    """Specular chains (Mirror, Glass, Uber with Kr): DirectLightingIntegrator recursing through
    specular_reflect / specular_transmit (integrator/utility.py:120-200) and PathIntegrator sampling
    specular lobes."""
    model = C4Mini('spec')
    cam = model.configure_camera(x_res=24, y_res=24, file_name='/tmp/_fixture.png')
    sc = dump_scene(model.scene, cam)
    for name, integ, xs in (('direct', DirectLightingIntegrator(), 1), ('path', PathIntegrator(max_depth=5), 2)):
        t = time.time()
        d = instrumented_render(model, 24, 24, xs, xs, integ, seed=11)
        d.update(sc)
        np.savez_compressed(os.path.join(OUT, 'specmini_%s_24x24x%d.npz' % (name, xs * xs)), **d)
        kinds = d['log_rays'][:, 1].astype(int)
        print('specmini %s: %.1fs, closest %d any %d (samples %d), sumL %.6f' % (
            name, time.time() - t, (kinds == 0).sum(), (kinds == 1).sum(), len(d['L']), d['L'].sum()))


def gen_loop():
    """F9: LoopSubdiv.refine (shape/loopsubdiv.py:293-446) on small meshes — a closed octahedron
    (valence-4 extraordinary vertices), an open 4x3 grid (boundary rules) and a closed mesh with
    valence-3 and valence-5 vertices — at 1..3 levels; plus the base head mesh of
    examples/head_model.py (data), whose 3-level result is the mesh held by c2_head_loop.npz."""
    from pytracer.shape import create_loop_subdiv
    from pytracer.transform import Transform
    ident = Transform.translate(geo.Vector(0., 0., 0.))
    cases = {}
    cases['octa'] = (np.array([[1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1]], float) * 1.3,
                     [0, 2, 4, 2, 1, 4, 1, 3, 4, 3, 0, 4, 2, 0, 5, 1, 2, 5, 3, 1, 5])    # face (0,3,5) left out, below
    gx, gy = 4, 3
    rng = np.random.default_rng(3)
    P = np.array([[x, y, .3 * rng.normal()] for y in range(gy) for x in range(gx)], float)
    idx = []
    for y in range(gy - 1):
        for x in range(gx - 1):
            a = y * gx + x
            idx += [a, a + 1, a + gx, a + 1, a + gx + 1, a + gx]
    cases['grid'] = (P, idx)
    # tetrahedron with one face split into three: valences 3, 3, 4, 4... and a 5-valent apex via a pentagonal bipyramid
    n = 5
    ring = [[np.cos(2 * np.pi * k / n), np.sin(2 * np.pi * k / n), 0.1 * k] for k in range(n)]
    P = np.array(ring + [[0, 0, 1.2], [0, 0, -.9]], float)
    idx = []
    for k in range(n):
        idx += [k, (k + 1) % n, n, (k + 1) % n, k, n + 1]
    cases['bipyramid'] = (P, idx[3:])          # first face left out
    # NB: every mesh is OPEN at vertex 0.  The tangent pass of refine() does `S += ...` for interior
    # vertices before S exists (loopsubdiv.py:409): a mesh whose first vertex is interior raises
    # UnboundLocalError in the reference, so closed meshes cannot be subdivided by it at all.
    d = {}
    for name, (P, idx) in cases.items():
        d[name + '_P'] = np.asarray(P, np.float64)
        d[name + '_idx'] = np.asarray(idx, np.int32)
        for lv in (1, 2, 3):
            sd = create_loop_subdiv(ident, ident.inverse(), False,
                                    {'indices': list(idx), 'P': list(np.asarray(P, float).ravel()), 'nlevels': lv})
            m = sd.refine()[0]
            d['%s_l%d_P' % (name, lv)] = np.array([np.array(q) for q in m.p], np.float64)
            d['%s_l%d_idx' % (name, lv)] = np.array(m.vertexIndex, np.int32)
            print('loop %s level %d: %d verts %d tris' % (name, lv, len(m.p), m.ntris))
    from examples.head_model import model as head_model
    d['head_P'] = np.array(head_model['P'], np.float64).reshape(-1, 3)
    d['head_idx'] = np.array(head_model['indices'], np.int32)
    np.savez_compressed(os.path.join(OUT, 'loop_subdiv.npz'), **d)
    print('loop: head base %d verts %d tris' % (len(d['head_P']), len(d['head_idx']) // 3))


def gen_xtra():
    """§8(f) rank 2: WhittedIntegrator (whitted.py:16-59), MetalMaterial (FresnelConductor),
    OrthoCamera / EnvironmentCamera and the thin-lens branches of the projective cameras."""
    runs = [('whitted', 'xtra', 'perspective', WhittedIntegrator(), 24, 24),
            ('metal_direct', 'xtra', 'perspective', DirectLightingIntegrator(), 20, 20),
            ('metal_path', 'xtra', 'perspective', PathIntegrator(max_depth=5), 16, 16),
            ('lens_direct', 'lights', 'perspective_lens', DirectLightingIntegrator(), 16, 16),
            ('ortho_direct', 'lights', 'ortho', DirectLightingIntegrator(), 16, 16),
            ('ortholens_whitted', 'xtra', 'ortho_lens', WhittedIntegrator(), 16, 16)]
    # EnvironmentCamera is not among them: with this scene the reference raises ValueError for every seed
    # tried (Camera.generate_ray_differential's shifted rays make DifferentialGeometry.compute_differential
    # produce NaN, which geometry.Vector refuses) — ('env_direct', 'lights', 'environment', Direct, 24, 12)
    for name, variant, camk, integ, xr, yr in runs:
        model = C4Mini(variant)
        model.camera_kind = camk
        t = time.time()
        cam = model.configure_camera(x_res=xr, y_res=yr, file_name='/tmp/_fixture.png')
        sc = dump_scene(model.scene, cam)
        d = instrumented_render(model, xr, yr, 1, 1, integ, seed=21)
        d.update(sc)
        np.savez_compressed(os.path.join(OUT, 'xtra_%s.npz' % name), **d)
        kinds = d['log_rays'][:, 1].astype(int)
        print('xtra %s: %.1fs, closest %d any %d (samples %d), sumL %.6f, zero-L %d' % (
            name, time.time() - t, (kinds == 0).sum(), (kinds == 1).sum(), len(d['L']), d['L'].sum(),
            int((d['L'].sum(1) == 0).sum())), flush=True)


ALL = dict(xtra=gen_xtra, loop=gen_loop, specmini=gen_specmini, c2=gen_c2, c4mini=gen_c4mini, scene=gen_scene, direct=gen_direct, path=gen_path, c1=gen_c1_window, bvh=gen_bvh)

def gen_converged():
    """The reference, untouched (its own StratifiedSampler, np.random streams and SamplerRenderer.render), on a
    24x24-pixel window of the 128x128 head mesh at 8x8 = 64 spp: the image the production (in-kernel) sample
    streams must converge to.  CONV_KIND=direct|path picks the integrator (run both, in parallel if you like)."""
    import contextlib
    import io
    import examples.head_mesh as model
    kind = os.environ.get('CONV_KIND', 'direct')
    sub = (52, 76, 44, 68)
    camera = model.configure_camera(x_res=128, y_res=128, file_name='/tmp/_fixture_conv_%s.png' % kind)
    sampler = StratifiedSampler(sub[0], sub[1], sub[2], sub[3], 8, 8, True, 0., 0.)
    integ = DirectLightingIntegrator() if kind == 'direct' else PathIntegrator(max_depth=5)
    renderer = SamplerRenderer(sampler, camera, integ, None)
    np.random.seed(1234 if kind == 'direct' else 4321)
    t = time.time()
    with contextlib.redirect_stdout(io.StringIO()):
        renderer.render(model.scene)
    dt = time.time() - t
    film = camera.film
    W, H = film.xPixel_cnt, film.yPixel_cnt
    Lxyz = np.zeros((H, W, 3))
    wsum = np.zeros((H, W))
    for x in range(W):
        for y in range(H):
            Lxyz[y, x] = film.pixels[x][y].Lxyz
            wsum[y, x] = film.pixels[x][y].weight_sum
    np.savez_compressed(os.path.join(OUT, 'converged_%s_24x24x64.npz' % kind), film_Lxyz=Lxyz, film_wsum=wsum,
                        sub=np.array(sub, np.int32), spp=np.array([8, 8], np.int32), seconds=np.float64(dt))
    print('converged %s: %.1fs, sum Lxyz %.6f' % (kind, dt, Lxyz.sum()))


ALL['converged'] = gen_converged


def gen_axis():
    """Axis-parallel rays through the soup2000 tree whose origins lie EXACTLY on node box planes: 1/d = +-inf and
    (plane - origin) = 0 give NaN slab distances, and a -0.0 direction component gives inv = -inf with
    dir_is_neg false — the cases where the outcome depends on the literal comparison order of
    BVH._intersect_p (bvh.py:295-320)."""
    from pytracer.shape import create_triangle_mesh
    from pytracer.transform import Transform
    from pytracer.material import MatteMaterial
    from pytracer.texture import ConstantTexture
    from pytracer.scene import Scene
    P, idx = soup(2000, 12, 0)
    tr = Transform.translate(geo.Vector(0., 0., 0.))
    mesh = create_triangle_mesh(tr, tr.inverse(), False, {'indices': list(idx), 'P': list(P.ravel())})
    mat = MatteMaterial(ConstantTexture(Spectrum([.5, .5, .5])), ConstantTexture(0.))
    bvh = BVH([GeometricPrimitive(mesh, mat)])
    d = dump_scene(Scene(bvh, [], None))
    nb = d['node_bounds']
    inner = [i for i in range(96) if d['node_meta'][i, 1] == 0]
    rng = np.random.default_rng(77)
    rows = []
    for k in range(720):
        # an interior node near the root: certainly visited, and its box corners are made of coordinates of
        # different triangles (a leaf's corner can be a vertex: a ray through it is a last-ulp coin toss between
        # BLAS ddot and any restatement, which is not what this fixture is about)
        node = nb[inner[rng.integers(0, len(inner))]]
        axis = k % 3                                       # the ray runs along this axis
        sign = 1. if (k // 3) % 2 == 0 else -1.
        o = rng.uniform(node[0:3], node[3:6])              # inside the node's box ...
        o[axis] = node[axis] - 3. if sign > 0 else node[3 + axis] + 3.     # ... seen from outside along the axis
        dvec = np.zeros(3)
        dvec[axis] = sign
        mode = (k // 6) % 6
        a1, a2 = (axis + 1) % 3, (axis + 2) % 3
        if mode in (1, 3, 5):
            o[a1] = node[a1 + (3 if rng.uniform() < .5 else 0)]           # on a plane of the box: 0 * inf
        if mode in (2, 3):
            o[a2] = node[a2 + (3 if rng.uniform() < .5 else 0)]
        if mode in (4, 5):
            dvec[a1] = -0.0                                                # inv = -inf, dir_is_neg = False
        if mode == 5:
            dvec[a2] = -0.0
        rows.append(np.concatenate([o, dvec, [0., np.inf if k % 5 else 25.]]))
    rays = np.array(rows)
    prim_index = {id(p): i for i, p in enumerate(bvh.primitives)}
    res = np.zeros((len(rays), 4))
    for i in range(len(rays)):
        r = geo.Ray(geo.Point(*rays[i, 0:3]), geo.Vector(*rays[i, 3:6]), rays[i, 6], rays[i, 7])
        isect = Intersection()
        hit = bvh.intersect(r, isect)
        r2 = geo.Ray(geo.Point(*rays[i, 0:3]), geo.Vector(*rays[i, 3:6]), rays[i, 6], rays[i, 7])
        anyh = bvh.intersect_p(r2)
        res[i] = (int(hit), prim_index[id(isect.primitive)] if hit else -1, r.maxt, int(anyh))
    d['rays'] = rays
    d['res'] = res
    np.savez_compressed(os.path.join(OUT, 'bvh_axis_rays.npz'), **d)
    print('axis rays: nodes %d hits %d any %d' % (len(bvh.nodes), int(res[:, 0].sum()), int(res[:, 3].sum())))


ALL['axis'] = gen_axis


def gen_tex():
    """§8(f) rank 3 (textures): C4Mini('tex') rendered by the reference with Direct (1 spp) and Path (4 spp)."""
    model = C4Mini('tex')
    cam = model.configure_camera(x_res=24, y_res=24, file_name='/tmp/_fixture.png')
    sc = dump_scene(model.scene, cam)
    for name, integ, xs in (('direct', DirectLightingIntegrator(), 1), ('path', PathIntegrator(max_depth=5), 2)):
        t = time.time()
        d = instrumented_render(model, 24, 24, xs, xs, integ, seed=31)
        d.update(sc)
        np.savez_compressed(os.path.join(OUT, 'texmini_%s_24x24x%d.npz' % (name, xs * xs)), **d)
        print('texmini %s: %.1fs, rays logged %d, sumL %.6f, zero-L %d of %d' % (
            name, time.time() - t, len(d['log_rays']), d['L'].sum(), int((d['L'].sum(1) == 0).sum()), len(d['L'])), flush=True)


ALL['tex'] = gen_tex


def gen_tex2():
    """Textured Mirror / Glass / Uber colours (per-hit lobe existence) under Direct, Path and Whitted."""
    model = C4Mini('tex2')
    cam = model.configure_camera(x_res=24, y_res=24, file_name='/tmp/_fixture.png')
    sc = dump_scene(model.scene, cam)
    for name, integ, xs in (('direct', DirectLightingIntegrator(), 1), ('path', PathIntegrator(max_depth=5), 2),
                            ('whitted', WhittedIntegrator(), 1)):
        t = time.time()
        d = instrumented_render(model, 24, 24, xs, xs, integ, seed=41)
        d.update(sc)
        np.savez_compressed(os.path.join(OUT, 'tex2mini_%s_24x24x%d.npz' % (name, xs * xs)), **d)
        print('tex2mini %s: %.1fs, rays logged %d, sumL %.6f, zero-L %d of %d' % (
            name, time.time() - t, len(d['log_rays']), d['L'].sum(), int((d['L'].sum(1) == 0).sum()), len(d['L'])), flush=True)


ALL['tex2'] = gen_tex2


def gen_wrink():
    """WrinkledTexture (texture/texture/__init__.py:252-266) as Matte sigma and Plastic / Uber roughness under
    Direct, Path and Whitted, plus the reference's own values of the three textures at 400 points."""
    model = C4Mini('wrink')
    cam = model.configure_camera(x_res=24, y_res=24, file_name='/tmp/_fixture.png')
    sc = dump_scene(model.scene, cam)
    rng = np.random.default_rng(77)
    pts = np.concatenate([rng.uniform(-12., 12., (300, 3)), rng.uniform(-300., 300., (60, 3)),
                          np.round(rng.uniform(-5., 5., (40, 3)))])        # lattice points: noise() = 0 there
    from pytracer.geometry.diffgeom import DifferentialGeometry

    class _DG(object):
        pass
    vals = np.zeros((len(model.wrinkled), len(pts)))
    for k, tex in enumerate(model.wrinkled):
        for i, p in enumerate(pts):
            dg = _DG()
            dg.p, dg.dpdx, dg.dpdy = geo.Point(*p), geo.Vector(0., 0., 0.), geo.Vector(0., 0., 0.)
            vals[k, i] = float(tex(dg))
    sc['wr_points'], sc['wr_values'] = pts, vals
    sc['wr_records_m'] = np.array([np.array(t.mapping.w2t.m, np.float64).ravel() for t in model.wrinkled])
    sc['wr_records_oo'] = np.array([[float(t.omega), float(t.octaves)] for t in model.wrinkled])
    for name, integ, xs in (('direct', DirectLightingIntegrator(), 1), ('path', PathIntegrator(max_depth=5), 2),
                            ('whitted', WhittedIntegrator(), 1)):
        t = time.time()
        d = instrumented_render(model, 24, 24, xs, xs, integ, seed=61)
        d.update(sc)
        np.savez_compressed(os.path.join(OUT, 'wrinkmini_%s_24x24x%d.npz' % (name, xs * xs)), **d)
        print('wrinkmini %s: %.1fs, rays logged %d, sumL %.6f, zero-L %d of %d' % (
            name, time.time() - t, len(d['log_rays']), d['L'].sum(), int((d['L'].sum(1) == 0).sum()), len(d['L'])), flush=True)


ALL['wrink'] = gen_wrink


def gen_uv():
    """Meshes with uv coordinates (shape/triangle.py:125-131,209-231) under Direct and Path."""
    model = C4Mini('uv')
    cam = model.configure_camera(x_res=24, y_res=24, file_name='/tmp/_fixture.png')
    sc = dump_scene(model.scene, cam)
    for name, integ, xs in (('direct', DirectLightingIntegrator(), 1), ('path', PathIntegrator(max_depth=5), 2)):
        t = time.time()
        d = instrumented_render(model, 24, 24, xs, xs, integ, seed=51)
        d.update(sc)
        np.savez_compressed(os.path.join(OUT, 'uvmini_%s_24x24x%d.npz' % (name, xs * xs)), **d)
        print('uvmini %s: %.1fs, rays logged %d, sumL %.6f, zero-L %d of %d, degenerate uv triangles %d' % (
            name, time.time() - t, len(d['log_rays']), d['L'].sum(), int((d['L'].sum(1) == 0).sum()), len(d['L']),
            int((np.abs(sc['tri_uv'][:, 0:2] - sc['tri_uv'][:, 2:4]).sum(1) == 0).sum())), flush=True)


ALL['uv'] = gen_uv


def gen_meas():
    """§8(f) rank 3 (measured BRDF): C4Mini('meas') under Direct, Path and Whitted, plus a table of direct
    IrIsotropicBRDF.f / sample_f / pdf evaluations, directions outside the table's range included."""
    from pytracer.reflection import IrIsotropicBRDF
    model = C4Mini('meas')
    cam = model.configure_camera(x_res=24, y_res=24, file_name='/tmp/_fixture.png')
    sc = dump_scene(model.scene, cam)
    rng = np.random.default_rng(61)
    rows = []
    for mi in (0, 1):
        mt = [p.material for p in model.scene.aggregate.primitives if type(p.material).__name__ == 'MeasuredMaterial']
        mt = [m for k, m in enumerate(mt) if m not in mt[:k]][mi]
        b = IrIsotropicBRDF(mt.theta_phi_tree, mt.theta_phi_samples)
        for k in range(150):
            wo = rng.normal(size=3)
            wi = rng.normal(size=3)
            if k % 3:                           # mostly unit vectors in one hemisphere; the rest arbitrary (Q1)
                wo /= np.linalg.norm(wo); wi /= np.linalg.norm(wi)
                wo[2] = abs(wo[2]); wi[2] = abs(wi[2])
            u1, u2 = rng.uniform(size=2)
            f = b.f(geo.Vector(*wo), geo.Vector(*wi))
            pdf = b.pdf(geo.Vector(*wo), geo.Vector(*wi))
            spdf, swi, sf = b.sample_f(geo.Vector(*wo), u1, u2)
            rows.append([mi] + list(wo) + list(wi) + [u1, u2] + list(f) + [pdf, spdf] + list(swi) + list(sf))
    sc['brdf_eval'] = np.array(rows, np.float64)
    for name, integ, xs in (('direct', DirectLightingIntegrator(), 1), ('path', PathIntegrator(max_depth=5), 2),
                            ('whitted', WhittedIntegrator(), 1)):
        t = time.time()
        d = instrumented_render(model, 24, 24, xs, xs, integ, seed=61)
        d.update(sc)
        np.savez_compressed(os.path.join(OUT, 'measmini_%s_24x24x%d.npz' % (name, xs * xs)), **d)
        print('measmini %s: %.1fs, rays logged %d, sumL %.6f, zero-L %d of %d' % (
            name, time.time() - t, len(d['log_rays']), d['L'].sum(), int((d['L'].sum(1) == 0).sum()), len(d['L'])), flush=True)


ALL['meas'] = gen_meas


if __name__ == '__main__':
    which = sys.argv[1:] or list(ALL)
    for w in which:
        ALL[w]()
