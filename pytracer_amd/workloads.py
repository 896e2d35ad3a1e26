"""Synthetic workloads of BASELINE.json / SURVEY.md §8(d), described without the reference.

``c4_soup``: configs[3] — a soup of random triangles (BVH stress), alternating Matte / Plastic-like
materials, one SpotLight and one sphere DiffuseAreaLight, camera at (0,0,-30) looking +z, fov 60.
The tree is built by the native reference-exact builder (pt_bvh_build)."""
import numpy as np

from . import _abi as A
from . import accel, host
from .flatten import lobes_from_params


def c4_triangles(n_tris=10 ** 6, seed=12345, edge_sigma=.05):
    """centres U[-10,10]^3, two edge vectors N(0, 0.05^2 I) each."""
    rng = np.random.default_rng(seed)
    c = rng.uniform(-10., 10., (n_tris, 3))
    e1 = rng.normal(0., edge_sigma, (n_tris, 3))
    e2 = rng.normal(0., edge_sigma, (n_tris, 3))
    return np.concatenate([c, c + e1, c + e2], axis=1)


def _rotate_x(deg):
    """Transform.rotate_x, transform/__init__.py:132-141."""
    s, c = np.sin(np.deg2rad(deg)), np.cos(np.deg2rad(deg))
    m = np.eye(4)
    m[1, 1], m[1, 2], m[2, 1], m[2, 2] = c, -s, s, c
    return m


def c4_lights():
    """SpotLight at (0,15,-25) aimed at the origin, width 30 deg, falloff 25 deg, I = 200
    (light/light.py:162-171), and a sphere DiffuseAreaLight r = 3, emit 10, ns = 2 (light.py:484-500)."""
    l2w = host._translate(0., 15., -25.).dot(_rotate_x(np.rad2deg(np.arctan2(15., 25.))))
    spot = dict(kind=A.PT_LIGHT_SPOT, pos=l2w[:3, 3].copy(), intensity=np.full(3, 200.), ns=1,
                w2l=np.linalg.inv(l2w), cos_width=np.cos(np.deg2rad(30.)), cos_falloff=np.cos(np.deg2rad(25.)))
    o2w = host._translate(-10., 10., -20.)
    sphere = dict(kind=A.PT_LIGHT_DIFFUSE_SPHERE, emit=np.full(3, 10.), ns=2, radius=3., o2w=o2w,
                  w2o=np.linalg.inv(o2w), ro=False, zmin=-3., zmax=3., phimax=2. * np.pi)
    return [spot, sphere]


def c4_materials(n_materials=64, seed=12345):
    """alternating Matte(Kd U[.1,.9]) / Plastic(Kd, Ks = .3, Blinn exponent U[5,100])."""
    rng = np.random.default_rng(seed + 1)
    lobes = []
    for i in range(n_materials):
        kd = rng.uniform(.1, .9, 3)
        if i % 2 == 0:
            lobes.append(lobes_from_params('matte', Kd=kd, sigma=0.))
        else:
            lobes.append(lobes_from_params('plastic', Kd=kd, Ks=np.full(3, .3), roughness=1. / rng.uniform(5., 100.)))
    return lobes


def c4_soup(n_tris=10 ** 6, seed=12345, n_materials=64, edge_sigma=.05):
    """FlatScene of the C4 soup (built natively; ~1 s per million triangles)."""
    tv = c4_triangles(n_tris, seed, edge_sigma)
    tm = (np.arange(n_tris) % n_materials).astype(np.int32)
    flat, _ = accel.flat_scene_from_triangles(tv, c4_materials(n_materials, seed), c4_lights(), tri_mat=tm)
    return flat


def c4_camera_film(x_res=1920, y_res=1080):
    """PerspectiveCamera at (0,0,-30) looking +z, fov 60, screen window scaled to the aspect ratio;
    Lanczos(4,4,3) film like the examples."""
    ar = float(x_res) / float(y_res)
    cam = host.perspective_camera(host._translate(0., 0., -30.), [-ar, ar, -1., 1.], 60., x_res, y_res)
    return cam, host.image_film(x_res, y_res, 'lanczos', 4., 4., tau=3.)


def brdf_remap(wo, wi):
    """reflection/utility.py:76-90 for arrays of unit directions [n,3] (host side: synthetic tables only)."""
    def sph_phi(v):
        p = np.arctan2(v[:, 1], v[:, 0])
        return np.where(p > 0., p, p + 2. * np.pi)
    dphi = sph_phi(wi) - sph_phi(wo)
    dphi = np.where(dphi < 0., dphi + 2. * np.pi, dphi)
    dphi = np.where(dphi > 2. * np.pi, dphi - 2. * np.pi, dphi)
    dphi = np.where(dphi > np.pi, 2. * np.pi - dphi, dphi)
    st = lambda v: np.sqrt(np.maximum(0., 1. - v[:, 2] * v[:, 2]))
    return np.stack([st(wi) * st(wo), dphi / np.pi, wi[:, 2] * wo[:, 2]], axis=1)


def synthetic_brdf_table(n_samples=1000, seed=5):
    """(p[n,3], v[n,3]) of a synthetic irregularly sampled isotropic BRDF in the form a loaded MeasuredMaterial
    holds it (material.py:463-479): positions = brdf_remap of uniformly drawn upper-hemisphere direction pairs,
    values = RGB (a tinted diffuse base plus a lobe around the mirror direction).  The reference ships no .brdf file."""
    rng = np.random.default_rng(seed)

    def dirs(n):
        z = rng.uniform(0., 1., n)
        ph = rng.uniform(0., 2. * np.pi, n)
        r = np.sqrt(np.maximum(0., 1. - z * z))
        return np.stack([r * np.cos(ph), r * np.sin(ph), z], axis=1)
    wo, wi = dirs(n_samples), dirs(n_samples)
    wh = wo + wi
    wh /= np.linalg.norm(wh, axis=1, keepdims=True)
    gloss = .8 * wh[:, 2] ** 40
    base = np.array([.30, .22, .15]) * rng.uniform(.8, 1.2, (n_samples, 1))
    return brdf_remap(wo, wi), base / np.pi + gloss[:, None]


def c5_head_measured(head_npz, n_samples=1000, seed=5):
    """configs[4] as far as the reference can run it: the head mesh of examples/head_mesh.py with a
    MeasuredMaterial (irregular isotropic table, `n_samples` RGB samples) in place of its Uber material and a
    closed-form checkerboard (UV mapping) as the backdrop's Kd."""
    from .flatten import flat_scene_from_npz
    d = dict(np.load(head_npz))
    head = [i for i in range(int(d['n_materials'])) if int(d['mat%d_kind' % i]) == 1][0]
    back = [i for i in range(int(d['n_materials'])) if int(d['mat%d_kind' % i]) == 0][0]
    for k in [k for k in d if k.startswith('mat%d_' % head)]:
        del d[k]
    p, v = synthetic_brdf_table(n_samples, seed)
    d['mat%d_kind' % head] = np.int32(6)
    d['mat%d_brdf_p' % head], d['mat%d_brdf_v' % head] = p, v
    del d['mat%d_Kd' % back]
    m = np.zeros(16)
    m[0:4] = [8., 8., 0., 0.]
    tv = np.zeros((4, 3))
    tv[0], tv[1] = [.4, .4, .4], [.05, .05, .05]
    d['mat%d_Kd_tex_kind' % back] = np.int32(A.PT_TEX_CHECKER2D)
    d['mat%d_Kd_tex_mapping' % back] = np.int32(A.PT_MAP_UV)
    d['mat%d_Kd_tex_m' % back], d['mat%d_Kd_tex_v' % back] = m, tv
    return flat_scene_from_npz(d)


def c2_meshes(c2_npz, loop_npz):
    """World-space meshes [(P, idx, material)] of examples/head_loop.py: the head = 3 levels of Loop
    subdivision of the control mesh (examples/head_model.py data, held by loop_subdiv.npz), translated
    like create_triangle_mesh does (head_loop.py:59-60); the backdrop from the C2 fixture."""
    d, base = np.load(c2_npz), np.load(loop_npz)
    P, idx = accel.loop_subdivide(base['head_P'], base['head_idx'], 3)
    return [(P + d['mesh0_o2w_translate'], idx, int(d['mesh0_mat'])),
            (d['mesh1_p'], d['mesh1_idx'], int(d['mesh1_mat']))], dict(d)


def c2_head_loop(c2_npz, loop_npz):
    """FlatScene of configs[1] (examples/head_loop.py), built natively end to end: Loop subdivision,
    then the reference-exact BVH (both bit-identical to the reference's; tests/test_c2_head_loop.py).
    Primitive.full_refine hands over each mesh's triangles in reverse (aggregate/primitive.py:68-75)."""
    from .flatten import FlatScene, _MAT_KINDS
    meshes, d = c2_meshes(c2_npz, loop_npz)
    tris = [P[idx].reshape(-1, 9)[::-1] for P, idx, _ in meshes]
    mats = [np.full(len(t), m, np.int32) for t, (_, _, m) in zip(tris, meshes)]
    tv, tm = np.concatenate(tris), np.concatenate(mats)
    nb, nm, order, _ = accel.build_bvh(accel.triangle_bounds(tv))
    lobes = []
    for i in range(int(d['n_materials'])):
        pre = 'mat%d_' % i
        params = {k[len(pre):]: d[k] for k in d if k.startswith(pre) and k != pre + 'kind'}
        lobes.append(lobes_from_params(_MAT_KINDS[int(d[pre + 'kind'])], **params))
    lights = []
    for i in range(int(d['n_lights'])):
        pre = 'light%d_' % i
        lights.append({k[len(pre):]: (d[k].item() if d[k].ndim == 0 else d[k]) for k in d if k.startswith(pre)})
    return FlatScene(nb, nm, tv[order], tm[order], np.zeros(len(tv), np.int32), lobes, lights)
