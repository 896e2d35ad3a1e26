"""Host mirror of the reference's accelerator interface (aggregate/accelerator/bvh.py) on top of
the native builder ``pt_bvh_build`` (include/pytracer_hip.h, pytracer_amd/csrc/pt_build.cpp).

    from pytracer_amd.accel import BVH            # instead of pytracer.aggregate.BVH
    bvh = BVH([GeometricPrimitive(mesh, material), ...], max_prim_per_node=4, method='sah')
    scene = Scene(bvh, lights, None)               # unchanged reference Scene

``BVH`` has the constructor of the reference class (bvh.py:68) and leaves the same public state
behind — ``primitives`` reordered exactly like ``ordered_prims`` (bvh.py:103) and ``nodes`` with
``bounds / offset / n_prim / axis`` (bvh.py:60-66) — so reference code that walks it and
``flatten_scene`` both work, while the 259 s Python build of the 177 k-triangle head_loop scene
takes a fraction of a second.
"""
import ctypes as C

import numpy as np

from . import _abi as A

_METHODS = {'sah': A.PT_BVH_SAH, 'middle': A.PT_BVH_MIDDLE}


def triangle_bounds(tri_verts):
    """Triangle.world_bound (shape/triangle.py:140-144) of [n,9] world-space triangles → [n,6]."""
    v = np.ascontiguousarray(tri_verts, np.float64).reshape(-1, 3, 3)
    return np.concatenate([v.min(axis=1), v.max(axis=1)], axis=1)


def build_bvh(prim_bounds, max_prim_per_node=4, method='sah'):
    """Reference-exact BVH over primitive bounds [n,6] (pMin, pMax).

    Returns (node_bounds [N,6] f64, node_meta [N,3] i32 = offset/n_prim/axis, prim_order [n] i32,
    depth).  ``ordered_prims[k] = primitives[prim_order[k]]``.  Raises ValueError where the
    reference build itself fails (its asserts)."""
    from ._lib import lib
    pb = np.ascontiguousarray(prim_bounds, np.float64).reshape(-1, 6)
    n = len(pb)
    m = _METHODS.get(str(method).lower())
    if m is None:
        # bvh.py:79-83: 'equal' raises AttributeError, anything else falls back to SAH with a log line
        if str(method).lower() == 'equal':
            raise AttributeError('SplitMethod.EQUAL_COUNTS does not exist in the reference (bvh.py:28,80)')
        m = A.PT_BVH_SAH
    cap = max(2 * n - 1, 1)
    nb = np.empty((cap, 6), np.float64)
    nm = np.empty((cap, 3), np.int32)
    order = np.empty(max(n, 1), np.int32)
    n_nodes = C.c_int64(0)
    depth = C.c_int32(0)
    L = lib()
    rc = L.pt_bvh_build(pb.ctypes.data, n, int(max_prim_per_node), m, nb.ctypes.data, nm.ctypes.data,
                        order.ctypes.data, C.byref(n_nodes), C.byref(depth))
    if rc != 0:
        raise ValueError('pt_bvh_build: %s' % L.pt_build_last_error().decode('utf-8', 'replace'))
    k = n_nodes.value
    return nb[:k].copy(), nm[:k].copy(), order[:n].copy(), depth.value


def loop_subdivide(P, indices, n_levels=3):
    """LoopSubdiv.refine (shape/loopsubdiv.py:293-446) natively: control mesh (P [nv,3] object space,
    indices [nf*3]) → (limit-surface vertices [nv',3], indices [nf * 4^n_levels * 3]), bit-identical to
    the reference's TriangleMesh before its object-to-world transform.  Raises ValueError where the
    reference itself raises (e.g. closed meshes: UnboundLocalError in its tangent pass)."""
    from ._lib import lib
    P = np.ascontiguousarray(P, np.float64).reshape(-1, 3)
    vi = np.ascontiguousarray(indices, np.int32).ravel()
    if len(vi) % 3:
        raise ValueError('indices must hold whole triangles')
    nv, nf = len(P), len(vi) // 3
    nf_out = nf * 4 ** int(n_levels)
    nv_cap = nv + nf * (4 ** int(n_levels) - 1)
    Po = np.empty((max(nv_cap, 1), 3), np.float64)
    io = np.empty(max(3 * nf_out, 1), np.int32)
    a, b = C.c_int64(0), C.c_int64(0)
    L = lib()
    rc = L.pt_loop_subdiv(P.ctypes.data, nv, vi.ctypes.data, nf, int(n_levels), Po.ctypes.data, len(Po),
                          io.ctypes.data, len(io), C.byref(a), C.byref(b))
    if rc != 0:
        raise ValueError('pt_loop_subdiv: %s' % L.pt_subdiv_last_error().decode('utf-8', 'replace'))
    return Po[:a.value].copy(), io[:3 * b.value].copy()


def create_loop_subdiv(o2w, w2o, ro, params):
    """Drop-in for pytracer.shape.create_loop_subdiv (shape/loopsubdiv.py:19-44) in a scene script that
    has the reference importable: subdivides natively and returns the refined reference ``TriangleMesh``
    (what ``LoopSubdiv.refine()`` would hand to the aggregate, loopsubdiv.py:441-446) — same vertices,
    same indices, without the minutes of Python object juggling."""
    from pytracer.shape import create_triangle_mesh        # the reference's own mesh class
    nlevels = params.get('nlevels', 3)
    vi, P = params.get('indices'), params.get('P')
    if vi is None or P is None or len(P) % 3 or len(vi) % 3:
        return None                                        # loopsubdiv.py:25-34
    Po, io = loop_subdivide(np.asarray(P, np.float64).reshape(-1, 3), vi, nlevels)
    return create_triangle_mesh(o2w, w2o, ro, {'indices': [int(k) for k in io], 'P': [float(x) for x in Po.ravel()]})


def flat_scene_from_triangles(tri_verts, material_lobes, light_records=(), tri_mat=None, tri_flip=None,
                              max_prim_per_node=4, method='sah'):
    """FlatScene from a world-space triangle list in primitive order (what a reference script hands
    to ``BVH([...])``): native build + reorder.  Returns (FlatScene, prim_order)."""
    from .flatten import FlatScene
    tv = np.ascontiguousarray(tri_verts, np.float64).reshape(-1, 9)
    n = len(tv)
    tm = np.zeros(n, np.int32) if tri_mat is None else np.asarray(tri_mat, np.int32)
    tf = np.zeros(n, np.int32) if tri_flip is None else np.asarray(tri_flip, np.int32)
    nb, nm, order, _ = build_bvh(triangle_bounds(tv), max_prim_per_node, method)
    return FlatScene(nb, nm, tv[order], tm[order], tf[order], material_lobes, list(light_records)), order


class _Pt(np.ndarray):
    """A 3-vector with .x/.y/.z like geometry.Point (geometry/__init__.py:213-285)."""
    x = property(lambda s: float(s[0]))
    y = property(lambda s: float(s[1]))
    z = property(lambda s: float(s[2]))


class _Box(object):
    """geometry.BBox as far as BVH users read it: pMin, pMax, box[0], box[1]."""
    __slots__ = ('pMin', 'pMax')

    def __init__(self, row):
        self.pMin = np.array(row[0:3], np.float64).view(_Pt)
        self.pMax = np.array(row[3:6], np.float64).view(_Pt)

    def __getitem__(self, key):
        if key == 0:
            return self.pMin
        if key == 1:
            return self.pMax
        raise KeyError(key)


class _LinearNode(object):
    """BVH._LinearNode, bvh.py:60-66"""
    __slots__ = ('bounds', 'offset', 'n_prim', 'axis')

    def __init__(self, bounds, offset, n_prim, axis):
        self.bounds, self.offset, self.n_prim, self.axis = bounds, offset, n_prim, axis


class BVH(object):
    """Drop-in for pytracer.aggregate.BVH construction (bvh.py:68-107)."""

    def __init__(self, p, max_prim_per_node=4, method='sah'):
        self.primitives = []
        self.max_prim_per_node = max_prim_per_node
        for prim in p:
            prim.full_refine(self.primitives)          # bvh.py:72-73
        self.split_method = str(method).lower()
        if len(self.primitives) == 0:
            self.nodes = None                          # bvh.py:85-87
            self.flat_nodes = (np.empty((0, 6)), np.empty((0, 3), np.int32))
            return
        pb = np.empty((len(self.primitives), 6), np.float64)
        for i, prim in enumerate(self.primitives):     # bvh.py:91
            b = prim.world_bound()
            pb[i, 0:3] = b.pMin
            pb[i, 3:6] = b.pMax
        nb, nm, order, self.depth = build_bvh(pb, max_prim_per_node, method)
        self.primitives = [self.primitives[k] for k in order]      # bvh.py:103
        self.flat_nodes = (nb, nm)                     # what flatten_scene uploads, no per-node objects
        self.nodes = [_LinearNode(_Box(nb[i]), int(nm[i, 0]), int(nm[i, 1]), int(nm[i, 2])) for i in range(len(nb))]

    # Aggregate protocol (aggregate/aggregate.py, aggregate/primitive.py:33-60)
    def world_bound(self):
        return self.nodes[0].bounds if self.nodes else None

    def can_intersect(self):
        return True

    def intersect(self, ray, isect):
        raise NotImplementedError('per-ray intersect() is a host-loop API; render through GpuSamplerRenderer or '
                                  'trace ray batches with Context.intersect')

    intersect_p = intersect
