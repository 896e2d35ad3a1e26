"""pt_intersect_static — the one entry point that allocates nothing (include/pytracer_hip.h) — against
pt_intersect through the C ABI: same primitive ids, t, b1, b2 (bitwise) and the same census, on two calls in a
row, the second with a smaller input (the file-scope scratch keeps the first call's sums and must be cleared).
Collected last (test_zz_*): it must pass alone in a fresh process and after all the other tests."""
import ctypes as C
import os

import numpy as np
import pytest

import pytracer_amd as pa
from pytracer_amd import _abi as A

pytestmark = pytest.mark.gpu


def camera_like_rays(n, seed):
    """rays from the head-mesh camera position towards the head's bounding box, plus a few that miss"""
    rng = np.random.default_rng(seed)
    r = np.empty((8, n))
    r[0], r[1], r[2] = 13., 1., 35.
    r[0:3] += rng.normal(0, .5, (3, n))
    tgt = rng.uniform([-8, -10, -6], [8, 10, 6], (n, 3)).T
    d = tgt - r[0:3]
    r[3:6] = d / np.linalg.norm(d, axis=0)
    r[6] = 0.
    r[7] = np.inf
    return np.ascontiguousarray(r)


def test_static_entry_matches_pt_intersect(golden_dir):
    import torch
    from pytracer_amd._lib import Context, lib
    assert torch.cuda.is_available()
    d = dict(np.load(os.path.join(golden_dir, 'head_mesh_scene.npz')))
    flat = pa.flat_scene_from_npz(d)
    ctx = Context(0)
    ctx.upload(flat)
    assert hasattr(lib(), 'pt_intersect_static')
    for any_hit in (False, True):
        for n, seed in ((50000, 1), (777, 2)):                 # two calls in a row, the second smaller
            rays = torch.from_numpy(camera_like_rays(n, seed)).cuda()
            s_ref, s_new = A.pt_stats(), A.pt_stats()
            ref = ctx.intersect(rays, any_hit=any_hit, stats=s_ref)
            out = (torch.full((n,), -7, dtype=torch.int32, device='cuda'),
                   None if any_hit else torch.full((n,), -7., dtype=torch.float64, device='cuda'),
                   None if any_hit else torch.full((n,), -7., dtype=torch.float64, device='cuda'),
                   None if any_hit else torch.full((n,), -7., dtype=torch.float64, device='cuda'))
            new = ctx.intersect(rays, any_hit=any_hit, stats=s_new, static=True, out=out)
            torch.cuda.synchronize()
            assert torch.equal(ref[0], new[0])
            if not any_hit:
                assert int((ref[0] >= 0).sum()) > n // 4          # the batch does hit the head
                for a, b in zip(ref[1:], new[1:]):
                    assert torch.equal(a, b)                       # bitwise, inf for misses included
            da, db = s_ref.as_dict(), s_new.as_dict()
            for k in ('rays_closest', 'rays_any', 'node_visits', 'tri_tests', 'node_visits_any', 'tri_tests_any'):
                assert da[k] == db[k], (k, da[k], db[k])
            assert (da['rays_any'] if any_hit else da['rays_closest']) == n
    # stats == NULL: asynchronous, results still equal
    rays = torch.from_numpy(camera_like_rays(4096, 3)).cuda()
    ref = ctx.intersect(rays)
    out = (torch.empty(4096, dtype=torch.int32, device='cuda'),) + tuple(
        torch.empty(4096, dtype=torch.float64, device='cuda') for _ in range(3))
    new = ctx.intersect(rays, static=True, out=out)
    torch.cuda.synchronize()
    for a, b in zip(ref, new):
        assert torch.equal(a, b)
    # sizes beyond the maximum are refused before anything is launched
    from pytracer_amd._lib import PtError
    rc = lib().pt_intersect_static(ctx._h, rays.data_ptr(), C.c_int64(A.PT_INTERSECT_STATIC_MAX_RAYS + 1), 0,
                                   out[0].data_ptr(), None, None, None, None, None)
    assert rc == A.PT_ERR_CAPACITY
    assert b'maximum' in lib().pt_last_error(ctx._h)
    ctx.close()


def test_static_calls_of_two_contexts_may_overlap(golden_dir):
    """The overlap rule of include/pytracer_hip.h: calls on ONE context are serialised by the caller (its traversal
    stacks are per context), calls on DIFFERENT contexts of one GPU may be in flight together and only race on the
    census.  Two contexts forced onto the children-pair walk (PT_TRACE_PAIR=3: todo stacks partly in each context's
    own spill area), each with its own stream, large batches issued back to back without synchronising: every hit
    record equals the serial pt_intersect result bit for bit."""
    import torch
    from pytracer_amd._lib import Context
    d = dict(np.load(os.path.join(golden_dir, 'head_mesh_scene.npz')))
    flat = pa.flat_scene_from_npz(d)
    os.environ.update({'PT_TRACE_PAIR': '3', 'PT_PAIR_LDS_LEVELS': '2'})     # almost every todo level in the spill area
    ctxs = []
    try:
        for _ in range(2):
            c = Context(0)
            c.upload(flat)
            ctxs.append(c)
        n = 400000
        rays = [torch.from_numpy(camera_like_rays(n, 10 + k)).cuda() for k in range(2)]
        ref = [ctxs[k].intersect(rays[k]) for k in range(2)]
        torch.cuda.synchronize()
        streams = [torch.cuda.Stream() for _ in range(2)]
        outs = [(torch.full((n,), -7, dtype=torch.int32, device='cuda'),) +
                tuple(torch.full((n,), -7., dtype=torch.float64, device='cuda') for _ in range(3)) for _ in range(2)]
        torch.cuda.synchronize()
        for rep in range(3):                        # 6 launches queued on two streams before anything is waited for
            for k in range(2):
                ctxs[k].intersect(rays[k], static=True, out=outs[k], stream=streams[k].cuda_stream)
        torch.cuda.synchronize()
        for k in range(2):
            assert int((ref[k][0] >= 0).sum()) > n // 4
            for a, b in zip(ref[k], outs[k]):
                assert torch.equal(a, b)
    finally:
        for k in ('PT_TRACE_PAIR', 'PT_PAIR_LDS_LEVELS'):
            del os.environ[k]
        for c in ctxs:
            c.close()
