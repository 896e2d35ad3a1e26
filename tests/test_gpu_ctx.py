"""Context behaviour through the C ABI: a rejected pt_scene_upload leaves the previous scene exactly as it was
(light_ns used to be clobbered by the rejected scene's lights, after which a Direct render of the old
scene pushed more shadow rays per hit than its shadow queue was sized for), and no entry point changes the
caller's current device."""
import copy
import os

import numpy as np
import pytest

import pytracer_amd as pa
from pytracer_amd import _abi as A

pytestmark = pytest.mark.gpu


def test_rejected_upload_keeps_previous_scene(golden_dir):
    import torch
    from pytracer_amd._lib import Context, PtError
    d = dict(np.load(os.path.join(golden_dir, 'head_mesh_scene.npz')))
    flat_a = pa.flat_scene_from_npz(d)                      # one sphere light, ns = 2
    assert flat_a.light_ns() == [2]
    recs = copy.deepcopy(flat_a.light_records)
    recs[0]['ns'] = 16                                      # accepted: would size 16 shadow rays per hit
    bad = copy.deepcopy(recs[0])
    bad['ns'] = 17                                          # rejected (PT_ERR_CAPACITY), after light 0 was looked at
    flat_b = pa.FlatScene(flat_a.node_bounds, flat_a.node_meta, flat_a.tri_verts, flat_a.tri_mat, flat_a.tri_flip,
                          flat_a.material_lobes, recs + [bad])
    ctx = Context(0)
    cam, film = pa.camera_film_from_npz(d)
    smp = pa.sampler_desc(40, 88, 40, 88, 2, 2, True, 3)
    tile = pa.tile_desc(40, 88, 40, 88, 0, 4)
    integ = pa.integrator_desc(A.PT_INTEG_DIRECT, 5, False)

    def render():
        acc = torch.zeros((film.y_cnt, film.x_cnt, 4), dtype=torch.float64, device='cuda')
        st = A.pt_stats()
        ctx.render(cam, film, smp, integ, tile, acc, stats=st)
        torch.cuda.synchronize()
        return acc.cpu().numpy(), st.as_dict()

    ctx.upload(flat_a)
    a1, s1 = render()
    with pytest.raises(PtError) as e:
        ctx.upload(flat_b)
    assert e.value.code == A.PT_ERR_CAPACITY
    a2, s2 = render()                                       # scene A is still there, sized as before
    assert np.abs(a2 - a1).max() <= 1e-12 * np.abs(a1).max()
    for k in ('camera_samples', 'rays_closest', 'rays_any', 'node_visits', 'tri_tests'):
        assert s1[k] == s2[k], k
    assert s1['rays_any'] <= 2 * s1['rays_closest']         # ns = 2 shadow rays per hit, not 16
    ctx.close()


def test_entry_points_keep_the_callers_device():
    import torch
    from pytracer_amd._lib import Context
    before = torch.cuda.current_device()
    ctx = Context(0)
    assert torch.cuda.current_device() == before
    ctx.close()
    assert torch.cuda.current_device() == before


def test_two_contexts_on_one_device_render_concurrently(golden_dir):
    """One process may hold several contexts (one per GPU of a node, or several on one GPU): two contexts on cuda:0
    render the two interleaved-row shards of one image on two torch streams at the same time; the sum equals the
    one-context render (per-device launch attributes, workspaces and the second render stream are per context)."""
    import torch
    from pytracer_amd import host
    from pytracer_amd._lib import Context
    from pytracer_amd.renderer import shard_tile
    d = dict(np.load(os.path.join(golden_dir, 'head_mesh_scene.npz')))
    flat = pa.flat_scene_from_npz(d)
    res, xs = 256, 4
    cam, film = host.head_mesh_camera_film(res, res)
    smp = pa.sampler_desc(0, res, 0, res, xs, xs, True, 11)
    integ = pa.integrator_desc(A.PT_INTEG_PATH, 5, False)
    ctxs = [Context(0), Context(0)]
    for c in ctxs:
        c.upload(flat)
    whole = torch.zeros((film.y_cnt, film.x_cnt, 4), dtype=torch.float64, device='cuda')
    s_whole = A.pt_stats()
    ctxs[0].render(cam, film, smp, integ, shard_tile(smp, 0, 1), whole, stats=s_whole)
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    accs = [torch.zeros_like(whole) for _ in range(2)]
    torch.cuda.synchronize()
    for r in range(2):                                      # stats=None: nothing synchronises, both renders are in flight
        ctxs[r].render(cam, film, smp, integ, shard_tile(smp, r, 2, 'interleaved'), accs[r],
                       stream=streams[r].cuda_stream)
    torch.cuda.synchronize()
    both = accs[0] + accs[1]
    assert float((both - whole).abs().max()) <= 1e-12 * float(whole.abs().max())
    assert float(accs[0].abs().sum()) > 0 and float(accs[1].abs().sum()) > 0
    for c in ctxs:
        c.close()


def test_one_process_drives_eight_contexts(golden_dir):
    """The other multi-GPU shape the ABI allows (include/pytracer_hip.h: "one Python process drives 8 ctxs"): ONE process,
    one context per rank, every context rendering its interleaved-row shard of the same job into its OWN full-frame film
    on its own stream, all in flight together, the films summed at the end (what `dist.reduce` does across processes).
    Here the eight contexts share the one GPU of the box: the sum equals the one-context film (1e-12: the order of the f64
    atomic adds is free), the summed census equals the one-context census, and no context disturbs another's state."""
    import torch
    from pytracer_amd import host
    from pytracer_amd._lib import Context
    from pytracer_amd.renderer import shard_tile
    flat = pa.flat_scene_from_npz(dict(np.load(os.path.join(golden_dir, 'head_mesh_scene.npz'))))
    R, K = 256, 8
    cam, film = host.head_mesh_camera_film(R, R)
    smp = pa.sampler_desc(0, R, 0, R, 4, 4, True, 9)
    integ = pa.integrator_desc(A.PT_INTEG_PATH, 5, False)
    one = Context(0)
    one.upload(flat)
    whole = torch.zeros((film.y_cnt, film.x_cnt, 4), dtype=torch.float64, device='cuda')
    s1 = A.pt_stats()
    one.render(cam, film, smp, integ, pa.tile_desc(0, R, 0, R, 0, 16), whole, stats=s1)
    torch.cuda.synchronize()
    one.close()
    ctxs = [Context(0) for _ in range(K)]
    try:
        for c in ctxs:
            c.upload(flat)
        streams = [torch.cuda.Stream() for _ in range(K)]
        films = [torch.zeros_like(whole) for _ in range(K)]
        torch.cuda.synchronize()
        for rep in range(2):                       # the second round re-uses every context's workspace while others run
            for f in films:
                f.zero_()
            torch.cuda.synchronize()
            for r in range(K):                     # stats=None: nothing synchronises, all eight renders are queued at once
                ctxs[r].render(cam, film, smp, integ, shard_tile(smp, r, K, 'interleaved'), films[r],
                               stream=streams[r].cuda_stream)
            torch.cuda.synchronize()
            total = torch.stack(films).sum(0)
            assert float((total - whole).abs().max()) <= 1e-12 * float(whole.abs().max())
        # the census, shard by shard (a synchronising call each), adds up to the one-context census
        sk = A.pt_stats()
        for r in range(K):
            films[r].zero_()
            ctxs[r].render(cam, film, smp, integ, shard_tile(smp, r, K, 'interleaved'), films[r], stats=sk,
                           stream=streams[r].cuda_stream)
        torch.cuda.synchronize()
        assert sk.as_dict() == s1.as_dict()
    finally:
        for c in ctxs:
            c.close()


def test_render_when_device_memory_is_short(golden_dir):
    """pt_render takes at most half of the free device memory for its wavefront batch.  With all but ~6 GB of the GPU
    taken by the caller, a 16.8 M-path render runs in several batches; rendered twice in that state (the second call
    keeps the first call's workspace: what is free moves by the allocation granule, and a batch re-fitted to it would
    free and allocate the whole workspace per call) it equals the one-batch render, census included."""
    import torch
    from pytracer_amd import host
    from pytracer_amd._lib import Context
    flat = pa.flat_scene_from_npz(dict(np.load(os.path.join(golden_dir, 'head_mesh_scene.npz'))))
    cam, film = host.head_mesh_camera_film(1024, 1024)
    smp = pa.sampler_desc(0, 1024, 0, 1024, 4, 4, True, 0)
    tile = pa.tile_desc(0, 1024, 0, 1024, 0, 16)
    integ = pa.integrator_desc(A.PT_INTEG_PATH, 5, False)

    def render(ctx):
        acc = torch.zeros((film.y_cnt, film.x_cnt, 4), dtype=torch.float64, device='cuda')
        st = A.pt_stats()
        ctx.render(cam, film, smp, integ, tile, acc, stats=st)
        torch.cuda.synchronize()
        n_batches = len({r['batch'] for r in ctx.launch_log()})
        return acc, st.as_dict(), n_batches

    ref_ctx = Context(0)
    ref_ctx.upload(flat)
    ref_ctx.set_profiling(1)
    ref, s_ref, nb = render(ref_ctx)
    assert nb == 1
    ref_ctx.close()
    torch.cuda.empty_cache()
    ctx = Context(0)
    ctx.upload(flat)
    ctx.set_profiling(1)
    free, _ = torch.cuda.mem_get_info()
    hog = torch.empty(free - (6 << 30), dtype=torch.uint8, device='cuda')
    try:
        for _ in range(2):
            acc, s, nb = render(ctx)
            assert nb >= 3
            assert s == s_ref
            assert float((acc - ref).abs().max()) <= 1e-11 * float(ref.abs().max())
    finally:
        del hog
        ctx.close()
        torch.cuda.empty_cache()
