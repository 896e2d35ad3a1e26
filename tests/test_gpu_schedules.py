"""The ways work reaches the waves — chunk cursors, one walk per wave at level 0, lane refill of the pair walk, hit
classes, two streams, wavefront batches — are scheduling only: on every fixture scene and integrator the production
render with the defaults gives the census and per-sample radiance of the render with all of them switched off (static
strides, per-lane walks, one stream, one batch), on windows whose sizes are not multiples of anything."""
import os
import zlib

import numpy as np
import pytest

import pytracer_amd as pa
from pytracer_amd import _abi as A

pytestmark = pytest.mark.gpu

PLAIN = {'PT_TRACE_DYNAMIC': '0', 'PT_SHADE_DYNAMIC': '0', 'PT_TRACE_PACKET': '0', 'PT_PAIR_REFILL': '0',
         'PT_NO_HIT_CLASSES': '1', 'PT_OVERLAP': '0'}
VARIANTS = [
    {'PT_OVERLAP': '0'},                                                    # the defaults on one stream
    {'PT_OVERLAP': '0', 'PT_BATCH': '4096'},                                # many wavefront batches
    {'PT_OVERLAP': '0', 'PT_SHADE_LOBESETS': '2'},                          # the diffuse-only kernel for the back hit list on every level (default: large levels only)
    {'PT_OVERLAP': '1', 'PT_SHADE_LOBESETS': '2', 'PT_BATCH': '4096'},
    {'PT_OVERLAP': '0', 'PT_TRACE_PACKET': '3'},                            # level-0 shadow rays per wave everywhere
    {'PT_OVERLAP': '0', 'PT_TRACE_PAIR': '3', 'PT_PAIR_REFILL': '1', 'PT_REFILL_MIN': '5'},   # pair walk, eager refill (half-size records)
    {'PT_OVERLAP': '0', 'PT_TRACE_PAIR': '3', 'PT_PAIR_REFILL': '1', 'PT_REFILL_MIN': '5', 'PT_PAIR_HALF': '0'},   # ... full-size records
    {'PT_OVERLAP': '0', 'PT_TRACE_PAIR': '3', 'PT_PAIR_REFILL': '1', 'PT_TRACE_PACKET': '0', 'PT_PAIR_LDS_LEVELS': '2'},
    {'PT_OVERLAP': '0', 'PT_TRACE_TOP': '1', 'PT_TRACE_PAIR': '0'},
    {'PT_OVERLAP': '0', 'PT_TRACE_META': '4'},                              # k_trace_meta_s (16-bit stacks, packed node words, 4 x 256 threads per CU) at sizes below its default threshold
    {'PT_OVERLAP': '0', 'PT_TRACE_META': '4', 'PT_TRACE_META_SHORT': '1'},   # ... its 512 x 2 form with 8-byte node words
    {'PT_OVERLAP': '0', 'PT_TRACE_META': '4', 'PT_TRACE_META_SHORT': '0'},   # ... the 1024-thread form with 4-byte stacks
    {'PT_OVERLAP': '1', 'PT_TRACE_PAIR': '3', 'PT_SORT_RAYS': '3', 'PT_SORT_WINDOW': '256'},   # ray binning before the pair walks
    # the two-stream schedule (the default, and the headline's): double-buffered shadow queues, ev_shade / ev_any
    # ordering, concurrent closest-hit and any-hit traces on split spill areas, escape radiance added in k_splat.
    # The order of the atomic adds into L is free there: radiance to 1e-12 instead of 1e-13.
    {'PT_OVERLAP': '1'},
    {'PT_OVERLAP': '2'},                                                    # ... also with measured-BRDF lobes
    {'PT_OVERLAP': '2', 'PT_BATCH': '4096'},                                # any_pending carried across batches
    {'PT_OVERLAP': '2', 'PT_TRACE_PAIR': '3', 'PT_PAIR_REFILL': '1', 'PT_REFILL_MIN': '5', 'PT_BATCH': '20000'},
]
SCENES = [('head_path_32x32x4.npz', A.PT_INTEG_PATH), ('head_direct_24x24x4.npz', A.PT_INTEG_DIRECT),
          ('c4mini_path_24x24x4.npz', A.PT_INTEG_PATH), ('tex2mini_whitted_24x24x1.npz', A.PT_INTEG_WHITTED),
          ('specmini_path_24x24x4.npz', A.PT_INTEG_PATH), ('c4mini_direct_24x24x1.npz', A.PT_INTEG_DIRECT),
          ('measmini_path_24x24x4.npz', A.PT_INTEG_PATH), ('tex2mini_path_24x24x4.npz', A.PT_INTEG_PATH),
          ('uvmini_path_24x24x4.npz', A.PT_INTEG_DIRECT), ('wrinkmini_path_24x24x4.npz', A.PT_INTEG_PATH)]


def render(torch, env, flat, d, kind, win, spp, seed):
    from pytracer_amd._lib import Context
    # the knobs stay set for the whole render: some are read when the context is created, PT_BATCH at every pt_render
    os.environ.update(env)
    c = None
    try:
        c = Context(0)
        c.upload(flat)
        cam, film = pa.camera_film_from_npz(dict(d))
        smp = pa.sampler_desc(0, film.x_cnt, 0, film.y_cnt, spp[0], spp[1], True, seed)
        tile = pa.tile_desc(win[0], win[1], win[2], win[3], 0, spp[0] * spp[1])
        n = (win[1] - win[0]) * (win[3] - win[2]) * spp[0] * spp[1]
        acc = torch.zeros((film.y_cnt, film.x_cnt, 4), dtype=torch.float64, device='cuda')
        L = torch.zeros((n, 3), dtype=torch.float64, device='cuda')
        st = A.pt_stats()
        c.render(cam, film, smp, pa.integrator_desc(kind, 5, False), tile, acc, sample_L=L, stats=st)
        torch.cuda.synchronize()
        return acc.cpu().numpy(), L.cpu().numpy(), st.as_dict()
    finally:
        for k in env:
            del os.environ[k]
        if c is not None:
            c.close()


@pytest.mark.parametrize('name,kind', SCENES)
def test_schedules_change_nothing(golden_dir, name, kind):
    import torch
    path = os.path.join(golden_dir, name)
    if not os.path.exists(path):
        pytest.skip('fixture %s not present' % name)
    d = dict(np.load(path))
    scene = d if 'n_materials' in d else dict(np.load(os.path.join(golden_dir, 'head_mesh_scene.npz')))   # head_* fixtures
    flat = pa.flat_scene_from_npz(scene)
    rng = np.random.default_rng(zlib.crc32(name.encode()) % 1000)       # the same window in every process
    cam, film = pa.camera_film_from_npz(dict(d))
    W, H = film.x_cnt, film.y_cnt
    x0, y0 = int(rng.integers(0, W // 3)), int(rng.integers(0, H // 3))
    win = (x0, min(W, x0 + 2 * W // 3 + 1), y0, min(H, y0 + 2 * H // 3 - 1))
    spp = (12, 9)                                          # 108 samples per pixel: chunks straddle pixels
    base = render(torch, PLAIN, flat, d, kind, win, spp, seed=7)
    assert base[2]['rays_closest'] >= (win[1] - win[0]) * (win[3] - win[2]) * 108
    for env in VARIANTS:
        acc, L, st = render(torch, env, flat, d, kind, win, spp, seed=7)
        assert st == base[2], (env, win, st, base[2])
        rtol = 1e-13 if env.get('PT_OVERLAP') == '0' else 1e-12
        np.testing.assert_allclose(L, base[1], rtol=rtol, atol=0, err_msg='%s window %s' % (env, win))
        np.testing.assert_allclose(acc, base[0], rtol=1e-11, atol=1e-14, err_msg='%s window %s' % (env, win))


def test_overlap_on_a_render_whose_launches_really_overlap(golden_dir):
    """The two-stream schedule at a size where the any-hit trace of level k does run beside level k + 1 (512^2 x 4 spp,
    1 M paths; and in five wavefront batches): census equal to the one-stream render, radiance to 1e-12."""
    import torch
    from pytracer_amd import host
    from pytracer_amd._lib import Context
    flat = pa.flat_scene_from_npz(dict(np.load(os.path.join(golden_dir, 'head_mesh_scene.npz'))))
    R = 512
    cam, film = host.head_mesh_camera_film(R, R)
    smp = pa.sampler_desc(0, R, 0, R, 2, 2, True, 11)
    tile = pa.tile_desc(0, R, 0, R, 0, 4)
    out = []
    for env in ({'PT_OVERLAP': '0'}, {'PT_OVERLAP': '1'}, {'PT_OVERLAP': '1', 'PT_BATCH': '230000'}):
        os.environ.update(env)
        c = None
        try:
            c = Context(0)
            c.upload(flat)
            acc = torch.zeros((film.y_cnt, film.x_cnt, 4), dtype=torch.float64, device='cuda')
            L = torch.zeros((R * R * 4, 3), dtype=torch.float64, device='cuda')
            st = A.pt_stats()
            c.render(cam, film, smp, pa.integrator_desc(A.PT_INTEG_PATH, 5, False), tile, acc, sample_L=L, stats=st)
            torch.cuda.synchronize()
            out.append((acc.cpu().numpy(), L.cpu().numpy(), st.as_dict()))
        finally:
            for k in env:
                del os.environ[k]
            if c is not None:
                c.close()
    for acc, L, st in out[1:]:
        assert st == out[0][2]
        np.testing.assert_allclose(L, out[0][1], rtol=1e-12, atol=0)
        np.testing.assert_allclose(acc, out[0][0], rtol=1e-11, atol=1e-14)
