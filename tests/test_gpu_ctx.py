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
