"""§8(f) rank 4 on the GPU: pt_film_to_u8 against the oracle and the reference's own pictures; GpuSamplerRenderer
with image='gpu', progressive output and checkpoint/resume."""
import os

import numpy as np
import pytest
import torch

import pytracer_amd as pa
from pytracer_amd import _abi as A
from pytracer_amd._lib import Context
from pytracer_amd.renderer import GpuSamplerRenderer
import oracle_lib as O
import standin

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def ctx():
    c = Context(0)
    yield c
    c.close()


@pytest.mark.parametrize('name', ['head_direct_40x40x1.npz', 'head_path_32x32x4.npz', 'c4mini_path_24x24x4.npz',
                                  'xtra_whitted.npz'])
def test_reference_picture_from_reference_accumulators(ctx, golden_dir, name):
    """pt_film_resolve + pt_film_to_u8 on the reference's accumulators = the bytes the reference wrote."""
    d = np.load(os.path.join(golden_dir, name))
    acc = np.concatenate([d['film_Lxyz'], d['film_wsum'][..., None]], axis=2)
    rgb = ctx.film_resolve(torch.from_numpy(np.ascontiguousarray(acc)).cuda())
    got = ctx.film_to_u8(rgb).cpu().numpy()
    assert np.array_equal(got, d['img8'])


@pytest.mark.parametrize('shape', [(1, 1), (17, 9), (257, 300), (1080, 1920)])
def test_film_to_u8_is_the_oracles(ctx, shape):
    rng = np.random.default_rng(shape[0])
    rgb = rng.uniform(0., 3., shape + (3,))
    rgb[:, :, 1] = 0.                                  # coef == 0
    rgb[shape[0] // 2, shape[1] // 3, 2] = -.75        # unclamped pixel
    rgb[-1, -1, 0] = 7.5                               # the maximum in the last element
    t = torch.from_numpy(rgb).cuda()
    got = ctx.film_to_u8(t).cpu().numpy()
    assert np.array_equal(got, O.film_to_u8(rgb))
    assert np.array_equal(t.cpu().numpy(), rgb)        # input not modified (the reference overwrites its copy)
    neg = -rng.uniform(.5, 2., shape + (3,))           # all-negative channels: key ordering of negative values
    assert np.array_equal(ctx.film_to_u8(torch.from_numpy(neg).cuda()).cpu().numpy(), O.film_to_u8(neg))
    # second call on the same context: the channel maxima are reset
    small = rgb * .01
    assert np.array_equal(ctx.film_to_u8(torch.from_numpy(small).cuda()).cpu().numpy(), O.film_to_u8(small))


def test_film_to_u8_empty(ctx):
    out = ctx.film_to_u8(torch.zeros((0, 0, 3), dtype=torch.float64, device='cuda'))
    assert tuple(out.shape) == (0, 0, 3)


def _setup(golden_dir, tmp_path, nx=3, ny=2):
    d = dict(np.load(os.path.join(golden_dir, 'head_mesh_scene.npz')))
    scene = standin.build_scene(d)
    camera = standin.build_camera(d)
    camera.film.filename = str(tmp_path / 'out.png')
    sampler = standin.build_sampler(0, 128, 0, 128, nx, ny)
    return d, scene, camera, sampler


def film_acc(film):
    W, H = film.xPixel_cnt, film.yPixel_cnt
    acc = np.zeros((H, W, 4))
    for x in range(W):
        for y in range(H):
            acc[y, x, 0:3] = film.pixels[x][y].Lxyz
            acc[y, x, 3] = film.pixels[x][y].weight_sum
    return acc


def test_renderer_writes_the_png_itself(golden_dir, tmp_path):
    import PIL.Image
    d, scene, camera, sampler = _setup(golden_dir, tmp_path)
    ren = GpuSamplerRenderer(sampler, camera, standin.build_integrator('direct'), None, device=0, seed=3, image='gpu')
    ren.render(scene)
    assert camera.film.written == 0                    # the film's own (slow) write_image was not needed
    with PIL.Image.open(camera.film.filename) as pic:
        png = np.array(pic)
    assert png.shape == (128, 128, 3) and np.array_equal(png, ren.image8)
    # the picture is the conversion of the film the renderer handed back
    want = O.film_to_u8(O.film_resolve(film_acc(camera.film)))
    assert np.array_equal(png, want)
    assert png.max(axis=(0, 1)).tolist() == [255, 255, 255] or png.max() >= 254


def test_checkpoint_resume_after_a_kill(golden_dir, tmp_path):
    d, scene, camera, sampler = _setup(golden_dir, tmp_path)
    ck = str(tmp_path / 'render.ckpt.npz')
    integ = standin.build_integrator('path')

    class Killed(Exception):
        pass

    seen = []

    def die_after_two(done, total):
        seen.append((done, total))
        if done >= 4:
            raise Killed()

    r1 = GpuSamplerRenderer(sampler, camera, integ, None, device=0, seed=5, image='gpu', checkpoint=ck,
                            checkpoint_every=2, on_piece=die_after_two)
    with pytest.raises(Killed):
        r1.render(scene)
    assert seen == [(2, 6), (4, 6)] and os.path.exists(ck)
    assert os.path.exists(camera.film.filename)        # progressive picture of the first 4 sample indices
    assert np.abs(film_acc(camera.film)).max() == 0.   # the film object is only filled at the end

    d3, scene3, camera3, sampler3 = _setup(golden_dir, tmp_path)
    r2 = GpuSamplerRenderer(sampler3, camera3, integ, None, device=0, seed=5, image='gpu', checkpoint=ck,
                            checkpoint_every=2)
    r2.render(scene3)
    assert r2.resumed_from == 4 and not os.path.exists(ck)
    assert r2.stats.camera_samples == 128 * 128 * 2    # only the missing piece was rendered

    d4, scene4, camera4, sampler4 = _setup(golden_dir, tmp_path)
    whole = GpuSamplerRenderer(sampler4, camera4, integ, None, device=0, seed=5, write_image=False)
    whole.render(scene4)
    a, b = film_acc(camera3.film), film_acc(camera4.film)
    assert np.abs(a - b).max() <= 1e-11 * np.abs(b).max()

    # a checkpoint of some other render under the same name is ignored, not resumed
    from pytracer_amd import imageio as IO
    IO.Checkpoint(ck, 'some other render').save(np.ones_like(b), 4)
    d5, scene5, camera5, sampler5 = _setup(golden_dir, tmp_path)
    r3 = GpuSamplerRenderer(sampler5, camera5, integ, None, device=0, seed=5, write_image=False, checkpoint=ck,
                            checkpoint_every=3)
    r3.render(scene5)
    assert r3.resumed_from is None
    c = film_acc(camera5.film)
    assert np.abs(c - b).max() <= 1e-11 * np.abs(b).max()
