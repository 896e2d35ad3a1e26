"""§8(f) rank 4 on the CPU: the oracle's 8-bit conversion against the reference's own pictures, the PNG
half of write_image (pytracer_amd.imageio) with the reference's error behaviour, checkpoints."""
import os

import numpy as np
import pytest

import pytracer_amd as pa
from pytracer_amd import _abi as A
from pytracer_amd import imageio as IO
import oracle_lib as O


def numpy_u8(rgb):
    """imageio.py:60-65 in numpy itself."""
    rgb = rgb.copy()
    for i in range(3):
        coef = np.max(rgb[:, :, i])
        if not coef == 0.:
            rgb[:, :, i] = (255. * rgb[:, :, i]) / coef
    return rgb.astype('uint8')


@pytest.mark.parametrize('name', ['head_direct_40x40x1.npz', 'head_direct_24x24x4.npz', 'head_path_32x32x4.npz',
                                  'c4mini_path_24x24x4.npz', 'specmini_direct_24x24x1.npz', 'xtra_whitted.npz'])
def test_reference_picture_from_reference_accumulators(golden_dir, name):
    """film resolve + 8-bit conversion of the reference's own accumulators give the reference's own bytes
    (ImageFilm.write_image → imageio.write_image), exactly."""
    d = np.load(os.path.join(golden_dir, name))
    acc = np.concatenate([d['film_Lxyz'], d['film_wsum'][..., None]], axis=2)
    got = O.film_to_u8(O.film_resolve(acc))
    assert got.dtype == np.uint8 and got.shape == d['img8'].shape
    assert np.array_equal(got, d['img8'])


def test_conversion_is_numpys_on_awkward_values():
    rng = np.random.default_rng(5)
    rgb = rng.uniform(0., 3., (17, 9, 3))
    rgb[:, :, 1] = 0.                          # coef == 0: channel left alone
    rgb[3, 4, 2] = -.75                        # unclamped pixel (weight_sum == 0 branch of film.py:251)
    rgb[5, 5, 0] = 1e-320                      # subnormal
    assert np.array_equal(O.film_to_u8(rgb), numpy_u8(rgb))
    neg = -rng.uniform(.5, 2., (4, 4, 3))      # all-negative channel: coef < 0 flips signs
    assert np.array_equal(O.film_to_u8(neg), numpy_u8(neg))
    assert O.film_to_u8(np.zeros((0, 0, 3))).shape == (0, 0, 3)


def test_write_image_full_and_cropped(tmp_path):
    import PIL.Image
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, (12, 20, 3)).astype(np.uint8)
    fn = str(tmp_path / 'a.png')
    ret = IO.write_image(fn, img, None, 20, 12, 20, 12, 0, 0)
    assert np.array_equal(ret, img)
    with PIL.Image.open(fn) as pic:
        assert pic.mode == 'RGB' and np.array_equal(np.array(pic), img)
    with pytest.raises(IOError):
        IO.write_image(None, img, None, 20, 12, 20, 12, 0, 0)              # imageio.py:57-59
    with pytest.raises(NotImplementedError):
        IO.write_image(fn, img, np.ones((12, 20)), 20, 12, 20, 12, 0, 0)   # imageio.py:91-93
    crop = rng.integers(0, 256, (5, 8, 3)).astype(np.uint8)
    with pytest.raises(RuntimeError):                                      # imageio.py:76-78
        IO.write_image(str(tmp_path / 'missing.png'), crop, None, 8, 5, 20, 12, 4, 3)
    with pytest.raises(ValueError):                                        # imageio.py:84 cannot broadcast
        IO.write_image(fn, crop, None, 8, 5, 20, 12, 4, 3)
    with PIL.Image.open(fn) as pic:
        assert np.array_equal(np.array(pic), img)                          # untouched by the failed update
    ret = IO.write_image(fn, crop, None, 8, 5, 20, 12, 4, 3, update='window')
    want = img.copy()
    want[3:8, 4:12] = crop
    assert np.array_equal(ret, want)
    with PIL.Image.open(fn) as pic:
        assert np.array_equal(np.array(pic), want)


def test_checkpoint_file(tmp_path):
    ck = IO.Checkpoint(str(tmp_path / 'c.npz'), 'k1')
    assert ck.load((4, 5, 4)) is None
    acc = np.random.default_rng(0).normal(size=(4, 5, 4))
    ck.save(acc, 7)
    got = ck.load((4, 5, 4))
    assert got[1] == 7 and np.array_equal(got[0], acc)
    assert ck.load((5, 4, 4)) is None                                      # other film
    assert IO.Checkpoint(ck.path, 'k2').load((4, 5, 4)) is None            # other render
    ck.save(2 * acc, 9)                                                    # replaced atomically
    assert ck.load((4, 5, 4))[1] == 9 and not os.path.exists(ck.path + '.tmp.npz')
    ck.remove()
    assert not os.path.exists(ck.path)


def test_sample_chunks_cover_the_tile():
    t = pa.tile_desc(2, 9, 1, 5, 3, 16)
    parts = list(IO.sample_chunks(t, 4))
    assert [(p.s0, p.s1) for p in parts] == [(3, 7), (7, 11), (11, 15), (15, 16)]
    assert all((p.x0, p.x1, p.y0, p.y1) == (2, 9, 1, 5) for p in parts)
    assert [(p.s0, p.s1) for p in IO.sample_chunks(t, 4, start=11)] == [(11, 15), (15, 16)]
    assert [(p.s0, p.s1) for p in IO.sample_chunks(t, None)] == [(3, 16)]
    assert list(IO.sample_chunks(t, 4, start=16)) == []
    with pytest.raises(ValueError):
        list(IO.sample_chunks(t, -1))


def test_pieces_add_up_and_key_tells_renders_apart(golden_dir):
    """The resume logic with the oracle as the device: pieces of the sample range sum to the whole render;
    the key changes with anything that changes the image."""
    d = dict(np.load(os.path.join(golden_dir, 'head_mesh_scene.npz')))
    flat = pa.flat_scene_from_npz(d)
    osc = O.OracleScene(flat)
    cam, film = pa.camera_film_from_npz(d)
    smp = pa.sampler_desc(44, 68, 44, 68, 3, 2, True, 9)
    integ = pa.integrator_desc(A.PT_INTEG_PATH, 5, False)
    tile = pa.tile_desc(44, 68, 44, 68, 0, 6)
    whole, _ = osc.render(cam, film, smp, integ, tile, want_L=False)
    acc = np.zeros_like(whole)
    for piece in IO.sample_chunks(tile, 4):
        a, _ = osc.render(cam, film, smp, integ, piece, want_L=False)
        acc += a
    assert np.abs(acc - whole).max() <= 1e-12 * np.abs(whole).max()
    k = IO.render_key(flat, cam, film, smp, integ, tile)
    assert k == IO.render_key(flat, cam, film, smp, integ, tile)
    assert k != IO.render_key(flat, cam, film, pa.sampler_desc(44, 68, 44, 68, 3, 2, True, 10), integ, tile)
    assert k != IO.render_key(flat, cam, film, smp, pa.integrator_desc(A.PT_INTEG_DIRECT, 5, False), tile)
    assert k != IO.render_key(flat, cam, film, smp, integ, pa.tile_desc(44, 68, 44, 68, 0, 3))
    assert k != IO.render_key(flat, cam, film, smp, integ, tile, 'host')
    assert k != IO.render_key(flat, cam, film, smp, integ, tile, 'host', np.zeros((4, 35)))
    assert (IO.render_key(flat, cam, film, smp, integ, tile, 'host', np.zeros((4, 35))) !=
            IO.render_key(flat, cam, film, smp, integ, tile, 'host', np.ones((4, 35))))


def test_key_covers_measured_brdf_tables(golden_dir):
    """Two measured-BRDF tables with the same sample count give the same packed lobe record (first, count):
    the key must still tell them apart, or a checkpoint of one material would be resumed into the other."""
    from pytracer_amd import workloads
    flat = workloads.c5_head_measured(os.path.join(golden_dir, 'head_mesh_scene.npz'))
    d = dict(np.load(os.path.join(golden_dir, 'head_mesh_scene.npz')))
    cam, film = pa.camera_film_from_npz(d)
    smp = pa.sampler_desc(44, 68, 44, 68, 2, 2, True, 9)
    integ = pa.integrator_desc(A.PT_INTEG_PATH, 5, False)
    tile = pa.tile_desc(44, 68, 44, 68, 0, 4)
    k = IO.render_key(flat, cam, film, smp, integ, tile)
    assert len(flat.brdf_v) > 0
    keep = flat.brdf_v.copy()
    flat.brdf_v[0, 0] += 1e-3
    assert k != IO.render_key(flat, cam, film, smp, integ, tile)
    flat.brdf_v[:] = keep
    assert k == IO.render_key(flat, cam, film, smp, integ, tile)
    flat.brdf_p[1, 2] += 1e-3
    assert k != IO.render_key(flat, cam, film, smp, integ, tile)
