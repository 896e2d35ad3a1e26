"""Film output on the host side of the boundary: the file half of the reference's
``imageio.write_image`` (pytracer/utility/imageio.py:53-93) and accumulator checkpoints.

The array half of ``write_image`` — per-channel ``255 * c / max(c)`` and ``astype('uint8')``,
imageio.py:60-65 — runs on the GPU (``pt_film_to_u8``); what is left here is the PNG itself, same
arguments and same error behaviour as the reference:

* full-resolution film: the bytes are saved as an RGB PNG and returned (imageio.py:68-73);
* cropped film: the reference opens the existing full image to update the crop window
  (imageio.py:76-89).  A missing file raises ``RuntimeError`` (imageio.py:76-78).  With the file
  present the reference then fails in numpy: it assigns the [yRes, xRes, 3] crop to the slice
  ``img[xOffset:xOffset+xRes-1, yOffset:yOffset+yRes-1]`` — x and y swapped and one short
  (imageio.py:84), a shape numpy cannot broadcast — so it raises ``ValueError`` for every crop
  window.  ``update='reference'`` (default) keeps that behaviour; ``update='window'`` writes the crop
  where it belongs, ``img[yOffset:yOffset+yRes, xOffset:xOffset+xRes]``, for users who want the
  feature rather than the quirk.

Checkpoints have no counterpart in the reference (a killed render loses everything, renderer.py:
97-111): ``Checkpoint`` stores the film accumulators of a partly rendered tile together with the
next sample index and a key of everything that determines the image; see GpuSamplerRenderer.
"""
import hashlib
import os

import numpy as np


def write_image(filename, rgb8, alpha, xRes, yRes, xFullRes, yFullRes, xOffset, yOffset, update='reference'):
    """imageio.write_image after its array conversion.  rgb8: uint8 [yRes, xRes, 3] (host)."""
    import PIL.Image
    if filename is None or rgb8 is None:
        raise IOError('write_image(): filename and rgb cannot be None')          # imageio.py:57-59
    rgb8 = np.ascontiguousarray(rgb8, np.uint8)
    if alpha is not None:
        raise NotImplementedError('write_image(): alpha is not supported by the reference either (imageio.py:91-93)')
    if xRes == xFullRes and yRes == yFullRes:
        PIL.Image.fromarray(rgb8, mode='RGB').save(filename, 'PNG')
        return rgb8.copy()
    if not os.path.exists(filename):
        raise RuntimeError('write_image(): cannot find {} to update'.format(filename))
    with PIL.Image.open(filename) as pic:
        img = np.array(pic)
    if update == 'reference':
        img[xOffset:xOffset + xRes - 1, yOffset:yOffset + yRes - 1, :] = rgb8    # imageio.py:84, raises ValueError
    elif update == 'window':
        img[yOffset:yOffset + yRes, xOffset:xOffset + xRes, 0:3] = rgb8
    else:
        raise ValueError("update must be 'reference' or 'window'")
    PIL.Image.fromarray(img, mode='RGB').save(filename, 'PNG')
    return img.copy()


def write_film_image(film_obj, rgb8, update='reference'):
    """ImageFilm.write_image's call into imageio (film/film.py:258-260) for a film-shaped object."""
    return write_image(film_obj.filename, rgb8, None, int(film_obj.xPixel_cnt), int(film_obj.yPixel_cnt),
                       int(film_obj.xResolution), int(film_obj.yResolution),
                       int(film_obj.xPixel_start), int(film_obj.yPixel_start), update=update)


def render_key(flat, cam, film, smp, integ, tile, sampler_mode='device', table=None):
    """Digest of everything that determines the accumulators of a tile: scene arrays (the measured BRDF
    tables included: a measured lobe's packed record only names its sample range), descriptors, and for
    sampler_mode='host' the host sampler's sample table."""
    h = hashlib.sha256()
    for a in (flat.node_bounds, flat.node_meta, flat.tri_verts, flat.tri_mat, flat.tri_flip):
        h.update(np.ascontiguousarray(a).tobytes())
    h.update(bytes(flat._materials))          # packed pt_material / pt_light records (zero-filled padding)
    h.update(bytes(flat._lights))
    h.update(bytes(flat._textures))
    if flat.tri_uv is not None:
        h.update(flat.tri_uv.tobytes())
    for a in (getattr(flat, 'brdf_p', None), getattr(flat, 'brdf_v', None)):
        if a is not None and len(a):
            h.update(np.ascontiguousarray(a).tobytes())
    if table is not None:
        h.update(np.ascontiguousarray(table).tobytes())
    for d in (cam, film, smp, integ, tile):
        h.update(bytes(d))
    h.update(sampler_mode.encode())
    return h.hexdigest()


class Checkpoint(object):
    """Accumulator checkpoint of one rank's tile.  File = npz {acc [y,x,4] f64, next_s, key}; written
    to a temporary name and renamed, so a kill during the write leaves the previous checkpoint."""

    def __init__(self, path, key):
        self.path, self.key = path, key

    def load(self, shape):
        """(acc, next_s) of a matching checkpoint, else None (absent, other render, other film size)."""
        if not os.path.exists(self.path):
            return None
        with np.load(self.path, allow_pickle=False) as d:
            if str(d['key']) != self.key or tuple(d['acc'].shape) != tuple(shape):
                return None
            return np.array(d['acc'], np.float64), int(d['next_s'])

    def save(self, acc, next_s):
        tmp = self.path + '.tmp.npz'
        np.savez(tmp, acc=np.asarray(acc, np.float64), next_s=np.int64(next_s), key=np.str_(self.key))
        os.replace(tmp, self.path)

    def remove(self):
        if os.path.exists(self.path):
            os.remove(self.path)


def sample_chunks(tile, every, start=None):
    """Split a tile's sample-index range [s0, s1) into pieces of `every` indices, from `start` on.
    Every sample's random stream depends only on (pixel, sample index, seed), so the pieces add up to
    the whole tile whatever the split."""
    from .flatten import tile_desc
    s = tile.s0 if start is None else max(int(start), tile.s0)
    every = int(every) if every else tile.s1 - tile.s0
    if every < 1:
        raise ValueError('checkpoint_every must be >= 1')
    while s < tile.s1:
        e = min(s + every, tile.s1)
        yield tile_desc(tile.x0, tile.x1, tile.y0, tile.y1, s, e, tile.y_step, tile.y_phase)
        s = e

