"""GpuSamplerRenderer — drop-in for the reference's ``SamplerRenderer``
(pytracer/renderer/renderer.py:46-111) whose ``render(scene)`` runs the body of
``SamplerRendererTask.__call__`` (renderer.py:132-198) on MI355X through the C ABI.

Same constructor and call shape as the reference::

    renderer = GpuSamplerRenderer(sampler, camera, surf_integrator, None)
    renderer.render(scene)            # fills camera.film.pixels[x][y].Lxyz / weight_sum,
                                      # then calls camera.film.write_image() like the reference

Only public attributes of the reference objects are read (see pytracer_amd.flatten), so any
object graph shaped like the reference's works — the reference itself does not need to be
importable.  Multi-GPU: under torch.distributed every rank renders its shard (a contiguous range
of sample indices of every pixel, or a band of rows) and the film accumulators are summed with
one reduce (RCCL on GPUs); see ``shard_tile``.
"""
import numpy as np

from . import _abi as A
from . import flatten as F
from . import imageio as IO


def shard_tile(sampler, rank, world, mode='interleaved'):
    """The work shard of `rank` out of `world` (include/pytracer_hip.h pt_tile).

    mode 'interleaved' (default): rows rank, rank + world, rank + 2 world, … with all their samples — every rank
    sees rows from all over the image (balanced like a sample split) and every pixel keeps its samples on one rank
    (waves stay coherent, the per-pixel film pass is not repeated on every rank).  Measured on one MI355X as shard 0
    of 8 of the 1024^2 x 64 spp head-mesh render: 'samples' 14.7 ms (splat 1.9 ms instead of
    0.4: its per-pixel cost does not shrink with the sample count), contiguous 'rows' bands are unbalanced (the top
    band 6.4 ms, the head's bands far more).
    mode 'samples': all pixels, sample indices [s0, s1).  mode 'rows': a contiguous band of rows, all sample
    indices.  The film footprint of a sample crosses rows, which is why every rank keeps a FULL film
    accumulator and the shards are summed.  'samples' falls back to rows when spp < world.
    The reference's own decomposition hook is Sampler.compute_subwindow
    (sampler/sampler/__init__.py:87-111), unused there (renderer.py:97-109,135)."""
    spp = sampler.xs * sampler.ys
    if mode == 'interleaved':
        if world <= 1:
            return F.tile_desc(sampler.x_start, sampler.x_end, sampler.y_start, sampler.y_end, 0, spp)
        return F.tile_desc(sampler.x_start, sampler.x_end, sampler.y_start, sampler.y_end, 0, spp, world, rank)
    if mode not in ('samples', 'rows'):
        raise ValueError("shard mode must be 'interleaved', 'samples' or 'rows'")
    if mode == 'samples' and spp >= world:
        s0 = (spp * rank) // world
        s1 = (spp * (rank + 1)) // world
        return F.tile_desc(sampler.x_start, sampler.x_end, sampler.y_start, sampler.y_end, s0, s1)
    h = sampler.y_end - sampler.y_start
    y0 = sampler.y_start + (h * rank) // world
    y1 = sampler.y_start + (h * (rank + 1)) // world
    return F.tile_desc(sampler.x_start, sampler.x_end, y0, y1, 0, spp)


def host_sample_table(sampler, integrator, scene, n_dims):
    """Run the HOST sampler (any object with the reference's ``generate(samples)`` protocol,
    sampler/sampler/stratified.py:45-89) over its whole window and flatten every Sample into a
    table row (imageX, imageY, lens_u, lens_v, time, oneD..., twoD...).  This is how a user-defined
    Sampler subclass drives the GPU path."""
    class _Rec(object):
        pass

    n1d, n2d = [], []

    class _Req(object):           # the minimal Sample protocol request_samples() needs (sample.py:66-76)
        def add_1d(self, n):
            n1d.append(n)
            return len(n1d) - 1

        def add_2d(self, n):
            n2d.append(n)
            return len(n2d) - 1
    integrator.request_samples(sampler, _Req(), scene)
    spp = sampler.maximum_sample_cnt()
    samples = []
    for _ in range(spp):
        s = _Rec()
        s.imageX = s.imageY = s.lens_u = s.lens_v = s.time = 0.
        s.n1D, s.n2D = n1d, n2d
        s.oneD, s.twoD = [None] * len(n1d), [None] * len(n2d)
        samples.append(s)
    rows = []
    while sampler.generate(samples):
        for s in samples:
            v = [s.imageX, s.imageY, s.lens_u, s.lens_v, s.time]
            for a in s.oneD:
                v.extend(np.asarray(a, np.float64).ravel())
            for a in s.twoD:
                v.extend(np.asarray(a, np.float64).ravel())
            rows.append(v)
    table = np.asarray(rows, np.float64)
    if table.shape[1] != n_dims:
        raise ValueError('host sampler produced %d dims per sample, the integrator layout needs %d'
                         % (table.shape[1], n_dims))
    return table


class GpuSamplerRenderer(object):
    """Sample-driven renderer on the GPU; constructor signature of SamplerRenderer (renderer.py:52-56)."""

    def __init__(self, s, c, si, vi=None, device=None, seed=0, sampler_mode='device', pretest=False,
                 shard_mode='interleaved', write_image=True, image='film', checkpoint=None, checkpoint_every=None,
                 crop_update='reference', on_piece=None):
        """Beyond the reference's (s, c, si, vi):

        image: how rank 0 produces the picture when `write_image` is true.
          'film' — fill ``camera.film.pixels`` and call the film object's own ``write_image()``, i.e. the
                   reference's per-pixel Python loop (film/film.py:237-260): slow, but any Film subclass works;
          'gpu'  — resolve and tone-map on the GPU (pt_film_resolve, pt_film_to_u8) and save the PNG to
                   ``film.filename`` directly (pytracer_amd.imageio); ``film.pixels`` is still filled.
        checkpoint / checkpoint_every: path of an accumulator checkpoint and the number of sample indices
          per piece.  The tile is rendered piece by piece; after each piece the accumulators are saved
          (and, with image='gpu', the picture so far is written: progressive output).  A later render of
          the same scene/camera/sampler/integrator/seed finds the file and continues after the last
          finished piece; a checkpoint of anything else is ignored.  Ranks keep separate files
          (``<path>.rank<k>`` for world > 1).  The file is removed when the render completes.
          With image='gpu' the picture shows THIS render's accumulators; what an earlier render left in
          ``film.pixels`` is added to the pixels (like add_sample would) but only image='film' paints it.
        on_piece: callable(s_done, s_total) after every finished piece (progress reporting)."""
        if vi is not None:
            raise NotImplementedError('volume integrators are outside the supported hot path '
                                      '(the reference ships none: integrator/volume is empty)')
        if image not in ('film', 'gpu'):
            raise ValueError("image must be 'film' or 'gpu'")
        self.sampler = s
        self.camera = c
        self.surf_integrator = si
        self.vol_integrator = vi
        self.device = device
        self.seed = seed
        self.sampler_mode = sampler_mode      # 'device': in-kernel counter stream; 'host': sampler.generate()
        self.pretest = pretest
        self.shard_mode = shard_mode
        self.write_image = write_image
        self.image = image
        self.checkpoint = checkpoint
        self.checkpoint_every = checkpoint_every
        self.crop_update = crop_update
        self.on_piece = on_piece
        self.stats = None
        self.image8 = None                    # the last 8-bit picture (image='gpu'), numpy [y, x, 3]
        self.resumed_from = None              # sample index a checkpoint let this render start from
        self._ctx = None

    # -- the part of the reference API that still makes sense on a batch machine -------------
    def transmittance(self, scene, ray, sample, rng=None):
        """SamplerRenderer.transmittance without a volume integrator (renderer.py:77-82)."""
        return np.ones(3)

    def li(self, scene, ray, sample, isect, rng=None):
        raise NotImplementedError('per-ray li() is a host-loop API; use render(scene) '
                                  '(batched, on the GPU) or Context.intersect for ray batches')

    # -----------------------------------------------------------------------------------------
    def render(self, scene):
        import torch
        from ._lib import Context
        dist = torch.distributed if (torch.distributed.is_available() and torch.distributed.is_initialized()) else None
        rank = dist.get_rank() if dist else 0
        world = dist.get_world_size() if dist else 1
        dev = self.device
        if dev is None:
            dev = torch.cuda.current_device()
        dev = torch.device('cuda', int(dev) if not isinstance(dev, torch.device) else dev.index)

        flat = F.flatten_scene(scene)
        cam = F.camera_desc_from_object(self.camera)
        film_obj = self.camera.film
        film = F.film_desc_from_object(film_obj)
        smp = F.sampler_desc_from_object(self.sampler, self.seed)
        integ = F.integrator_desc_from_object(self.surf_integrator, self.pretest)
        tile = shard_tile(smp, rank, world, self.shard_mode)

        if self._ctx is None or self._ctx.device != dev.index:
            self._ctx = Context(dev.index)
        ctx = self._ctx
        ctx.upload(flat)
        table = table_host = None
        if self.sampler_mode == 'host':
            table_host = host_sample_table(self.sampler, self.surf_integrator, scene, flat.table_dims(integ.kind))
            table = torch.from_numpy(table_host).to(dev)
        acc = torch.zeros((film.y_cnt, film.x_cnt, 4), dtype=torch.float64, device=dev)
        self.stats = A.pt_stats()
        ckpt, start = None, None
        self.resumed_from = None
        if self.checkpoint:
            path = self.checkpoint if world == 1 else '%s.rank%d' % (self.checkpoint, rank)
            ckpt = IO.Checkpoint(path, IO.render_key(flat, cam, film, smp, integ, tile, self.sampler_mode, table_host))
            found = ckpt.load(tuple(acc.shape))
            if found is not None:
                acc.copy_(torch.from_numpy(found[0]))
                start = self.resumed_from = found[1]
        progressive = ckpt is not None and self.write_image and self.image == 'gpu' and world == 1
        with torch.cuda.device(dev):
            for piece in IO.sample_chunks(tile, self.checkpoint_every if ckpt else None, start):
                ctx.render(cam, film, smp, integ, piece, acc, table=table, stats=self.stats)
                if ckpt is not None and piece.s1 < tile.s1:
                    torch.cuda.synchronize()
                    ckpt.save(acc.cpu().numpy(), piece.s1)
                    if progressive:
                        self._write_gpu_image(ctx, film_obj, acc)
                if self.on_piece is not None:
                    self.on_piece(piece.s1 - tile.s0, tile.s1 - tile.s0)
            if dist is not None and world > 1:
                dist.reduce(acc, dst=0)        # the only collective: sum of film accumulators
            torch.cuda.synchronize()
            if rank == 0:
                write_film(film_obj, acc.cpu().numpy())
                if self.write_image:
                    if self.image == 'gpu':
                        self._write_gpu_image(ctx, film_obj, acc)
                    else:
                        film_obj.write_image()
        if ckpt is not None:
            ckpt.remove()
        return None                            # SamplerRenderer.render returns None (renderer.py:84-111)

    def _write_gpu_image(self, ctx, film_obj, acc):
        """ImageFilm.write_image (film/film.py:237-260) with both array passes on the device."""
        rgb8 = ctx.film_to_u8(ctx.film_resolve(acc)).cpu().numpy()
        self.image8 = IO.write_film_image(film_obj, rgb8, update=self.crop_update)


def write_film(film_obj, acc):
    """Hand the accumulators back into ImageFilm.pixels (film/film.py:93-104,122-125):
    pixels[x][y].Lxyz (3,) and .weight_sum, indexed x-first."""
    H, W = acc.shape[0], acc.shape[1]
    for x in range(W):
        col = film_obj.pixels[x]
        for y in range(H):
            p = col[y]
            p.Lxyz = p.Lxyz + acc[y, x, 0:3]          # add_sample accumulates (film.py:187-188)
            p.weight_sum = p.weight_sum + float(acc[y, x, 3])
