"""ctypes binding of pytracer_amd/csrc/libpytracer_hip.so (the C ABI of include/pytracer_hip.h).

The library is built in-tree by ``__graft_entry__.build()`` / ``make -C pytracer_amd/csrc``.
There is no fallback: if it is missing or does not load, importing this module's users fails.
"""
import ctypes as C
import os

from . import _abi as A

LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'csrc', 'libpytracer_hip.so')
if os.environ.get('PYTRACER_HIP_LIB'):        # kernel A/B experiments only (scripts/ab_libs.sh)
    LIB_PATH = os.path.abspath(os.environ['PYTRACER_HIP_LIB'])

EXPORTS = ('pt_version', 'pt_ctx_create', 'pt_ctx_destroy', 'pt_last_error', 'pt_scene_upload', 'pt_intersect',
           'pt_intersect_static', 'pt_render', 'pt_film_resolve', 'pt_film_to_u8', 'pt_set_profiling', 'pt_launch_log', 'pt_kernel_name', 'pt_bvh_build', 'pt_build_last_error',
           'pt_loop_subdiv', 'pt_subdiv_last_error')

_lib = None


class PtError(RuntimeError):
    def __init__(self, code, msg):
        RuntimeError.__init__(self, 'pytracer_hip error %d: %s' % (code, msg))
        self.code = code


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError('HIP extension %s is missing: run `python -c "import __graft_entry__ as g; g.build()"` '
                              'or `make -C pytracer_amd/csrc` (there is no CPU fallback)' % LIB_PATH)
        # torch-ROCm ships its own libamdhip64/libhsa-runtime64 (same SONAME as /opt/rocm's).  Two HIP
        # runtimes in one process fight over the device ("no ROCm-capable device is detected" in the
        # second one), so load torch's first: ours then binds to the already-loaded SONAME and shares
        # the runtime that owns the tensors we are handed.
        import torch  # noqa: F401
        L = C.CDLL(LIB_PATH)
        L.pt_version.restype = C.c_int
        L.pt_ctx_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
        L.pt_ctx_destroy.argtypes = [C.c_void_p]
        L.pt_last_error.restype = C.c_char_p
        L.pt_last_error.argtypes = [C.c_void_p]
        L.pt_scene_upload.argtypes = [C.c_void_p, C.POINTER(A.pt_scene_desc)]
        L.pt_intersect.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                   C.c_void_p, C.POINTER(A.pt_stats), C.c_void_p]
        L.pt_intersect_static.argtypes = L.pt_intersect.argtypes
        L.pt_render.argtypes = [C.c_void_p, C.POINTER(A.pt_camera), C.POINTER(A.pt_film_desc),
                                C.POINTER(A.pt_sampler_desc), C.POINTER(A.pt_integrator_desc), C.POINTER(A.pt_tile),
                                C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.POINTER(A.pt_stats), C.c_void_p]
        L.pt_set_profiling.argtypes = [C.c_void_p, C.c_int]
        L.pt_launch_log.argtypes = [C.c_void_p, C.POINTER(A.pt_launch_record), C.c_int32, C.POINTER(C.c_int32)]
        L.pt_kernel_name.restype = C.c_char_p
        L.pt_kernel_name.argtypes = [C.c_int32]
        L.pt_film_resolve.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]
        L.pt_film_to_u8.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]
        L.pt_bvh_build.argtypes = [C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                   C.POINTER(C.c_int64), C.POINTER(C.c_int32)]
        L.pt_build_last_error.restype = C.c_char_p
        L.pt_loop_subdiv.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_int64,
                                     C.c_void_p, C.c_int64, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
        L.pt_subdiv_last_error.restype = C.c_char_p
        if L.pt_version() != A.PT_ABI_VERSION:
            raise ImportError('libpytracer_hip.so ABI %d != expected %d' % (L.pt_version(), A.PT_ABI_VERSION))
        _lib = L
    return _lib


class Context(object):
    """One pt_ctx (one GPU).  Device buffers are torch-ROCm tensors owned by the caller."""

    def __init__(self, device=0):
        self._h = C.c_void_p()
        self.device = int(device)
        rc = lib().pt_ctx_create(self.device, C.byref(self._h))
        if rc != 0:
            raise PtError(rc, 'pt_ctx_create(device=%d): %s' % (self.device, lib().pt_last_error(None).decode('utf-8', 'replace')))
        self._scene = None

    def close(self):
        if getattr(self, '_h', None):
            lib().pt_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            raise PtError(rc, lib().pt_last_error(self._h).decode('utf-8', 'replace'))

    def set_profiling(self, enable):
        """0 / False off; 1 / True: every launch timed and logged; 2: only the trace launches timed"""
        self._check(lib().pt_set_profiling(self._h, int(enable)))

    def launch_log(self):
        """The launches of the last profiled render / intersect (pt_launch_log) as a list of dicts."""
        n = C.c_int32(0)
        self._check(lib().pt_launch_log(self._h, None, 0, C.byref(n)))
        buf = (A.pt_launch_record * max(n.value, 1))()
        self._check(lib().pt_launch_log(self._h, buf, n.value, C.byref(n)))
        return [{'stage': A.STAGE_NAMES[r.stage], 'kernel': lib().pt_kernel_name(r.kernel).decode(), 'level': r.level,
                 'batch': r.batch, 'rays': int(r.rays), 'node_visits': int(r.node_visits), 'tri_tests': int(r.tri_tests),
                 'ms': float(r.ms)} for r in buf[:n.value]]

    def upload(self, flat):
        d = flat.desc()
        self._check(lib().pt_scene_upload(self._h, C.byref(d)))
        self._scene = flat

    def _stream(self, stream):
        if stream is None:
            import torch
            return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)   # the CONTEXT's device
        return C.c_void_p(int(stream))

    def intersect(self, rays, any_hit=False, stats=None, stream=None, want_bary=True, static=False, out=None):
        """rays: torch float64 cuda tensor [8, n] (ox,oy,oz,dx,dy,dz,mint,maxt).
        Returns (prim int32[n], t, b1, b2) — t/b1/b2 are None for any_hit.
        static=True calls pt_intersect_static (the entry point that allocates nothing); `out` = a
        (prim, t, b1, b2) tuple of preallocated tensors makes the whole call allocation-free."""
        import torch
        assert rays.is_cuda and rays.dtype == torch.float64 and rays.dim() == 2 and rays.shape[0] == 8
        assert rays.device.index == self.device, 'rays live on cuda:%s, the context on cuda:%d' % (rays.device.index, self.device)
        if out is None:
            rays = rays.contiguous()
        else:
            assert rays.is_contiguous()
        n = rays.shape[1]
        if out is not None:
            prim, t, b1, b2 = out
            assert prim.dtype == torch.int32 and prim.is_cuda and prim.is_contiguous() and prim.numel() >= n
            for x in (t, b1, b2):
                assert x is None or (x.dtype == torch.float64 and x.is_cuda and x.is_contiguous() and x.numel() >= n)
        else:
            prim = torch.empty(n, dtype=torch.int32, device=rays.device)
            t = b1 = b2 = None
            if not any_hit:
                t = torch.empty(n, dtype=torch.float64, device=rays.device)
                if want_bary:
                    b1 = torch.empty_like(t)
                    b2 = torch.empty_like(t)
        fn = lib().pt_intersect_static if static else lib().pt_intersect
        self._check(fn(self._h, rays.data_ptr(), n, int(bool(any_hit)), prim.data_ptr(),
                                       t.data_ptr() if t is not None else None,
                                       b1.data_ptr() if b1 is not None else None,
                                       b2.data_ptr() if b2 is not None else None,
                                       C.byref(stats) if stats is not None else None, self._stream(stream)))
        return prim, t, b1, b2

    def render(self, cam, film, sampler, integ, tile, film_accum, table=None, sample_L=None, stats=None, stream=None):
        """film_accum: torch float64 cuda [y_cnt, x_cnt, 4], ADDED to.  table: optional torch float64 cuda
        [n_window_samples, dims] (parity mode).  sample_L: optional torch float64 cuda [n_tile_samples, 3]."""
        import torch
        if film_accum is not None:
            assert film_accum.is_cuda and film_accum.dtype == torch.float64 and film_accum.is_contiguous()
            assert tuple(film_accum.shape) == (film.y_cnt, film.x_cnt, 4)
        dims = 0
        if table is not None:
            assert table.is_cuda and table.dtype == torch.float64 and table.is_contiguous() and table.dim() == 2
            window = (sampler.x_end - sampler.x_start) * (sampler.y_end - sampler.y_start) * sampler.xs * sampler.ys
            assert table.shape[0] == window, (table.shape, window)
            dims = table.shape[1]
        if sample_L is not None:
            n = tile.n_samples()
            assert sample_L.is_cuda and sample_L.dtype == torch.float64 and sample_L.is_contiguous()
            assert tuple(sample_L.shape) == (n, 3)
        self._check(lib().pt_render(self._h, C.byref(cam), C.byref(film), C.byref(sampler), C.byref(integ),
                                    C.byref(tile), table.data_ptr() if table is not None else None, dims,
                                    film_accum.data_ptr() if film_accum is not None else None,
                                    sample_L.data_ptr() if sample_L is not None else None,
                                    C.byref(stats) if stats is not None else None, self._stream(stream)))

    def film_resolve(self, film_accum, stream=None):
        import torch
        assert film_accum.is_cuda and film_accum.dtype == torch.float64 and film_accum.is_contiguous()
        y, x = film_accum.shape[0], film_accum.shape[1]
        rgb = torch.empty((y, x, 3), dtype=torch.float64, device=film_accum.device)
        self._check(lib().pt_film_resolve(self._h, film_accum.data_ptr(), x, y, rgb.data_ptr(), self._stream(stream)))
        return rgb

    def film_to_u8(self, rgb, stream=None):
        """imageio.write_image's 8-bit conversion (utility/imageio.py:60-65) of a resolved film:
        torch float64 cuda [y, x, 3] → torch uint8 cuda [y, x, 3]."""
        import torch
        assert rgb.is_cuda and rgb.dtype == torch.float64 and rgb.is_contiguous() and rgb.dim() == 3 and rgb.shape[2] == 3
        out = torch.empty(tuple(rgb.shape), dtype=torch.uint8, device=rgb.device)
        self._check(lib().pt_film_to_u8(self._h, rgb.data_ptr(), rgb.shape[1], rgb.shape[0], out.data_ptr(),
                                        self._stream(stream)))
        return out
