"""Compatibility shim for importing the reference pyTracer (Python 3.5-era, numpy 1.13)
under Python 3.10 / numpy 2.x.  Used ONLY in the build container to generate golden
fixtures (tests/golden/*.npz); never shipped to / used on the GPU box.  No reference
file is modified.  See SURVEY.md §8(c).
"""
import sys, collections, collections.abc
for n in ('Iterable', 'Sequence', 'MutableSequence', 'Mapping', 'Callable'):
    setattr(collections, n, getattr(collections.abc, n))
import numpy as np
if not hasattr(np, 'float'):
    np.float = float
if not hasattr(np, 'int'):
    np.int = int
sys.path.insert(0, '/root/reference')
sys.dont_write_bytecode = True
import pytracer.reflection  # noqa: E402
_bdfmod = sys.modules['pytracer.reflection.bdf.bdf']
_bdfmod.UINT = lambda v: np.uint32(int(v) & 0xFFFFFFFF)
import pytracer.renderer.renderer as _rr  # noqa: E402
_rr.util.progress_reporter = lambda *a, **k: None
