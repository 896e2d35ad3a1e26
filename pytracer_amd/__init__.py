"""pytracer_amd — MI355X (gfx950) back end for pyTracer's ray–scene intersection and
light-integration loop.  See DESIGN.md; the C ABI is declared in include/pytracer_hip.h."""
from . import _abi as abi                      # noqa: F401
from .flatten import (FlatScene, flatten_scene, flat_scene_from_npz, lobes_from_params,   # noqa: F401
                      camera_desc, film_desc, sampler_desc, integrator_desc, tile_desc,
                      camera_film_from_npz)

__all__ = ['abi', 'FlatScene', 'flatten_scene', 'flat_scene_from_npz', 'lobes_from_params', 'camera_desc',
           'film_desc', 'sampler_desc', 'integrator_desc', 'tile_desc', 'camera_film_from_npz']
