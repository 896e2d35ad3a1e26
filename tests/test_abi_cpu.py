"""CPU-side checks of the boundary: the shared library loads and exports every symbol that
include/pytracer_hip.h declares (no compute calls without a GPU), ctypes struct sizes match
the header, and the flattening logic reproduces the lobe lists of the reference materials."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

import pytracer_amd as pa
from pytracer_amd import _abi as A
from pytracer_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, 'include', 'pytracer_hip.h')


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(pt_[a-z0-9_]+)\s*\(', src)))


def test_header_symbols_exported():
    assert os.path.exists(_lib.LIB_PATH), 'build the HIP extension first (__graft_entry__.build())'
    names = declared_functions()
    assert set(names) == set(_lib.EXPORTS)
    L = C.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(L, n), n
    L.pt_version.restype = C.c_int
    assert L.pt_version() == A.PT_ABI_VERSION


def test_struct_sizes_match_header(tmp_path):
    names = ['pt_lobe', 'pt_material', 'pt_light', 'pt_scene_desc', 'pt_camera', 'pt_film_desc', 'pt_sampler_desc',
             'pt_integrator_desc', 'pt_tile', 'pt_stats']
    prog = '#include <stdio.h>\n#include "%s"\nint main(){%s return 0;}\n' % (
        HEADER, ''.join('printf("%%zu\\n", sizeof(%s));' % n for n in names))
    src = tmp_path / 'sz.c'
    src.write_text(prog)
    exe = tmp_path / 'sz'
    subprocess.check_call(['gcc', '-o', str(exe), str(src)])
    out = subprocess.check_output([str(exe)]).decode().split()
    for n, s in zip(names, out):
        assert C.sizeof(getattr(A, n)) == int(s), n


def test_uber_lobes_order_and_black_skipping():
    # examples/head_mesh.py:50-57 → [Lambert(.25), Microfacet(.25, Fr(1.5,1), Blinn(10)), SpecTrans(.25,1.5,1)]
    l = pa.lobes_from_params('uber', Kd=[.25] * 3, Ks=[.25] * 3, Kr=[0.] * 3, Kt=[.25] * 3, roughness=.1,
                             opacity=[1.] * 3, eta=1.5)
    assert [x[0] for x in l] == [A.PT_LOBE_LAMBERT, A.PT_LOBE_MICROFACET_BLINN, A.PT_LOBE_SPEC_TRANSMIT]
    assert l[1][2:5] == (1.5, 1., 10.)
    assert l[2][2:4] == (1.5, 1.)
    # opacity != 1 adds the leading pass-through transmission; all five lobes
    l = pa.lobes_from_params('uber', Kd=[.5] * 3, Ks=[.5] * 3, Kr=[.5] * 3, Kt=[.5] * 3, roughness=1e-5,
                             opacity=[.5] * 3, eta=1.3)
    assert [x[0] for x in l] == [A.PT_LOBE_SPEC_TRANSMIT, A.PT_LOBE_LAMBERT, A.PT_LOBE_MICROFACET_BLINN,
                                 A.PT_LOBE_SPEC_REFLECT, A.PT_LOBE_SPEC_TRANSMIT]
    assert l[2][4] == 10000.                      # Blinn exponent clip (torrance.py:66-67)
    np.testing.assert_allclose(l[1][1], .25)


def test_matte_lambert_vs_oren_nayar():
    assert pa.lobes_from_params('matte', Kd=[.1] * 3, sigma=0.)[0][0] == A.PT_LOBE_LAMBERT
    on = pa.lobes_from_params('matte', Kd=[.1] * 3, sigma=20.)[0]
    sig = np.deg2rad(20.) ** 2
    assert on[0] == A.PT_LOBE_OREN_NAYAR
    assert on[2] == pytest.approx(1. - sig / (2. * (sig + .33))) and on[3] == pytest.approx(.45 * sig / (sig + .09))


def test_flat_scene_rejects_bad_tree(golden_dir):
    d = dict(np.load(os.path.join(golden_dir, 'bvh_soup300.npz')))
    meta = d['node_meta'].copy()
    meta[0, 0] = 10 ** 6
    d2 = dict(d, node_meta=meta)
    with pytest.raises(ValueError):
        pa.flat_scene_from_npz(d2)


def test_host_camera_and_film_match_reference_fixture(golden_dir):
    """pytracer_amd.host restates PerspectiveCamera/ImageFilm construction; the fixture holds the
    matrices and the Lanczos table the reference built for the 128x128 head-mesh camera."""
    from pytracer_amd import host
    d = dict(np.load(os.path.join(golden_dir, 'head_mesh_scene.npz')))
    cam, film = host.head_mesh_camera_film(128, 128)
    np.testing.assert_allclose(np.array(cam.r2c).reshape(4, 4), d['cam_r2c'], rtol=1e-13, atol=1e-18)
    np.testing.assert_allclose(np.array(cam.c2w).reshape(4, 4), d['cam_c2w'], rtol=0, atol=0)
    np.testing.assert_allclose(np.array(cam.dx), d['cam_dx'], rtol=1e-12, atol=1e-20)
    np.testing.assert_allclose(np.array(cam.dy), d['cam_dy'], rtol=1e-12, atol=1e-20)
    np.testing.assert_allclose(np.array(film.table).reshape(16, 16), d['filt_table'], rtol=1e-14, atol=1e-18)
    assert (film.x_start, film.x_cnt, film.y_start, film.y_cnt) == tuple(int(v) for v in d['film_win'])
