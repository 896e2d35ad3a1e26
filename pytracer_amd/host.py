"""Host-side constructors for the descriptors of the C ABI, restating the reference's own
constructors so that a workload can be described without the reference being importable
(the GPU box has no /root/reference).  Each function cites what it follows."""
import numpy as np

from . import flatten as F

EPS = 1e-5
FILTER_TABLE_SIZE = 16     # pytracer/__init__.py


def _scale(x, y, z):
    return np.diag([x, y, z, 1.]).astype(np.float64)


def _translate(x, y, z):
    m = np.eye(4)
    m[0, 3], m[1, 3], m[2, 3] = x, y, z
    return m


def perspective(fov, n, f):
    """Transform.perspective, transform/__init__.py:239-250 (returns the 4x4 matrix)."""
    m = np.eye(4)
    m[2, 2] = f / (f - n)
    m[2, 3] = -f * n / (f - n)
    m[3, 2] = 1.
    m[3, 3] = 0.
    tan_inv = 1. / np.tan(np.deg2rad(fov) / 2.)
    return _scale(tan_inv, tan_inv, 1.).dot(m)


def _xf_point(m, p):
    """Transform.__call__ on a Point, transform/__init__.py:55-60."""
    res = m[0:4, 0:3].dot(p) + m[0:4, 3]
    out = res[0:3].copy()
    if res[3] < 1. - EPS or res[3] > 1. + EPS:
        out /= res[3]
    return out


def perspective_camera(c2w, scr_win, fov, x_res, y_res, lens_rad=0., focal_dist=1e100):
    """PerspectiveCamera.__init__ (camera/projective.py:137-143) on top of
    ProjectiveCamera.__init__ (:42-59).  c2w: 4x4 camera-to-world matrix.  Returns pt_camera."""
    c2s = perspective(fov, .001, 1000.)
    s2r = _scale(x_res, y_res, 1.).dot(_scale(1. / (scr_win[1] - scr_win[0]), 1. / (scr_win[2] - scr_win[3]), 1.)) \
        .dot(_translate(-scr_win[0], -scr_win[3], 0.))
    r2s = np.linalg.inv(s2r)
    # Transform.__mul__ multiplies the matrices (transform/__init__.py:98-101); c2s.inverse() is the
    # numerically inverted perspective matrix (Transform(m) → np.linalg.inv, :25-29, times scale^-1)
    tan_inv = 1. / np.tan(np.deg2rad(fov) / 2.)
    m = np.eye(4)
    m[2, 2] = 1000. / (1000. - .001)
    m[2, 3] = -1000. * .001 / (1000. - .001)
    m[3, 2] = 1.
    m[3, 3] = 0.
    c2s_inv = np.linalg.inv(m).dot(_scale(1. / tan_inv, 1. / tan_inv, 1.))
    del c2s
    r2c = c2s_inv.dot(r2s)
    o = _xf_point(r2c, np.zeros(3))
    dx = _xf_point(r2c, np.array([1., 0., 0.])) - o
    dy = _xf_point(r2c, np.array([0., 1., 0.])) - o
    return F.camera_desc(r2c, np.asarray(c2w, np.float64), dx, dy, lens_rad, focal_dist)


def lanczos_sinc(x, tau):
    """LanczosSincFilter.sinc, filter/filter.py:100-118 (np.sinc is the normalised sinc)."""
    x = abs(x)
    if x < EPS:
        return 1.
    if x > 1.:
        return 0.
    return float(np.sinc(x * tau) * np.sinc(x))


def filter_table(kind, xw, yw, **kw):
    """ImageFilm.__init__ filter table, film/film.py:127-135, for the filters of filter/filter.py."""
    if kind == 'lanczos':
        tau = kw['tau']
        f = lambda x, y: lanczos_sinc(x * (1. / xw), tau) * lanczos_sinc(y * (1. / yw), tau)
    elif kind == 'box':
        f = lambda x, y: 1.
    elif kind == 'triangle':
        f = lambda x, y: max(0., xw - abs(x)) * max(0., yw - abs(y))
    elif kind == 'gaussian':
        a = kw['alpha']
        ex, ey = np.exp(-a * xw * xw), np.exp(-a * yw * yw)
        f = lambda x, y: max(0., np.exp(-a * x * x) - ex) * max(0., np.exp(-a * y * y) - ey)
    else:
        raise NotImplementedError(kind)
    t = np.empty((FILTER_TABLE_SIZE, FILTER_TABLE_SIZE))
    dx, dy = xw / FILTER_TABLE_SIZE, yw / FILTER_TABLE_SIZE
    for y in range(FILTER_TABLE_SIZE):
        fy = (y + .5) * dy
        for x in range(FILTER_TABLE_SIZE):
            t[y, x] = f((x + .5) * dx, fy)
    return t


def image_film(x_res, y_res, kind='lanczos', xw=4., yw=4., crop=(0., 1., 0., 1.), **kw):
    """ImageFilm.__init__ extents (film/film.py:115-119, including its use of xr for the y start)."""
    x_start = int(np.ceil(x_res * crop[0]))
    x_cnt = max(1, int(np.ceil(x_res * crop[1])) - x_start)
    y_start = int(np.ceil(x_res * crop[2]))
    y_cnt = max(1, int(np.ceil(y_res * crop[3])) - y_start)
    return F.film_desc(x_res, y_res, (x_start, x_cnt, y_start, y_cnt), xw, yw, filter_table(kind, xw, yw, **kw))


def head_mesh_camera_film(x_res, y_res):
    """examples/head_mesh.py:69-75 configure_camera: Lanczos(4,4,3) film, camera translated to
    (-13,-1,-35), screen window [0,1]^2, fov 90."""
    c2w = _translate(-13., -1., -35.)
    cam = perspective_camera(c2w, [0., 1., 0., 1.], 90., x_res, y_res, 0., 1e100)
    film = image_film(x_res, y_res, 'lanczos', 4., 4., tau=3.)
    return cam, film
