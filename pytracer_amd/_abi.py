"""ctypes mirror of include/pytracer_hip.h (struct layouts and enums only)."""
import ctypes as C

PT_ABI_VERSION = 12
PT_INTERSECT_STATIC_MAX_RAYS = 0x40000000
PT_MAX_LOBES = 5

PT_OK, PT_ERR_ARG, PT_ERR_HIP, PT_ERR_NO_SCENE, PT_ERR_DEPTH, PT_ERR_CAPACITY = 0, -1, -2, -3, -4, -5

(PT_LOBE_LAMBERT, PT_LOBE_OREN_NAYAR, PT_LOBE_MICROFACET_BLINN, PT_LOBE_SPEC_REFLECT, PT_LOBE_SPEC_TRANSMIT,
 PT_LOBE_MICROFACET_BLINN_COND, PT_LOBE_MEASURED_IRISO) = range(7)
PT_LIGHT_DIFFUSE_SPHERE, PT_LIGHT_SPOT, PT_LIGHT_POINT = range(3)
PT_LOBE_DROP_IF_BLACK = 0x10000
PT_TEX_CHECKER2D, PT_TEX_CHECKER3D, PT_TEX_BILERP, PT_TEX_UV, PT_TEX_WRINKLED = range(5)
PT_LOBE_PARAM_TEX_SHIFT = 20
PT_MAP_UV, PT_MAP_PLANAR, PT_MAP_SPHERICAL, PT_MAP_IDENTITY3D = range(4)
PT_INTEG_DIRECT, PT_INTEG_PATH, PT_INTEG_WHITTED = 0, 1, 2
PT_CAM_PERSPECTIVE, PT_CAM_ORTHO = 0, 1
PT_BVH_SAH, PT_BVH_MIDDLE = 0, 1


class pt_lobe(C.Structure):
    _fields_ = [('type', C.c_int32), ('tex', C.c_int32), ('R', C.c_double * 3),
                ('a', C.c_double), ('b', C.c_double), ('c', C.c_double),
                ('eta', C.c_double * 3), ('k', C.c_double * 3)]


class pt_material(C.Structure):
    _fields_ = [('n_lobes', C.c_int32), ('_pad', C.c_int32), ('lobes', pt_lobe * PT_MAX_LOBES)]


class pt_texture(C.Structure):
    _fields_ = [('kind', C.c_int32), ('mapping', C.c_int32), ('m', C.c_double * 16), ('v', (C.c_double * 3) * 4)]


class pt_light(C.Structure):
    _fields_ = [('kind', C.c_int32), ('ns', C.c_int32), ('ro', C.c_int32), ('_pad', C.c_int32),
                ('emit', C.c_double * 3), ('pos', C.c_double * 3), ('radius', C.c_double),
                ('w2o', C.c_double * 12), ('o2w', C.c_double * 12),
                ('cos_width', C.c_double), ('cos_falloff', C.c_double)]


class pt_scene_desc(C.Structure):
    _fields_ = [('n_nodes', C.c_int32), ('n_tris', C.c_int32), ('n_materials', C.c_int32), ('n_lights', C.c_int32),
                ('node_bounds', C.c_void_p), ('node_meta', C.c_void_p), ('tri_verts', C.c_void_p),
                ('tri_mat', C.c_void_p), ('tri_flip', C.c_void_p),
                ('materials', C.POINTER(pt_material)), ('lights', C.POINTER(pt_light)),
                ('n_textures', C.c_int32), ('_pad', C.c_int32), ('textures', C.POINTER(pt_texture)),
                ('tri_uv', C.c_void_p),
                ('n_brdf_samples', C.c_int64), ('brdf_p', C.c_void_p), ('brdf_v', C.c_void_p)]


class pt_camera(C.Structure):
    _fields_ = [('r2c', C.c_double * 16), ('c2w', C.c_double * 16), ('dx', C.c_double * 3), ('dy', C.c_double * 3),
                ('lens_rad', C.c_double), ('focal_dist', C.c_double), ('kind', C.c_int32), ('_pad', C.c_int32)]


class pt_film_desc(C.Structure):
    _fields_ = [('x_res', C.c_int32), ('y_res', C.c_int32), ('x_start', C.c_int32), ('x_cnt', C.c_int32),
                ('y_start', C.c_int32), ('y_cnt', C.c_int32), ('xw', C.c_double), ('yw', C.c_double),
                ('table', C.c_double * 256)]


class pt_sampler_desc(C.Structure):
    _fields_ = [('x_start', C.c_int32), ('x_end', C.c_int32), ('y_start', C.c_int32), ('y_end', C.c_int32),
                ('xs', C.c_int32), ('ys', C.c_int32), ('jitter', C.c_int32), ('seed', C.c_uint32)]


class pt_integrator_desc(C.Structure):
    _fields_ = [('kind', C.c_int32), ('max_depth', C.c_int32), ('pretest', C.c_int32), ('shade_f32', C.c_int32)]


class pt_tile(C.Structure):
    _fields_ = [('x0', C.c_int32), ('x1', C.c_int32), ('y0', C.c_int32), ('y1', C.c_int32),
                ('s0', C.c_int32), ('s1', C.c_int32), ('y_step', C.c_int32), ('y_phase', C.c_int32)]

    def rows(self):
        """number of rows the tile keeps (include/pytracer_hip.h: y_step / y_phase)"""
        st = max(int(self.y_step), 1)
        return max(0, (self.y1 - self.y0 - self.y_phase + st - 1) // st)

    def n_samples(self):
        return self.rows() * (self.x1 - self.x0) * (self.s1 - self.s0)


PT_STAGE_RAYGEN, PT_STAGE_TRACE_CLOSEST, PT_STAGE_TRACE_ANY, PT_STAGE_SHADE, PT_STAGE_SPLAT = range(5)
STAGE_NAMES = ('raygen', 'trace_closest', 'trace_any', 'shade', 'splat')


class pt_launch_record(C.Structure):
    _fields_ = [('stage', C.c_int32), ('kernel', C.c_int32), ('level', C.c_int32), ('batch', C.c_int32),
                ('rays', C.c_uint64), ('node_visits', C.c_uint64), ('tri_tests', C.c_uint64), ('ms', C.c_double)]


class pt_stats(C.Structure):
    _fields_ = [('camera_samples', C.c_uint64), ('rays_closest', C.c_uint64), ('rays_any', C.c_uint64),
                ('node_visits', C.c_uint64), ('tri_tests', C.c_uint64),
                ('guard_nan', C.c_uint64), ('guard_neg', C.c_uint64), ('guard_inf', C.c_uint64),
                ('node_visits_any', C.c_uint64), ('tri_tests_any', C.c_uint64),
                ('launches_trace_closest', C.c_uint64), ('launches_trace_any', C.c_uint64),
                ('ms_raygen', C.c_double), ('ms_trace_closest', C.c_double), ('ms_trace_any', C.c_double),
                ('ms_shade', C.c_double), ('ms_splat', C.c_double)]

    COUNTERS = ('camera_samples', 'rays_closest', 'rays_any', 'node_visits', 'tri_tests', 'guard_nan', 'guard_neg',
                'guard_inf', 'node_visits_any', 'tri_tests_any')

    def as_dict(self):
        return {k: int(getattr(self, k)) for k in self.COUNTERS}

    def timing(self):
        return {k: float(getattr(self, k)) for k, _ in self._fields_ if k.startswith('ms_')}
