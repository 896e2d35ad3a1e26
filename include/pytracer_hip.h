/* pytracer_hip.h — C ABI of the MI355X (gfx950) back end for pyTracer's
 * ray–scene intersection and light-integration loop.
 *
 * The reference (zjiayao/pyTracer) is pure Python and has no FFI layer; its seams are
 * duck-typed methods.  Each entry point below names the reference interface it replaces
 * (file:line relative to the reference root).  Plain C types only: no torch, no C++.
 *
 * Conventions
 *   - every function returns 0 on success, a negative pt_status otherwise; the message
 *     is available from pt_last_error(ctx).  No exceptions cross the boundary.
 *   - "host" pointers are read during the call and copied; "device" pointers must be
 *     device memory on the context's GPU (e.g. torch tensor.data_ptr()) and stay owned
 *     by the caller.
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream).  Calls on
 *     one ctx are serialised by the caller; distinct ctxs are independent.
 *   - all reals are IEEE binary64 ("f64"): the reference computes in numpy.float64
 *     (pytracer/__init__.py:18) and hit/miss + primitive ids must match it bit-exactly.
 */
#ifndef PYTRACER_HIP_H
#define PYTRACER_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PT_ABI_VERSION 12

typedef enum {
    PT_OK = 0,
    PT_ERR_ARG = -1,        /* bad argument (null pointer, size, unsupported enum) */
    PT_ERR_HIP = -2,        /* a HIP runtime call failed */
    PT_ERR_NO_SCENE = -3,   /* pt_intersect / pt_render before pt_scene_upload */
    PT_ERR_DEPTH = -4,      /* BVH deeper than the 64-entry stack of bvh.py:332 */
    PT_ERR_CAPACITY = -5    /* request exceeds a documented fixed capacity */
} pt_status;

/* ---- BSDF lobes: flattened result of Material.get_bsdf with ConstantTexture inputs
 * (material/material.py:122-141 Matte, :166-189 Plastic, :208-222 Mirror, :280-299 Glass,
 *  :329-364 Uber).  Lobes are stored in push_back order. ---------------------------- */
typedef enum {
    PT_LOBE_LAMBERT = 0,          /* reflection/bdf/bdf.py:310-338; R                       */
    PT_LOBE_OREN_NAYAR = 1,       /* reflection/bdf/orennayar.py:36-62; R, a=A, b=B         */
    PT_LOBE_MICROFACET_BLINN = 2, /* reflection/bdf/torrance.py:187-228 + Blinn :52-102 +
                                     FresnelDielectric(a=eta_i, b=eta_t); c = exponent      */
    PT_LOBE_SPEC_REFLECT = 3,     /* bdf.py:229-256; R; FresnelDielectric(a,b); c!=0 →
                                     FresnelFullReflect (fresnel.py:75-83)                  */
    PT_LOBE_SPEC_TRANSMIT = 4,    /* bdf.py:259-307; R=T, a=ei, b=et                        */
    PT_LOBE_MICROFACET_BLINN_COND = 5,/* the same Microfacet/Blinn with FresnelConductor(eta[3],
                                     k[3]) (fresnel.py:30-41, reflection/utility.py:60-73):
                                     MetalMaterial, material.py:242-258; c = exponent        */
    PT_LOBE_MEASURED_IRISO = 6    /* IrIsotropicBRDF of MeasuredMaterial (reflection/bdf/measured.py:39-66,
                                     material.py:403-524): REFLECTION|GLOSSY, f = Gaussian-weighted mean of
                                     the measured samples inside a ball around brdf_remap(wo, wi)
                                     (reflection/utility.py:76-90) whose radius grows from .01 by 1.414
                                     until it holds more than 2 samples or exceeds 1.5; sampled and
                                     weighted like every base BDF (bdf.py:97-126).  a = first sample,
                                     b = number of samples (integers) in pt_scene_desc.brdf_p / brdf_v;
                                     R, c, eta, k unused                                     */
} pt_lobe_type;

#define PT_MAX_LOBES 5
#define PT_LOBE_DROP_IF_BLACK 0x10000
/* bits 20..30 of pt_lobe.tex: k > 0 = the lobe's FLOAT parameter is texture k-1 evaluated at the hit (its first
 * component), as the materials evaluate their float textures per hit:
 *   PT_LOBE_OREN_NAYAR        sigma of MatteMaterial (material.py:132-139): clipped to [0, 90]; 0 makes the lobe a
 *                             Lambertian at that hit, else a, b = A, B of OrenNayar.__init__ (orennayar.py:36-42)
 *   PT_LOBE_MICROFACET_BLINN  roughness of Plastic / Uber (material.py:178-179, 349-350): c = min(1 / roughness, 10000)
 * (no other lobe type may carry one) */
#define PT_LOBE_PARAM_TEX_SHIFT 20
#define PT_LOBE_PARAM_TEX(tex) (((tex) >> PT_LOBE_PARAM_TEX_SHIFT) & 0x7FF)

typedef struct {
    int32_t type;       /* pt_lobe_type */
    int32_t tex;        /* bits 20..30: PT_LOBE_PARAM_TEX above.  Low 16 bits k > 0: colour = R * clip(texture k-1 at the
                           hit) (pt_scene_desc.textures; Material.get_bsdf evaluates its textures per hit,
                           material/material.py:131,175-176).  PT_LOBE_DROP_IF_BLACK set: the lobe exists at a
                           hit only if that colour is not black (`if not k.is_black()` of Uber / Mirror /
                           Glass, material.py:217-219,289-295,343-362; is_black = every component < 1e-5) */
    double R[3];        /* RGB scale (Spectrum = RGBSpectrum, spectral/spectrum.py:610) */
    double a, b, c;
    double eta[3], k[3];/* conductor lobes only */
} pt_lobe;

typedef struct {
    int32_t n_lobes;
    int32_t _pad;
    pt_lobe lobes[PT_MAX_LOBES];
} pt_material;

/* ---- spectrum textures of a constant-free kind (texture/texture/__init__.py) over constant
 * children, with their mappings (texture/texturemap/__init__.py).  Materials hand textures a COPY of
 * the shading geometry whose screen-space differentials are zero (material.py:128,172: dg_s.copy(),
 * geometry/diffgeom.py:21-42), so the closed-form antialiasing of the checkerboard always takes its
 * "filter inside one check" branch and equals point sampling (texture/__init__.py:156-166). */
typedef enum {
    PT_TEX_CHECKER2D = 0,   /* Chekcerboard2DTexture :119-179, aa 'none' or 'closedform'; v[0], v[1]   */
    PT_TEX_CHECKER3D = 1,   /* Checkboard3DTexture :182-196; v[0], v[1]                                */
    PT_TEX_BILERP = 2,      /* BilerpTexture :85-104; v[0..3] = v00, v01, v10, v11                      */
    PT_TEX_UV = 3,          /* UVTexture :107-116: (s - floor s, t - floor t, 0)                        */
    PT_TEX_WRINKLED = 4     /* WrinkledTexture :252-266 (a FLOAT texture: all three components equal):
                               turbulence(p, 0, 0, omega, octaves) of texture/utility.py:78-101 over Perlin
                               noise() :121-164 with the table of data/noise.py, as they execute (quirks in
                               DESIGN.md); IdentityMapping3D only; v[0][0] = omega ("roughness"),
                               v[0][1] = octaves (0..64), v[1][0] = the constant of a ScaleTexture(constant,
                               WrinkledTexture) around it (:49-61), else 1.  The reference's other noise textures (FBm, Windy,
                               Marble, Dots) raise when evaluated and are not offered                      */
} pt_tex_kind;
typedef enum {
    PT_MAP_UV = 0,          /* UVMapping2D texturemap:38-52; m = su, sv, du, dv                         */
    PT_MAP_PLANAR = 1,      /* PlannarMapping2D :135-149; m = vs[3], vt[3], ds, dt                      */
    PT_MAP_SPHERICAL = 2,   /* SphericalMapping2D :55-68; m = w2t 4x4 row-major                         */
    PT_MAP_IDENTITY3D = 3   /* IdentityMapping3D :175-180 (3D textures); m = w2t 4x4 row-major          */
} pt_map_kind;
typedef struct {
    int32_t kind;       /* pt_tex_kind */
    int32_t mapping;    /* pt_map_kind */
    double m[16];
    double v[4][3];
} pt_texture;

/* ---- lights (light/light.py) ------------------------------------------------------ */
typedef enum {
    PT_LIGHT_DIFFUSE_SPHERE = 0,  /* DiffuseAreaLight over one full Sphere (light.py:484-547,
                                     shape/sphere.py:78-123,125-224)                        */
    PT_LIGHT_SPOT = 1,            /* SpotLight (light.py:155-222)                           */
    PT_LIGHT_POINT = 2            /* PointLight (light.py:118-152)                          */
} pt_light_kind;

typedef struct {
    int32_t kind;       /* pt_light_kind */
    int32_t ns;         /* Light.ns (light.py:38) */
    int32_t ro;         /* sphere: Shape.ro (reverse orientation) */
    int32_t _pad;
    double emit[3];     /* DiffuseAreaLight.emit | SpotLight/PointLight.intensity */
    double pos[3];      /* sphere: o2w(0,0,0) | spot/point: Light.pos */
    double radius;      /* sphere radius */
    double w2o[12];     /* rows 0..2 of the 4x4: sphere world→object | spot world→light */
    double o2w[12];     /* rows 0..2 of the inverse */
    double cos_width;   /* spot */
    double cos_falloff; /* spot */
} pt_light;

/* ---- scene: linearised BVH (aggregate/accelerator/bvh.py:60-66,277-293) + triangles
 * in BVH.primitives order (shape/triangle.py:426-435 world-space mesh.p) ------------- */
typedef struct {
    int32_t n_nodes, n_tris, n_materials, n_lights;
    const double *node_bounds;   /* host [n_nodes][6]: pMin.xyz, pMax.xyz */
    const int32_t *node_meta;    /* host [n_nodes][3]: offset, n_prim, axis */
    const double *tri_verts;     /* host [n_tris][9]: p1,p2,p3 */
    const int32_t *tri_mat;      /* host [n_tris]: material index */
    const int32_t *tri_flip;     /* host [n_tris]: shape.ro ^ transform_swaps_handedness */
    const pt_material *materials;/* host [n_materials] */
    const pt_light *lights;      /* host [n_lights] */
    int32_t n_textures, _pad;
    const pt_texture *textures;  /* host [n_textures]; referenced by pt_lobe.tex */
    const double *tri_uv;        /* host [n_tris][6] or NULL: (u,v) of the triangle's three corners as
                                    Triangle.get_uvs returns them (shape/triangle.py:125-131; NULL = the
                                    default (0,0),(1,0),(1,1) for every triangle) */
    int64_t n_brdf_samples;      /* measured BRDF samples of all MeasuredMaterials, table after table */
    const double *brdf_p;        /* host [n_brdf_samples][3]: IrIsotropicBRDFSample.p = brdf_remap(wo, wi) of
                                    the measurement (material.py:471-476) = the cKDTree's data */
    const double *brdf_v;        /* host [n_brdf_samples][3]: IrIsotropicBRDFSample.v, the RGB the reference
                                    made of the measured spectrum (Spectrum.from_sampled, material.py:473) */
} pt_scene_desc;

/* ---- camera (camera/projective.py:42-59; OrthoCamera :65-127; PerspectiveCamera
 * :137-143,178-232).  EnvironmentCamera is not offered: the reference cannot render with it
 * (its ray differentials make compute_differential produce NaN). ------------- */
typedef enum { PT_CAM_PERSPECTIVE = 0, PT_CAM_ORTHO = 1 } pt_cam_kind;
typedef struct {
    double r2c[16];     /* raster→camera, row-major 4x4 (ProjectiveCamera.r2c.m) */
    double c2w[16];     /* camera→world (c2w.startTransform.m) */
    double dx[3], dy[3];/* dxCam, dyCam */
    double lens_rad, focal_dist;
    int32_t kind;       /* pt_cam_kind */
    int32_t _pad;
} pt_camera;

/* ---- film (film/film.py:108-135) ---------------------------------------------------- */
typedef struct {
    int32_t x_res, y_res;
    int32_t x_start, x_cnt, y_start, y_cnt;   /* xPixel_start/cnt, yPixel_start/cnt */
    double xw, yw;                            /* filter half-widths */
    double table[256];                        /* filter_table[16][16] (film.py:129-135) */
} pt_film_desc;

/* ---- sampler (sampler/sampler/stratified.py:34-89) ---------------------------------- */
typedef struct {
    int32_t x_start, x_end, y_start, y_end;   /* pixel window of the StratifiedSampler */
    int32_t xs, ys;                           /* strata per pixel; spp = xs*ys */
    int32_t jitter;
    uint32_t seed;                            /* production stream only (table == NULL) */
} pt_sampler_desc;

typedef enum { PT_INTEG_DIRECT = 0, PT_INTEG_PATH = 1, PT_INTEG_WHITTED = 2 } pt_integ_kind;

/* integrator (integrator/surface/direct.py:36-105, path.py:27-144, whitted.py:16-59) */
typedef struct {
    int32_t kind;        /* pt_integ_kind */
    int32_t max_depth;
    int32_t pretest;     /* 1: trace the camera any-hit pre-test of renderer.py:168 as a
                            separate ray (its result equals the closest-hit flag);
                            0: skip it (identical image, fewer rays) */
    int32_t shade_f32;   /* 0: every kernel computes in binary64 (the reference's arithmetic; required with a
                            sample table).  1, production streams only: the BSDF / light ARITHMETIC of the shade
                            stage (lobe f / pdf / sample_f, Fresnel, pow, sincos, MIS weights) runs in binary32;
                            points, rays, the sphere-light intersection, every discrete choice made from a
                            sample value, throughput, radiance and the film stay binary64.  north_star's
                            tolerance: radiance within 1e-4 relative of the float64 reference */
} pt_integrator_desc;

/* work shard: pixels [x0,x1)×[y0,y1) of the sampler window, sample indices [s0,s1).
 * y_step > 1 keeps only the rows y0 + y_phase + k*y_step (k = 0, 1, …) below y1: the interleaved-rows shard
 * of a multi-GPU render (rank r of N: y_step = N, y_phase = r) — every rank gets rows from all over the image
 * (balanced load) and every pixel keeps all its samples on one rank (coherent waves, one film footprint pass).
 * y_step = 0 or 1: every row.  Samples are numbered in tile order: kept rows top to bottom, x, then sample index
 * (the order of sample_L).  The reference's decomposition hook is Sampler.compute_subwindow
 * (sampler/sampler/__init__.py:87-111), unused there (renderer/renderer.py:97-109,135). */
typedef struct { int32_t x0, x1, y0, y1, s0, s1; int32_t y_step, y_phase; } pt_tile;

typedef struct {
    uint64_t camera_samples;
    uint64_t rays_closest;
    uint64_t rays_any;
    uint64_t node_visits;     /* slab tests executed (BVH._intersect_p calls) */
    uint64_t tri_tests;       /* Triangle.intersect/_p calls */
    uint64_t guard_nan, guard_neg, guard_inf;   /* renderer.py:185-193 */
    uint64_t node_visits_any;  /* the share of node_visits / tri_tests spent on any-hit rays */
    uint64_t tri_tests_any;
    /* measurement only (no reference counterpart): filled when pt_set_profiling(ctx, 1) is on.
     * Kernel time per stage from HIP events recorded on the caller's stream around each launch. */
    uint64_t launches_trace_closest, launches_trace_any;
    double ms_raygen, ms_trace_closest, ms_trace_any, ms_shade, ms_splat;
} pt_stats;

typedef struct pt_ctx pt_ctx;

/* library/ABI version */
int pt_version(void);

/* one context per GPU */
int pt_ctx_create(int device, pt_ctx **out);
int pt_ctx_destroy(pt_ctx *ctx);
const char *pt_last_error(const pt_ctx *ctx);

/* replaces: Scene(aggregate, lights, vr) holding a BVH (scene/scene.py:21-29) */
int pt_scene_upload(pt_ctx *ctx, const pt_scene_desc *scene);

/* replaces: Scene.intersect / Scene.intersect_p on a batch (scene/scene.py:35-39 →
 * BVH.intersect bvh.py:322-365, BVH.intersect_p bvh.py:367-410, Triangle.intersect(_p)
 * triangle.py:173-315).
 *   rays   device f64 [8][n]: ox,oy,oz,dx,dy,dz,mint,maxt (component-major SoA)
 *   any_hit=0: prim[n] = index into BVH.primitives or -1; t,b1,b2 [n] (t = ray.maxt after)
 *   any_hit=1: prim[n] = 1 if occluded else 0; t,b1,b2 may be NULL
 *   stats  host, may be NULL: node_visits / tri_tests / rays_* are ADDED to it */
int pt_intersect(pt_ctx *ctx, const double *rays, int64_t n, int any_hit,
                 int32_t *prim, double *t, double *b1, double *b2,
                 pt_stats *stats, void *stream);

/* No counterpart in the reference.  pt_intersect for callers that cannot allocate device memory: code under
 * HIP graph capture, tight serving loops, harnesses that forbid hipMalloc.  Same arguments, same results
 * and the same census as pt_intersect (it launches the same kernels), but the call allocates and frees
 * nothing: the census counters of the launch — the one scratch buffer the trace kernels use beyond their
 * inputs and outputs — live in a file-scope `static __device__` array of the library (20 KB per GPU the
 * library has a context on, held until the process ends) instead of the context's hipMalloc'ed block, and no
 * profiling events are created (pt_set_profiling is ignored).
 *   n      at most PT_INTERSECT_STATIC_MAX_RAYS per call; more → PT_ERR_CAPACITY, nothing launched
 *   stats  NULL: the call enqueues a memset of the scratch and the trace kernel on `stream` and returns
 *          without synchronising (capturable).  Non-NULL: it then copies the 20 KB of counters to the host
 *          stack and synchronises `stream`, like pt_intersect.
 * All calls on one GPU share that GPU's array, which keeps the last call's sums (every call clears it first).
 * Overlap: like every entry point, calls on ONE context must be serialised by the caller — one stream at a
 * time, and not concurrently with pt_render / pt_intersect on that context: the traversal kernels of large
 * trees keep their todo stacks in a per-context spill area indexed by grid thread, so two launches of one
 * context in flight together would corrupt each other's walks (wrong hit records, not only a wrong census).
 * Calls on DIFFERENT contexts of the same GPU may overlap: each context has its own spill area, so the hit
 * records are correct, and only the census — the one thing in the shared array — races; pass stats == NULL on
 * all but one of them, or serialise them. */
#define PT_INTERSECT_STATIC_MAX_RAYS 0x40000000LL
int pt_intersect_static(pt_ctx *ctx, const double *rays, int64_t n, int any_hit,
                        int32_t *prim, double *t, double *b1, double *b2,
                        pt_stats *stats, void *stream);

/* replaces: the body of SamplerRendererTask.__call__ (renderer/renderer.py:132-198):
 * sample generation → camera ray → SamplerRenderer.li → guards → ImageFilm.add_sample.
 *   sample_table  device f64 [n_window_samples][dims] or NULL.  Non-NULL = parity mode:
 *                 row r holds (imageX,imageY,lens_u,lens_v,time, oneD..., twoD...) of the
 *                 r-th sample in reference order (y, x, s) over the sampler window, and
 *                 rng() draws are the counter hash u(r,k) documented in DESIGN.md.
 *   film_accum    device f64 [y_cnt][x_cnt][4] = Lxyz[3], weight_sum; ADDED to.
 *   sample_L      device f64 [n_tile_samples][3] or NULL: per-sample radiance after guards.
 *   stats         host, may be NULL: counters are ADDED.
 * Device memory: the tile is rendered in wavefront batches of up to 2^28 paths (~460 bytes of queue and path state
 * each; the environment variable PT_BATCH sets another limit) and never more than fit in HALF of the device memory
 * that is free at the call; the workspace is allocated by the context (hipMalloc, at the first call and when the
 * batch size changes), kept until pt_ctx_destroy, and kept as it is by later calls that are bounded by memory.  If
 * hipMalloc refuses it the batch is halved (down to 2^20 paths) before the call fails with PT_ERR_HIP. */
int pt_render(pt_ctx *ctx, const pt_camera *cam, const pt_film_desc *film,
              const pt_sampler_desc *sampler, const pt_integrator_desc *integ,
              const pt_tile *tile, const double *sample_table, int32_t table_dims,
              double *film_accum, double *sample_L, pt_stats *stats, void *stream);

/* No counterpart in the reference (it has no timers): measurement hook for
 * bench.py.  enable = 1: pt_render / pt_intersect bracket every kernel launch with HIP events on the
 * stream it is launched on and add the per-stage totals to pt_stats (requires stats != NULL); every
 * trace launch counts its rays, node visits and triangle tests into a census block of its own, and
 * the launches of the call are kept for pt_launch_log.  enable = 2: the same, but only the trace
 * launches are bracketed with events (a quarter of the event records).  0: off. */
int pt_set_profiling(pt_ctx *ctx, int enable);

/* No counterpart in the reference.  The launches of the last pt_render / pt_intersect that ran with
 * profiling on and stats != NULL, in issue order: which kernel traced which level of which wavefront
 * batch, what it counted and how long it took — so that each traversal kernel symbol can be priced
 * against the roofline by itself (bench.py roofline.per_launch).
 *   out    host [cap]; n_out receives the number of launches of that call (may exceed cap: only cap
 *          records are written) */
typedef enum { PT_STAGE_RAYGEN = 0, PT_STAGE_TRACE_CLOSEST = 1, PT_STAGE_TRACE_ANY = 2, PT_STAGE_SHADE = 3,
               PT_STAGE_SPLAT = 4 } pt_stage;
typedef enum { PT_KERNEL_NONE = 0, PT_KERNEL_TRACE = 1, PT_KERNEL_TRACE_META = 2, PT_KERNEL_TRACE_TOP = 3,
               PT_KERNEL_TRACE_PAIR = 4, PT_KERNEL_TRACE_PAIR_REFILL = 5, PT_KERNEL_TRACE_PACKET = 6 } pt_kernel_id;
typedef struct {
    int32_t stage;      /* pt_stage; PT_STAGE_SHADE = k_list_hits + k_shade of one level */
    int32_t kernel;     /* pt_kernel_id of a trace launch (the fetch form of the walk), else 0 */
    int32_t level;      /* bounce level of the launch; -1: ray generation, camera pre-test, splat */
    int32_t batch;      /* wavefront batch of the render (0 for pt_intersect) */
    uint64_t rays, node_visits, tri_tests;   /* trace launches: the launch's own census (all 0 if the
                                                call had more trace launches than census blocks, 256) */
    double ms;          /* HIP events around the launch; -1 when this profiling level does not time it */
} pt_launch_record;
int pt_launch_log(pt_ctx *ctx, pt_launch_record *out, int32_t cap, int32_t *n_out);
const char *pt_kernel_name(int32_t id);      /* "k_trace_meta", ... ("" for PT_KERNEL_NONE) */

/* replaces: the pixel loop of ImageFilm.write_image (film/film.py:237-256):
 * rgb[y][x] = max(0, xyz2rgb(Lxyz)/weight_sum).  film_accum device [y][x][4] → rgb device
 * [y][x][3]. */
int pt_film_resolve(pt_ctx *ctx, const double *film_accum, int32_t x_cnt, int32_t y_cnt,
                    double *rgb, void *stream);

/* replaces: the array conversion at the top of imageio.write_image (utility/imageio.py:60-65):
 * per channel c, coef = max(rgb[:,:,c]); if coef != 0: rgb[:,:,c] = (255 * rgb[:,:,c]) / coef;
 * then rgb.astype('uint8') (truncation; the same binary64 operations, so the bytes are those numpy
 * produces from the same rgb).  rgb device [y][x][3] f64 (the output of pt_film_resolve; not
 * modified) → out device [y][x][3] u8, ready for PIL.Image.fromarray(..., 'RGB').  Calls on one
 * context are ordered by their stream; two streams must not run this concurrently on one context
 * (the three channel maxima live in the context). */
int pt_film_to_u8(pt_ctx *ctx, const double *rgb, int32_t x_cnt, int32_t y_cnt, uint8_t *out,
                  void *stream);

/* replaces: BVH.__init__ → _recursive_build → _flatten_tree (aggregate/accelerator/bvh.py:
 * 68-107,152-275,277-293, with _partition :110-120, _insertion_sort :123-130,
 * _bucket_partition :133-150).  Host-side, needs no GPU and no context.  The tree, the bounds
 * and the primitive order are those of the Python build (SAH with 12 buckets, traversal cost
 * .125; n_prim <= 4 → insertion sort + median; the partition loops that never classify element
 * k; the mid==start/end fix-ups), so primitive ids and traversal order match the reference.
 *   prim_bounds  host f64 [n][6] = Primitive.world_bound() pMin[3], pMax[3] of every refined
 *                primitive, in the order of BVH.primitives BEFORE the build (bvh.py:70-73,91)
 *   method       PT_BVH_SAH ('sah') or PT_BVH_MIDDLE ('middle'), bvh.py:75-83
 *   node_bounds  host f64 [2n-1][6], node_meta host i32 [2n-1][3] = (offset, n_prim, axis) as in
 *                pt_scene_desc; prim_order host i32 [n]: ordered_prims[k] = primitives[prim_order[k]]
 *   n_nodes      out: nodes written (0 for n == 0: the reference leaves BVH.nodes = None)
 *   max_depth    out, may be NULL: depth of the tree (pt_scene_upload rejects > 64)
 * Returns PT_ERR_ARG where the reference itself fails (its asserts at bvh.py:227,267-268; the
 * TypeError of the unreachable-by-default leaf branch :261-265); pt_build_last_error() has the
 * text (thread-local; valid until the next pt_bvh_build on the same thread). */
typedef enum { PT_BVH_SAH = 0, PT_BVH_MIDDLE = 1 } pt_bvh_method;
int pt_bvh_build(const double *prim_bounds, int64_t n, int32_t max_prim_per_node, int32_t method,
                 double *node_bounds, int32_t *node_meta, int32_t *prim_order, int64_t *n_nodes,
                 int32_t *max_depth);
const char *pt_build_last_error(void);

/* replaces: LoopSubdiv.__init__ + LoopSubdiv.refine (shape/loopsubdiv.py:195-247,293-446, created by
 * create_loop_subdiv :19-44).  Host-side, needs no GPU and no context.  Same vertex and face order
 * and the same floating-point summation order as the reference, so the result equals its
 * TriangleMesh bit for bit (vertex normals, which never reach shading — triangle.py:325 — are not
 * produced).
 *   P         host f64 [nv][3] object-space control vertices, vi host i32 [nf][3]
 *   n_levels  create_loop_subdiv's 'nlevels' (default 3 there)
 *   P_out     host f64 [p_cap][3] limit-surface vertices (object space; the caller applies o2w like
 *             create_triangle_mesh, triangle.py:426-435), idx_out host i32 [idx_cap]
 *   nv_out/nf_out  vertices / triangles produced (nf * 4^n_levels triangles, at most
 *             nv + nf * (4^n_levels - 1) vertices); PT_ERR_CAPACITY if the buffers are smaller
 * Returns PT_ERR_ARG where the reference itself fails: inconsistent topology (its RuntimeError in
 * vnum), unused vertices, and any mesh whose vertex 0 is interior — refine() raises
 * UnboundLocalError in its tangent pass then (:409), which includes every closed mesh.
 * pt_subdiv_last_error() has the text (thread-local). */
int pt_loop_subdiv(const double *P, int64_t nv, const int32_t *vi, int64_t nf, int32_t n_levels,
                   double *P_out, int64_t p_cap, int32_t *idx_out, int64_t idx_cap,
                   int64_t *nv_out, int64_t *nf_out);
const char *pt_subdiv_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* PYTRACER_HIP_H */
