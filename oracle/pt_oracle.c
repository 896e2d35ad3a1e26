/* pt_oracle.c — TEST INFRASTRUCTURE ONLY.
 *
 * A scalar, float64, single-threaded CPU restatement of pyTracer's hot path
 * (camera ray → BVH closest/any hit → BSDF + light sampling → film splat), written from
 * the reference's algorithm with the reference file:line each function follows.  It is
 * the checker for the HIP library (pytracer_amd/csrc) and the timed `cpu_baseline`
 * ("port") of bench.py.  Nothing in the product path may call into it: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg load liboracle.so.
 *
 * Pinning: checked in tests/test_oracle_golden.py against fixtures produced by running
 * the reference itself in the build container (tools/oracle/gen_fixtures.py →
 * tests/golden/ *.npz): hit flags / primitive ids bit-exact, continuous values 1e-9.
 * The reference's own tests pin none of this path.
 *
 * Arithmetic notes.  The reference is numpy.float64 throughout; cross products and
 * squared lengths are explicit unfused scalar expressions (geometry/__init__.py:79-85),
 * dot products go through ndarray.dot (BLAS ddot, whose FMA use is build dependent), so
 * continuous results can differ from the reference in the last ulp; discrete results are
 * compared exactly.  Built with -ffp-contract=off so every expression below is evaluated
 * exactly as written (the HIP kernels do the same and match this file bit-for-bit on
 * traversal results).  np.roots in Sphere.intersect (sphere.py:146) is replaced by the
 * closed-form quadratic.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/pytracer_hip.h"

#define PT_EPS 1e-5                    /* pytracer/__init__.py: EPS */
#define PT_PI 3.14159265358979323846   /* np.pi */
#define PT_INV_PI (1.0 / PT_PI)
#define PT_INV_2PI (1.0 / (2.0 * PT_PI))
#define PT_INF (1.0 / 0.0)

/* BDFType bits, reflection/bdf/bdf.py:60-70 */
#define BDF_REFLECTION 1u
#define BDF_TRANSMISSION 2u
#define BDF_DIFFUSE 4u
#define BDF_GLOSSY 8u
#define BDF_SPECULAR 16u
#define BDF_ALL 31u

typedef struct { double x, y, z; } v3;

static inline v3 V(double x, double y, double z) { v3 r = {x, y, z}; return r; }
static inline v3 vadd(v3 a, v3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 vsub(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 vmul(v3 a, double s) { return V(a.x * s, a.y * s, a.z * s); }
static inline v3 vdiv(v3 a, double s) { return V(a.x / s, a.y / s, a.z / s); }
static inline v3 vneg(v3 a) { return V(-a.x, -a.y, -a.z); }
static inline v3 vmulv(v3 a, v3 b) { return V(a.x * b.x, a.y * b.y, a.z * b.z); }
/* ndarray.dot on 3-vectors, left-to-right accumulation */
static inline double vdot(v3 a, v3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
/* geometry/__init__.py:79-82 */
static inline v3 vcross(v3 a, v3 b) {
    return V(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
/* geometry/__init__.py:84-88 */
static inline double vsqlen(v3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; }
static inline double vlen(v3 a) { return sqrt(vsqlen(a)); }
/* geometry/utility.py:55-73 */
static inline v3 vnormalize(v3 a) {
    double l = vlen(a);
    if (!(l == 0.0)) return vdiv(a, l);
    return a;
}
static inline double clipd(double x, double lo, double hi) { return x < lo ? lo : (x > hi ? hi : x); }
static inline double max0(double x) { return x > 0.0 ? x : 0.0; }   /* python max(0., x) */

/* ----------------------------------------------------------------------------------
 * counter-based uniform used for the injected rng of the parity fixtures and for the
 * production sample stream (DESIGN.md "sample streams").  splitmix64 finaliser.
 * -------------------------------------------------------------------------------- */
static inline double hash_uniform(uint32_t sid, uint32_t k) {
    uint64_t z = ((uint64_t)sid << 32) | (uint64_t)k;
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
    return (double)(z >> 11) * (1.0 / 9007199254740992.0);
}
double pto_hash_uniform(uint32_t sid, uint32_t k) { return hash_uniform(sid, k); }

/* ----------------------------------------------------------------------------------
 * scene
 * -------------------------------------------------------------------------------- */
typedef struct {
    int32_t n_nodes, n_tris, n_materials, n_lights;
    double *node_bounds;
    int32_t *node_meta;
    double *tri_verts;
    int32_t *tri_mat, *tri_flip;
    pt_material *materials;
    pt_light *lights;
    int32_t n_textures;
    pt_texture *textures;
    double *tri_uv;        /* [n_tris][6] or NULL */
    int64_t n_brdf;
    double *brdf_p, *brdf_v;   /* [n_brdf][3] measured BRDF samples (PT_LOBE_MEASURED_IRISO) */
} oscene;

static void *dupmem(const void *p, size_t n) {
    void *q = malloc(n ? n : 1);
    if (n) memcpy(q, p, n);
    return q;
}

int pto_scene_destroy(void *p);

int pto_scene_create(const pt_scene_desc *d, void **out) {
    if (!d || !out) return PT_ERR_ARG;
    oscene *s = (oscene *)calloc(1, sizeof(oscene));
    s->n_nodes = d->n_nodes; s->n_tris = d->n_tris;
    s->n_materials = d->n_materials; s->n_lights = d->n_lights;
    s->node_bounds = (double *)dupmem(d->node_bounds, sizeof(double) * 6 * d->n_nodes);
    s->node_meta = (int32_t *)dupmem(d->node_meta, sizeof(int32_t) * 3 * d->n_nodes);
    s->tri_verts = (double *)dupmem(d->tri_verts, sizeof(double) * 9 * d->n_tris);
    s->tri_mat = (int32_t *)dupmem(d->tri_mat, sizeof(int32_t) * d->n_tris);
    s->tri_flip = (int32_t *)dupmem(d->tri_flip, sizeof(int32_t) * d->n_tris);
    s->materials = (pt_material *)dupmem(d->materials, sizeof(pt_material) * d->n_materials);
    s->lights = (pt_light *)dupmem(d->lights, sizeof(pt_light) * d->n_lights);
    s->n_textures = d->n_textures;
    s->textures = (pt_texture *)dupmem(d->textures, sizeof(pt_texture) * d->n_textures);
    s->tri_uv = d->tri_uv ? (double *)dupmem(d->tri_uv, sizeof(double) * 6 * d->n_tris) : NULL;
    s->n_brdf = d->n_brdf_samples;
    s->brdf_p = (double *)dupmem(d->brdf_p, sizeof(double) * 3 * (size_t)(d->n_brdf_samples > 0 ? d->n_brdf_samples : 0));
    s->brdf_v = (double *)dupmem(d->brdf_v, sizeof(double) * 3 * (size_t)(d->n_brdf_samples > 0 ? d->n_brdf_samples : 0));
    for (int i = 0; i < d->n_materials; i++)
        for (int k = 0; k < s->materials[i].n_lobes; k++) {
            pt_lobe *l = &s->materials[i].lobes[k];
            if (l->tex < 0 || (l->tex & 0xFFFF) > d->n_textures || PT_LOBE_PARAM_TEX(l->tex) > d->n_textures) {
                pto_scene_destroy(s);
                return PT_ERR_ARG;
            }
            if (l->type == PT_LOBE_MEASURED_IRISO) {
                /* the lobe functions see no scene: this private copy of the lobe carries the addresses of
                 * its table in the fields the ABI leaves unused for this lobe type */
                if (!(l->a >= 0. && l->b >= 0. && l->a + l->b <= (double)s->n_brdf)) {
                    pto_scene_destroy(s);
                    return PT_ERR_ARG;
                }
                const double *pp = s->brdf_p + 3 * (size_t)l->a, *pv = s->brdf_v + 3 * (size_t)l->a;
                memcpy(&l->eta[0], &pp, sizeof(pp));
                memcpy(&l->eta[1], &pv, sizeof(pv));
            }
        }
    *out = s;
    return PT_OK;
}

int pto_scene_destroy(void *p) {
    oscene *s = (oscene *)p;
    if (!s) return PT_OK;
    free(s->node_bounds); free(s->node_meta); free(s->tri_verts); free(s->tri_mat);
    free(s->tri_flip); free(s->materials); free(s->lights); free(s->textures); free(s->tri_uv);
    free(s->brdf_p); free(s->brdf_v); free(s);
    return PT_OK;
}

/* ----------------------------------------------------------------------------------
 * rays and traversal
 * -------------------------------------------------------------------------------- */
typedef struct { v3 o, d; double mint, maxt; } oray;
typedef struct { int32_t prim; double t, b1, b2; } ohit;

/* Triangle.intersect / intersect_p up to the range test, shape/triangle.py:173-207 and
 * :245-279 (identical arithmetic; the differential geometry that follows cannot reject
 * because no alpha texture is present). */
static inline int tri_test(const double *tv, const oray *r, double *t_out, double *b1_out, double *b2_out) {
    v3 p1 = V(tv[0], tv[1], tv[2]), p2 = V(tv[3], tv[4], tv[5]), p3 = V(tv[6], tv[7], tv[8]);
    v3 e1 = vsub(p2, p1);
    v3 e2 = vsub(p3, p1);
    v3 s1 = vcross(r->d, e2);
    double div = vdot(s1, e1);
    if (div == 0.0) return 0;
    double divInv = 1.0 / div;
    v3 d = vsub(r->o, p1);
    double b1 = vdot(d, s1) * divInv;
    if (b1 < 0.0 || b1 > 1.0) return 0;
    v3 s2 = vcross(d, e1);
    double b2 = vdot(r->d, s2) * divInv;
    if (b2 < 0.0 || (b1 + b2) > 1.0) return 0;
    double t = vdot(e2, s2) * divInv;
    if (t < r->mint || t > r->maxt) return 0;
    *t_out = t; *b1_out = b1; *b2_out = b2;
    return 1;
}

/* BVH._intersect_p, bvh.py:295-320.  b = bounds (pMin xyz, pMax xyz). */
static inline int slab_test(const double *b, const oray *r, v3 inv, const int *neg) {
    double tmin = (b[3 * neg[0] + 0] - r->o.x) * inv.x;
    double tmax = (b[3 * (1 - neg[0]) + 0] - r->o.x) * inv.x;
    double tmin_y = (b[3 * neg[1] + 1] - r->o.y) * inv.y;
    double tmax_y = (b[3 * (1 - neg[1]) + 1] - r->o.y) * inv.y;
    if (tmin > tmax_y || tmin_y > tmax) return 0;
    if (tmin_y > tmin) tmin = tmin_y;
    if (tmax_y < tmax) tmax = tmax_y;
    double tmin_z = (b[3 * neg[2] + 2] - r->o.z) * inv.z;
    double tmax_z = (b[3 * (1 - neg[2]) + 2] - r->o.z) * inv.z;
    if (tmin > tmax_z || tmin_z > tmax) return 0;
    if (tmin_z > tmin) tmin = tmin_z;
    if (tmax_z < tmax) tmax = tmax_z;
    return tmin < r->maxt && tmax > r->mint;
}

/* BVH.intersect (any_hit=0), bvh.py:322-365, and BVH.intersect_p (any_hit=1), :367-410.
 * ray->maxt is shrunk on every accepted closest hit (aggregate/primitive.py:107).
 * Returns -2 if the 64-entry todo stack of bvh.py:332 would overflow (IndexError there). */
static int bvh_traverse(const oscene *s, oray *r, int any_hit, ohit *h, uint64_t *nvis, uint64_t *ntri) {
    h->prim = -1; h->t = r->maxt; h->b1 = 0; h->b2 = 0;
    if (s->n_nodes == 0) return 0;
    int hit = 0;
    v3 inv = V(1.0 / r->d.x, 1.0 / r->d.y, 1.0 / r->d.z);
    int neg[3] = {r->d.x < 0.0, r->d.y < 0.0, r->d.z < 0.0};
    int todo[64];
    int todo_idx = 0, node_idx = 0;
    for (;;) {
        const int32_t *m = s->node_meta + 3 * node_idx;
        (*nvis)++;
        if (slab_test(s->node_bounds + 6 * node_idx, r, inv, neg)) {
            if (m[1] > 0) {
                for (int i = 0; i < m[1]; i++) {
                    double t, b1, b2;
                    (*ntri)++;
                    if (tri_test(s->tri_verts + 9 * (m[0] + i), r, &t, &b1, &b2)) {
                        if (any_hit) { h->prim = 1; return 1; }
                        hit = 1;
                        h->prim = m[0] + i; h->t = t; h->b1 = b1; h->b2 = b2;
                        r->maxt = t;
                    }
                }
                if (todo_idx == 0) break;
                node_idx = todo[--todo_idx];
            } else {
                if (todo_idx >= 64) return -2;
                if (neg[m[2]]) { todo[todo_idx++] = node_idx + 1; node_idx = m[0]; }
                else { todo[todo_idx++] = m[0]; node_idx = node_idx + 1; }
            }
        } else {
            if (todo_idx == 0) break;
            node_idx = todo[--todo_idx];
        }
    }
    if (any_hit) h->prim = 0;
    return hit;
}

/* batch entry; same argument meaning as pt_intersect (host pointers here) */
int pto_intersect(void *scene, const double *rays, int64_t n, int any_hit,
                  int32_t *prim, double *t, double *b1, double *b2, pt_stats *stats) {
    const oscene *s = (const oscene *)scene;
    if (!s || !rays || !prim || n < 0) return PT_ERR_ARG;
    uint64_t nv = 0, nt = 0;
    for (int64_t i = 0; i < n; i++) {
        oray r;
        r.o = V(rays[0 * n + i], rays[1 * n + i], rays[2 * n + i]);
        r.d = V(rays[3 * n + i], rays[4 * n + i], rays[5 * n + i]);
        r.mint = rays[6 * n + i]; r.maxt = rays[7 * n + i];
        ohit h;
        int rc = bvh_traverse(s, &r, any_hit, &h, &nv, &nt);
        if (rc == -2) return PT_ERR_DEPTH;
        prim[i] = h.prim;
        if (!any_hit) { if (t) t[i] = r.maxt; if (b1) b1[i] = h.b1; if (b2) b2[i] = h.b2; }
    }
    if (stats) {
        stats->node_visits += nv; stats->tri_tests += nt;
        if (any_hit) { stats->rays_any += (uint64_t)n; stats->node_visits_any += nv; stats->tri_tests_any += nt; }
        else stats->rays_closest += (uint64_t)n;
    }
    return PT_OK;
}

/* ----------------------------------------------------------------------------------
 * differential geometry + BSDF frame
 * -------------------------------------------------------------------------------- */
typedef struct {
    v3 p, nn, sn, tn;      /* hit point, shading(=geometric) normal, BSDF frame */
    v3 dpdu, dpdv;
    double u, v, rEps;
    const pt_material *mat;
    pt_material tmat;      /* the material with its textured lobe colours evaluated at this vertex */
} overtex;

/* ---- textures (texture/texture/__init__.py) over constant children, evaluated where Material.get_bsdf
 * evaluates Kd / Ks (material/material.py:131,175-176) on a copy of the shading geometry whose differentials
 * are zero (dg_s.copy(), geometry/diffgeom.py:21-42,47-48), so Chekcerboard2DTexture's closed-form
 * antialiasing takes its "inside one check" branch (texture/__init__.py:156-166) = point sampling. */
static v3 xform_point(const double *m, v3 p) {           /* Transform.__call__(Point), transform/__init__.py:55-60 */
    double x = ((m[0] * p.x + m[1] * p.y) + m[2] * p.z) + m[3];
    double y = ((m[4] * p.x + m[5] * p.y) + m[6] * p.z) + m[7];
    double z = ((m[8] * p.x + m[9] * p.y) + m[10] * p.z) + m[11];
    double w = ((m[12] * p.x + m[13] * p.y) + m[14] * p.z) + m[15];
    if (w < 1. - PT_EPS || w > 1. + PT_EPS) { x /= w; y /= w; z /= w; }      /* util.ne_unity */
    return V(x, y, z);
}
static void tex_map2d(const pt_texture *t, const overtex *vx, double *s_out, double *t_out) {
    const double *m = t->m;
    switch (t->mapping) {
    case PT_MAP_UV:            /* texturemap:45-47 */
        *s_out = m[0] * vx->u + m[2];
        *t_out = m[1] * vx->v + m[3];
        return;
    case PT_MAP_PLANAR: {      /* texturemap:142-145 */
        v3 v = vx->p;
        *s_out = m[6] + vdot(v, V(m[0], m[1], m[2]));
        *t_out = m[7] + vdot(v, V(m[3], m[4], m[5]));
        return;
    }
    default: {                 /* PT_MAP_SPHERICAL, texturemap:59-68 */
        v3 v = vnormalize(xform_point(m, vx->p));
        double cz = v.z < -1. ? -1. : (v.z > 1. ? 1. : v.z);
        double theta = acos(cz);                          /* geometry/utility.py:106-112 */
        double ph = atan2(v.y, v.x);
        if (!(ph > 0.)) ph = ph + 2. * M_PI;              /* geometry/utility.py:121-122 */
        *s_out = theta * (1. / M_PI);
        *t_out = ph * (1. / (2. * M_PI));
        return;
    }
    }
}
/* Perlin noise and turbulence as the reference executes them.  NOISE_PERM: pytracer/data/noise.py (Perlin's
 * reference permutation; the reference stores it twice over, 512 entries, so that x + 1 = 256 needs no wrap —
 * reducing every index modulo 256 is the same look-up). */
static const unsigned char NOISE_PERM[256] = {
    151, 160, 137,  91,  90,  15, 131,  13, 201,  95,  96,  53, 194, 233,   7, 225,
    140,  36, 103,  30,  69, 142,   8,  99,  37, 240,  21,  10,  23, 190,   6, 148,
    247, 120, 234,  75,   0,  26, 197,  62,  94, 252, 219, 203, 117,  35,  11,  32,
     57, 177,  33,  88, 237, 149,  56,  87, 174,  20, 125, 136, 171, 168,  68, 175,
     74, 165,  71, 134, 139,  48,  27, 166,  77, 146, 158, 231,  83, 111, 229, 122,
     60, 211, 133, 230, 220, 105,  92,  41,  55,  46, 245,  40, 244, 102, 143,  54,
     65,  25,  63, 161,   1, 216,  80,  73, 209,  76, 132, 187, 208,  89,  18, 169,
    200, 196, 135, 130, 116, 188, 159,  86, 164, 100, 109, 198, 173, 186,   3,  64,
     52, 217, 226, 250, 124, 123,   5, 202,  38, 147, 118, 126, 255,  82,  85, 212,
    207, 206,  59, 227,  47,  16,  58,  17, 182, 189,  28,  42, 223, 183, 170, 213,
    119, 248, 152,   2,  44, 154, 163,  70, 221, 153, 101, 155, 167,  43, 172,   9,
    129,  22,  39, 253,  19,  98, 108, 110,  79, 113, 224, 232, 178, 185, 112, 104,
    218, 246,  97, 228, 251,  34, 242, 193, 238, 210, 144,  12, 191, 179, 162, 241,
     81,  51, 145, 235, 249,  14, 239, 107,  49, 192, 214,  31, 181, 199, 106, 157,
    184,  84, 204, 176, 115, 121,  50,  45, 127,   4, 150, 254, 138, 236, 205,  93,
    222, 114,  67,  29,  24,  72, 243, 141, 128, 195,  78,  66, 215,  61, 156, 180,
};
/* _grad, texture/utility.py:104-114.  `if h & 1: u = -1` assigns the constant -1 (pbrt negates u). */
static double noise_grad(int x, int y, int z, double dx, double dy, double dz) {
    int h = NOISE_PERM[(NOISE_PERM[(NOISE_PERM[x & 255] + y) & 255] + z) & 255];
    h &= 15;
    double u = (h < 8 || h == 12 || h == 13) ? dx : dy;
    double v = (h < 4 || h == 12 || h == 13) ? dy : dz;
    if (h & 1) u = -1.;
    if (h & 2) v = -v;
    return u + v;
}
static inline double olerp(double t, double v1, double v2) { return (1. - t) * v1 + t * v2; }   /* util.lerp, utility.py:89-91 */
/* noise(), texture/utility.py:121-164.  y1 is interpolated with wz (pbrt uses wy). */
static double perlin_noise(v3 p) {
    double fx = floor(p.x), fy = floor(p.y), fz = floor(p.z);
    double dx = p.x - fx, dy = p.y - fy, dz = p.z - fz;
    int xi = (int)((long long)fx & 255), yi = (int)((long long)fy & 255), zi = (int)((long long)fz & 255);
    double w000 = noise_grad(xi, yi, zi, dx, dy, dz);
    double w100 = noise_grad(xi + 1, yi, zi, dx - 1, dy, dz);
    double w010 = noise_grad(xi, yi + 1, zi, dx, dy - 1, dz);
    double w110 = noise_grad(xi + 1, yi + 1, zi, dx - 1, dy - 1, dz);
    double w001 = noise_grad(xi, yi, zi + 1, dx, dy, dz - 1);
    double w101 = noise_grad(xi + 1, yi, zi + 1, dx - 1, dy, dz - 1);
    double w011 = noise_grad(xi, yi + 1, zi + 1, dx, dy - 1, dz - 1);
    double w111 = noise_grad(xi + 1, yi + 1, zi + 1, dx - 1, dy - 1, dz - 1);
    double wx = dx * dx * dx * (10. - 15. * dx + 6. * dx * dx);
    double wy = dy * dy * dy * (10. - 15. * dy + 6. * dy * dy);
    double wz = dz * dz * dz * (10. - 15. * dz + 6. * dz * dz);
    double x00 = olerp(wx, w000, w100), x10 = olerp(wx, w010, w110);
    double x01 = olerp(wx, w001, w101), x11 = olerp(wx, w011, w111);
    double y0 = olerp(wy, x00, x10), y1 = olerp(wz, x01, x11);
    return olerp(wz, y0, y1);
}
/* turbulence(), texture/utility.py:78-101, with the zero differentials every material look-up has (material.py:128
 * dg_s.copy(), diffgeom.py:40-42): s2 = 0, np.log2(0) = -inf, foct = min(max_octaves, inf) = max_octaves.
 * smooth_step (:43-50) clips its argument INTO [lo, hi] instead of normalising it. */
static double turbulence(v3 p, double omega, double max_octaves) {
    double foct = max_octaves, fo = floor(foct);
    int octaves = fo > 64. ? 64 : (fo > 0. ? (int)fo : 0);
    double s = 0., l = 1., o = 1.;
    for (int i = 0; i < octaves; i++) {
        s += o * fabs(perlin_noise(vmul(p, l)));
        l *= 1.99;
        o *= omega;
    }
    double par = foct - fo;
    double v = par < .3 ? .3 : (par > .7 ? .7 : par);
    s += o * (v * v * (3. - 2. * v)) * fabs(perlin_noise(vmul(p, l)));
    return s;
}

static v3 tex_eval(const pt_texture *t, const overtex *vx) {
    double s = 0., tt = 0.;
    switch (t->kind) {
    case PT_TEX_WRINKLED: {    /* WrinkledTexture.__call__, texture/__init__.py:262-266 */
        /* v[1][0]: the constant of a ScaleTexture(constant, WrinkledTexture) (texture/__init__.py:49-61), else 1 */
        double w = t->v[1][0] * turbulence(xform_point(t->m, vx->p), t->v[0][0], t->v[0][1]);
        return V(w, w, w);
    }
    case PT_TEX_CHECKER2D:     /* :147-152 / :162-166 */
        tex_map2d(t, vx, &s, &tt);
        return ((((long long)floor(s) + (long long)floor(tt)) & 1) == 0) ? V(t->v[0][0], t->v[0][1], t->v[0][2])
                                                                          : V(t->v[1][0], t->v[1][1], t->v[1][2]);
    case PT_TEX_CHECKER3D: {   /* :191-196 */
        v3 p = xform_point(t->m, vx->p);
        return ((((long long)floor(p.x) + (long long)floor(p.y) + (long long)floor(p.z)) & 1) == 0)
                   ? V(t->v[0][0], t->v[0][1], t->v[0][2]) : V(t->v[1][0], t->v[1][1], t->v[1][2]);
    }
    case PT_TEX_BILERP: {      /* :101-104 */
        tex_map2d(t, vx, &s, &tt);
        v3 a = vmul(V(t->v[0][0], t->v[0][1], t->v[0][2]), (1. - s) * (1. - tt));
        v3 b = vmul(V(t->v[1][0], t->v[1][1], t->v[1][2]), (1. - s) * tt);
        v3 c = vmul(V(t->v[2][0], t->v[2][1], t->v[2][2]), (1. - tt) * s);
        v3 d = vmul(V(t->v[3][0], t->v[3][1], t->v[3][2]), s * tt);
        return vadd(vadd(vadd(a, b), c), d);
    }
    default:                   /* PT_TEX_UV :114-116 */
        tex_map2d(t, vx, &s, &tt);
        return V(s - floor(s), tt - floor(tt), 0.);
    }
}
/* lobe colour = R * util.clip(texture(dgs)) (material.py:131: r = util.clip(self.Kd(dgs))) */
static void apply_textures(const oscene *sc, overtex *vx) {
    int any = 0;
    for (int k = 0; k < vx->mat->n_lobes; k++) any |= vx->mat->lobes[k].tex;
    if (!any) return;
    const pt_material *m = vx->mat;
    int n = 0;
    for (int k = 0; k < m->n_lobes; k++) {
        pt_lobe l = m->lobes[k];
        const int ti = l.tex & 0xFFFF;
        if (ti) {
            v3 c = tex_eval(sc->textures + (ti - 1), vx);
            c = V(c.x < 0. ? 0. : c.x, c.y < 0. ? 0. : c.y, c.z < 0. ? 0. : c.z);
            l.R[0] *= c.x; l.R[1] *= c.y; l.R[2] *= c.z;
            /* `if not k.is_black()`: the lobe is not pushed at this hit (spectrum.py:182-187) */
            if ((l.tex & PT_LOBE_DROP_IF_BLACK) && l.R[0] < PT_EPS && l.R[1] < PT_EPS && l.R[2] < PT_EPS) continue;
        }
        const int pi = PT_LOBE_PARAM_TEX(l.tex);
        if (pi) {   /* a float texture as the lobe's parameter */
            double f = tex_eval(sc->textures + (pi - 1), vx).x;
            if (l.type == PT_LOBE_OREN_NAYAR) {
                /* MatteMaterial.get_bsdf, material.py:132-139 + OrenNayar.__init__, orennayar.py:36-42 */
                double sigma = f < 0. ? 0. : (f > 90. ? 90. : f);
                if (sigma == 0.) { l.type = PT_LOBE_LAMBERT; l.a = 0.; l.b = 0.; }
                else {
                    double sig = sigma * (M_PI / 180.);
                    double sig_sq = sig * sig;
                    l.a = 1. - (sig_sq / (2. * (sig_sq + 0.33)));
                    l.b = .45 * sig_sq / (sig_sq + .09);
                }
            } else {
                /* Plastic / Uber: Blinn(1. / rough), material.py:178-179,349-350; exponent clipped, torrance.py:66-67 */
                double e = 1. / f;
                l.c = e > 10000. ? 10000. : e;
            }
        }
        l.tex = 0;
        vx->tmat.lobes[n++] = l;
    }
    vx->tmat.n_lobes = n;
    vx->mat = &vx->tmat;
}

/* Triangle.intersect tail (triangle.py:209-243) with the default uvs (0,0),(1,0),(1,1)
 * (triangle.py:125-127), DifferentialGeometry.__init__ (geometry/diffgeom.py:21-42) and
 * BSDF.__init__ (reflection/bsdf.py:78-94).  get_shading_geometry returns dg unchanged
 * when the mesh has no N/S (triangle.py:323-326); Material.get_bsdf copies it, which
 * recomputes the same nn. */
static void make_vertex(const oscene *s, const oray *r, const ohit *h, overtex *vx) {
    const double *tv = s->tri_verts + 9 * h->prim;
    v3 p1 = V(tv[0], tv[1], tv[2]), p2 = V(tv[3], tv[4], tv[5]), p3 = V(tv[6], tv[7], tv[8]);
    double uv00 = 0., uv01 = 0., uv10 = 1., uv11 = 0., uv20 = 1., uv21 = 1.;      /* triangle.py:125-131 */
    if (s->tri_uv) {
        const double *q = s->tri_uv + 6 * (size_t)h->prim;
        uv00 = q[0]; uv01 = q[1]; uv10 = q[2]; uv11 = q[3]; uv20 = q[4]; uv21 = q[5];
    }
    double du1 = uv00 - uv20, du2 = uv10 - uv20, dv1 = uv01 - uv21, dv2 = uv11 - uv21;
    v3 dp1 = vsub(p1, p3), dp2 = vsub(p2, p3);
    double det = du1 * dv2 - du2 * dv1;       /* == 1 for the default uvs */
    if (det == 0.) {
        /* triangle.py:219-221: _, dpdu, dpdv = coordinate_system(normalize(e2 x e1)), geometry/utility.py:40-52 */
        v3 v1 = vnormalize(vcross(vsub(p3, p1), vsub(p2, p1)));
        v3 v2;
        if (vlen(v1) == 0.) { vx->dpdu = V(0., 1., 0.); vx->dpdv = V(0., 0., 1.); }
        else {
            if (fabs(v1.x) > fabs(v1.y)) {
                double inv_len = 1. / sqrt(v1.x * v1.x + v1.z * v1.z);
                v2 = V(-v1.z * inv_len, 0., v1.x * inv_len);
            } else {
                double inv_len = 1. / sqrt(v1.y * v1.y + v1.z * v1.z);
                v2 = V(0., -v1.z * inv_len, v1.y * inv_len);
            }
            vx->dpdu = v2;
            vx->dpdv = vcross(v1, v2);
        }
    } else {
        double detInv = 1.0 / det;
        vx->dpdu = vmul(vsub(vmul(dp1, dv2), vmul(dp2, dv1)), detInv);
        vx->dpdv = vmul(vadd(vmul(dp1, -du2), vmul(dp2, du1)), detInv);
    }
    double b0 = 1.0 - h->b1 - h->b2;
    vx->u = b0 * uv00 + h->b1 * uv10 + h->b2 * uv20;
    vx->v = b0 * uv01 + h->b1 * uv11 + h->b2 * uv21;
    vx->p = vadd(r->o, vmul(r->d, h->t));     /* Ray.__call__, geometry/__init__.py:320-324 */
    vx->nn = vnormalize(vcross(vx->dpdu, vx->dpdv));
    if (s->tri_flip[h->prim]) vx->nn = vmul(vx->nn, -1.0);
    vx->rEps = 1e-3 * h->t;
    vx->sn = vnormalize(vx->dpdu);
    vx->tn = vcross(vx->nn, vx->sn);
    vx->mat = s->materials + s->tri_mat[h->prim];
    apply_textures(s, vx);
}

/* BSDF.w2l, bsdf.py:187-194 */
static inline v3 w2l(const overtex *vx, v3 v) { return V(vdot(v, vx->sn), vdot(v, vx->tn), vdot(v, vx->nn)); }
/* BSDF.l2w AS WRITTEN (bsdf.py:196-205): every basis vector is scaled by the SAME
 * component, so the result is not a rotation and not unit length. */
static inline v3 l2w(const overtex *vx, v3 v) {
    return V(vx->sn.x * v.x + vx->tn.x * v.x + vx->nn.x * v.x,
             vx->sn.y * v.y + vx->tn.y * v.y + vx->nn.y * v.y,
             vx->sn.z * v.z + vx->tn.z * v.z + vx->nn.z * v.z);
}

static inline unsigned lobe_flags(int type) {
    switch (type) {
    case PT_LOBE_LAMBERT: case PT_LOBE_OREN_NAYAR: return BDF_REFLECTION | BDF_DIFFUSE;
    case PT_LOBE_MICROFACET_BLINN: case PT_LOBE_MICROFACET_BLINN_COND: case PT_LOBE_MEASURED_IRISO:
        return BDF_REFLECTION | BDF_GLOSSY;     /* measured.py:47 */
    case PT_LOBE_SPEC_REFLECT: return BDF_REFLECTION | BDF_SPECULAR;
    default: return BDF_TRANSMISSION | BDF_SPECULAR;
    }
}
/* BDF.match_flag, bdf.py:166-170 */
static inline int match_flag(int type, unsigned flags) { unsigned t = lobe_flags(type); return (t & flags) == t; }

/* fr_diel AS WRITTEN (reflection/utility.py:46-57; r_para denominator uses eta_t twice, Q8) */
static inline double fr_diel(double cos_i, double cos_t, double eta_i, double eta_t) {
    double r_para = ((eta_t * cos_i) - (eta_i * cos_t)) / ((eta_t * cos_i) + (eta_t * cos_t));
    double r_perp = ((eta_i * cos_i) - (eta_t * cos_t)) / ((eta_i * cos_i) + (eta_t * cos_t));
    return .5 * (r_para * r_para + r_perp * r_perp);
}
/* FresnelDielectric.__call__, reflection/fresnel.py:54-72 */
static inline double fresnel_dielectric(double cosi, double eta_i, double eta_t) {
    double ci = clipd(cosi, -1., 1.);
    double ei = eta_i, et = eta_t;
    if (cosi < 0.) { double tmp = ei; ei = et; et = tmp; }
    double st = ei / et * sqrt(max0(1. - ci * ci));
    if (st >= 1.) return 1.;
    double ct = sqrt(max0(1. - st * st));
    return fr_diel(fabs(ci), ct, ei, et);
}

/* reflection/utility.py:31-42 */
static inline double sin_theta(v3 w) { return sqrt(max0(1. - w.z * w.z)); }
static inline double cos_phi(v3 w) {
    double s2 = max0(1. - w.z * w.z);
    return s2 == 0. ? 1. : clipd(w.x / sqrt(s2), -1., 1.);
}
static inline double sin_phi(v3 w) {
    double s2 = max0(1. - w.z * w.z);
    return s2 == 0. ? 0. : clipd(w.y / sqrt(s2), -1., 1.);
}

/* Blinn.D, torrance.py:69-70 */
static inline double blinn_D(double e, v3 wh) { return (e + 2) * PT_INV_2PI * pow(fabs(wh.z), e); }
/* Blinn.pdf, torrance.py:95-102 */
static inline double blinn_pdf(double e, v3 wo, v3 wi) {
    v3 wh = vnormalize(vadd(wo, wi));
    double ct = fabs(wh.z);
    if (vdot(wo, wh) <= 0.) return 0.;
    return ((e + 1.) * pow(ct, e)) / (2. * PT_PI * 4. * vdot(wo, wh));
}

/* fr_cond, reflection/utility.py:60-73, one channel; FresnelConductor.__call__ passes |cos| (fresnel.py:40-41) */
static inline double fr_cond(double cos_i, double eta, double k) {
    double tmp = (eta * eta + k * k) * cos_i * cos_i;
    double r_para_sq = (tmp - (2. * eta * cos_i) + 1.) / (tmp + (2. * eta * cos_i) + 1.);
    tmp = eta * eta + k * k;
    double r_perp_sq = (tmp - (2. * eta * cos_i) + cos_i * cos_i) / (tmp + (2. * eta * cos_i) + cos_i * cos_i);
    return .5 * (r_para_sq + r_perp_sq);
}

/* geometry/utility.py:115-122 */
static inline double spherical_phi(v3 v) {
    double p = atan2(v.y, v.x);
    return p > 0. ? p : p + 2 * PT_PI;
}
/* brdf_remap, reflection/utility.py:76-90 (Marschner's isotropic parametrisation) */
static v3 brdf_remap(v3 wo, v3 wi) {
    double dphi = spherical_phi(wi) - spherical_phi(wo);
    if (dphi < 0.) dphi += 2. * PT_PI;
    if (dphi > 2. * PT_PI) dphi -= 2. * PT_PI;
    if (dphi > PT_PI) dphi = 2. * PT_PI - dphi;
    return V(sin_theta(wi) * sin_theta(wo), dphi * PT_INV_PI, wi.z * wo.z);
}
/* IrIsotropicBRDF.f, reflection/bdf/measured.py:51-66.  tree.query_ball_point(m, r) (scipy cKDTree, p = 2)
 * returns the samples with squared distance <= r*r; restated as a scan of the table.  The reference adds the
 * samples in the tree's traversal order, this scan in table order: the sums of these positive terms agree to
 * rounding.  No sample at all (an empty table) gives 0/0 = NaN like the reference. */
static v3 measured_f(const pt_lobe *l, v3 wo, v3 wi) {
    const double *pp, *pv;
    memcpy(&pp, &l->eta[0], sizeof(pp));
    memcpy(&pv, &l->eta[1], sizeof(pv));
    const int64_t n = (int64_t)l->b;
    v3 m = brdf_remap(wo, wi);
    double max_dist = .01;
    for (;;) {
        const double r2 = max_dist * max_dist;
        int64_t cnt = 0;
        for (int64_t i = 0; i < n; i++)
            if (vsqlen(vsub(V(pp[3 * i], pp[3 * i + 1], pp[3 * i + 2]), m)) <= r2) cnt++;
        if (cnt > 2 || max_dist > 1.5) {
            double sum_wt = 0.;
            v3 v = V(0, 0, 0);
            for (int64_t i = 0; i < n; i++) {
                double d2 = vsqlen(vsub(V(pp[3 * i], pp[3 * i + 1], pp[3 * i + 2]), m));
                if (d2 <= r2) {
                    double wt = exp(d2 * -100.);
                    sum_wt += wt;
                    v = vadd(v, vmul(V(pv[3 * i], pv[3 * i + 1], pv[3 * i + 2]), wt));
                }
            }
            return vdiv(V(max0(v.x), max0(v.y), max0(v.z)), sum_wt);     /* v.clip() / sum_wt */
        }
        max_dist *= 1.414;
    }
}

/* lobe.f(wo, wi) in the local frame */
static v3 lobe_f(const pt_lobe *l, v3 wo, v3 wi) {
    v3 R = V(l->R[0], l->R[1], l->R[2]);
    switch (l->type) {
    case PT_LOBE_LAMBERT:                       /* bdf.py:327-332 */
        return vmul(R, PT_INV_PI);
    case PT_LOBE_OREN_NAYAR: {                  /* orennayar.py:44-62 */
        double st_i = sin_theta(wi), st_o = sin_theta(wo);
        double mc = 0.;
        if (st_i > PT_EPS && st_o > PT_EPS) {
            double dcos = cos_phi(wi) * cos_phi(wo) + sin_phi(wi) * sin_phi(wo);
            mc = max0(dcos);
        }
        double sa, tb;
        if (fabs(wi.z) > fabs(wo.z)) { sa = st_o; tb = st_i / fabs(wi.z); }
        else { sa = st_i; tb = st_o / fabs(wo.z); }
        return vmul(vmul(R, PT_INV_PI), (l->a + l->b * mc * sa * tb));
    }
    case PT_LOBE_MICROFACET_BLINN_COND: {       /* torrance.py:200-216 with a Spectrum-valued Fresnel term */
        double ct_i = wi.z, ct_o = wo.z;
        if (ct_i == 0. || ct_o == 0.) return V(0, 0, 0);
        v3 wh = vnormalize(vadd(wi, wo));
        double ch = fabs(vdot(wi, wh));
        v3 F = V(fr_cond(ch, l->eta[0], l->k[0]), fr_cond(ch, l->eta[1], l->k[1]), fr_cond(ch, l->eta[2], l->k[2]));
        double awh = fabs(wh.z), adot = fabs(vdot(wo, wh));
        double g1 = 2. * awh * fabs(wo.z) / adot, g2 = 2. * awh * fabs(wi.z) / adot;
        double gm = (g2 < g1) ? g2 : g1;
        double G = gm < 1. ? gm : 1.;
        double D = blinn_D(l->c, wh);
        return vdiv(vmulv(vmul(vmul(R, D), G), F), (4. * ct_i * ct_o));
    }
    case PT_LOBE_MICROFACET_BLINN: {            /* torrance.py:200-216 */
        double ct_i = wi.z, ct_o = wo.z;
        if (ct_i == 0. || ct_o == 0.) return V(0, 0, 0);
        v3 wh = vnormalize(vadd(wi, wo));
        double ct_h = vdot(wi, wh);
        double F = fresnel_dielectric(ct_h, l->a, l->b);
        double awh = fabs(wh.z), adot = fabs(vdot(wo, wh));
        double g1 = 2. * awh * fabs(wo.z) / adot, g2 = 2. * awh * fabs(wi.z) / adot;
        double gm = (g2 < g1) ? g2 : g1;         /* python min(a,b): b if b < a else a */
        double G = gm < 1. ? gm : 1.;
        double D = blinn_D(l->c, wh);
        return vdiv(vmul(vmul(vmul(R, D), G), F), (4. * ct_i * ct_o));
    }
    case PT_LOBE_MEASURED_IRISO:
        return measured_f(l, wo, wi);
    default:                                     /* specular lobes: bdf.py:241-246,274-279 */
        return V(0, 0, 0);
    }
}

/* BDF.pdf (bdf.py:97-106), Microfacet.pdf (torrance.py:225-228), specular pdf = 0 */
static double lobe_pdf(const pt_lobe *l, v3 wo, v3 wi) {
    switch (l->type) {
    case PT_LOBE_LAMBERT: case PT_LOBE_OREN_NAYAR: case PT_LOBE_MEASURED_IRISO:
        if (wo.z * wi.z > 0.) return fabs(wi.z) * PT_INV_PI;
        return 0.;
    case PT_LOBE_MICROFACET_BLINN: case PT_LOBE_MICROFACET_BLINN_COND:
        if (wi.z * wo.z <= 0.) return 0.;
        return blinn_pdf(l->c, wo, wi);
    default:
        return 0.;
    }
}

/* lobe.sample_f(wo,u1,u2) → pdf, wi, f.  Returns pdf (0 ⇒ wi/f undefined). */
static double lobe_sample_f(const pt_lobe *l, v3 wo, double u1, double u2, v3 *wi_out, v3 *f_out) {
    v3 R = V(l->R[0], l->R[1], l->R[2]);
    switch (l->type) {
    case PT_LOBE_LAMBERT: case PT_LOBE_OREN_NAYAR: case PT_LOBE_MEASURED_IRISO: {
        /* BDF.sample_f (bdf.py:108-126) with cosine_sample_hemisphere AS WRITTEN
         * (montecarlo/utility.py:171-184 via uniform_sample_sphere :83-95; Q6) */
        double z = 1. - 2. * u1;
        double r = sqrt(max0(1. - z * z));
        double phi = 2. * PT_PI * u2;
        v3 wi = V(r * cos(phi), r * sin(phi), z);
        wi.z = sqrt(max0(1. - wi.x * wi.x - wi.y * wi.y));
        if (wo.z < 0.) wi.z *= -1.;
        *wi_out = wi;
        *f_out = lobe_f(l, wo, wi);
        return lobe_pdf(l, wo, wi);
    }
    case PT_LOBE_MICROFACET_BLINN: case PT_LOBE_MICROFACET_BLINN_COND: {   /* torrance.py:218-223 + Blinn.sample_f :72-93 */
        double e = l->c;
        double ct = pow(u1, 1. / (e + 1));
        double st = sqrt(max0(1. - ct * ct));
        double phi = u2 * 2. * PT_PI;
        v3 wh = V(st * cos(phi), st * sin(phi), ct);   /* geometry/utility.py:92-101 */
        if (wo.z * wh.z <= 0.) wh = vmul(wh, -1.);
        double dwh = vdot(wo, wh);
        v3 wi = vadd(vneg(wo), vmul(wh, 2. * dwh));
        double pdf = ((e + 1.) * pow(ct, e)) / (2. * PT_PI * 4. * dwh);
        if (dwh <= 0.) pdf = 0.;
        *wi_out = wi;
        if (wi.z * wo.z <= 0.) *f_out = V(0, 0, 0);
        else *f_out = lobe_f(l, wo, wi);
        return pdf;
    }
    case PT_LOBE_SPEC_REFLECT: {                /* bdf.py:251-256 */
        v3 wi = V(-wo.x, -wo.y, wo.z);
        double F = (l->c != 0.) ? 1. : fresnel_dielectric(wo.z, l->a, l->b);
        *wi_out = wi;
        *f_out = vdiv(vmul(R, F), fabs(wi.z));
        return 1.;
    }
    default: {                                   /* SpecularTransmission.sample_f, bdf.py:284-307 */
        double ei = l->a, et = l->b;
        if (wo.z > 0.) { double tmp = ei; ei = et; et = tmp; }
        double si_sq = max0(1. - wo.z * wo.z);
        double eta = ei / et;
        double st_sq = eta * eta * si_sq;
        if (st_sq >= 1.) return 0.;
        double ct = sqrt(max0(1. - st_sq));
        if (wo.z > 0.) ct = -ct;
        v3 wi = V(eta * (-wo.x), eta * (-wo.y), ct);
        double F = fresnel_dielectric(wo.z, l->a, l->b);
        *wi_out = wi;
        *f_out = vdiv(vmul(R, (et * et) / (ei * ei) * (1. - F)), fabs(wi.z));
        return 1.;
    }
    }
}

/* BSDF.f, bsdf.py:207-224 */
static v3 bsdf_f(const overtex *vx, v3 wo_w, v3 wi_w, unsigned flags) {
    v3 wi = w2l(vx, wi_w), wo = w2l(vx, wo_w);
    if (vdot(wi_w, vx->nn) * vdot(wo_w, vx->nn) > 0.) flags &= ~BDF_TRANSMISSION;
    else flags &= ~BDF_REFLECTION;
    v3 f = V(0, 0, 0);
    for (int i = 0; i < vx->mat->n_lobes; i++)
        if (match_flag(vx->mat->lobes[i].type, flags)) f = vadd(f, lobe_f(&vx->mat->lobes[i], wo, wi));
    return f;
}

/* BSDF.pdf, bsdf.py:156-167 */
static double bsdf_pdf(const overtex *vx, v3 wo_w, v3 wi_w, unsigned flags) {
    if (vx->mat->n_lobes == 0) return 0.;
    v3 wo = w2l(vx, wo_w), wi = w2l(vx, wi_w);
    double pdf = 0.; int n_match = 0;
    for (int i = 0; i < vx->mat->n_lobes; i++)
        if (match_flag(vx->mat->lobes[i].type, flags)) { n_match++; pdf += lobe_pdf(&vx->mat->lobes[i], wo, wi); }
    return n_match > 0 ? pdf / n_match : 0.;
}

/* BSDF.sample_f AS IT EXECUTES (bsdf.py:100-154): `bdf.type & SPECULAR` is an object and
 * always truthy, so neither the pdf sum (:132-135) nor the f re-evaluation (:141-152)
 * runs (Q2): pdf = lobe pdf / n_match, f = sampled lobe's f, wi through the faulty l2w. */
static double bsdf_sample_f(const overtex *vx, v3 wo_w, double u0, double u1, double u_com, unsigned flags,
                            v3 *wi_w, unsigned *type, v3 *f) {
    const pt_material *m = vx->mat;
    int n_match = 0;
    for (int i = 0; i < m->n_lobes; i++) if (match_flag(m->lobes[i].type, flags)) n_match++;
    *f = V(0, 0, 0); *type = 0; *wi_w = V(0, 0, 0);
    if (n_match == 0) return 0.;
    int cnt = (int)floor(u_com * n_match);
    if (cnt > n_match - 1) cnt = n_match - 1;
    const pt_lobe *l = 0;
    for (int i = 0; i < m->n_lobes; i++)
        if (match_flag(m->lobes[i].type, flags)) { if (cnt == 0) { l = &m->lobes[i]; break; } cnt--; }
    if (!l) return 0.;   /* unreachable for u_com >= 0 */
    v3 wo = w2l(vx, wo_w), wi, fl;
    double pdf = lobe_sample_f(l, wo, u0, u1, &wi, &fl);
    *type = lobe_flags(l->type);
    if (pdf == 0.) return 0.;
    *wi_w = l2w(vx, wi);
    if (n_match > 1) pdf /= n_match;
    *f = fl;
    return pdf;
}

/* ----------------------------------------------------------------------------------
 * lights
 * -------------------------------------------------------------------------------- */
static inline v3 xf_point(const double *m, v3 p) {   /* affine rows only (w == 1) */
    return V((m[0] * p.x + m[1] * p.y + m[2] * p.z) + m[3],
             (m[4] * p.x + m[5] * p.y + m[6] * p.z) + m[7],
             (m[8] * p.x + m[9] * p.y + m[10] * p.z) + m[11]);
}
static inline v3 xf_vec(const double *m, v3 v) {
    return V(m[0] * v.x + m[1] * v.y + m[2] * v.z, m[4] * v.x + m[5] * v.y + m[6] * v.z,
             m[8] * v.x + m[9] * v.y + m[10] * v.z);
}
/* Normal: inverse transpose, transform/__init__.py:66-69; minv = rows of the inverse */
static inline v3 xf_normal(const double *minv, v3 n) {
    return V(minv[0] * n.x + minv[4] * n.y + minv[8] * n.z, minv[1] * n.x + minv[5] * n.y + minv[9] * n.z,
             minv[2] * n.x + minv[6] * n.y + minv[10] * n.z);
}

/* geometry/utility.py:25-52 */
static void coordinate_system(v3 v1, v3 *v2, v3 *v3o) {
    if (vlen(v1) == 0.) { *v2 = V(0, 1, 0); *v3o = V(0, 0, 1); return; }
    if (fabs(v1.x) > fabs(v1.y)) {
        double inv_len = 1. / sqrt(v1.x * v1.x + v1.z * v1.z);
        *v2 = V(-v1.z * inv_len, 0., v1.x * inv_len);
    } else {
        double inv_len = 1. / sqrt(v1.y * v1.y + v1.z * v1.z);
        *v2 = V(0., -v1.z * inv_len, v1.y * inv_len);
    }
    *v3o = vcross(v1, *v2);
}

/* Sphere.intersect for a FULL sphere (shape/sphere.py:125-157; the clipping tests at
 * :165-183 cannot fire when zmin=-r, zmax=r, phiMax=2π).  np.roots → closed form. */
static int sphere_intersect(const pt_light *lt, const oray *rw, double *thit, v3 *phit_obj) {
    oray r;
    r.o = xf_point(lt->w2o, rw->o); r.d = xf_vec(lt->w2o, rw->d); r.mint = rw->mint; r.maxt = rw->maxt;
    double A = vsqlen(r.d);
    double B = 2 * vdot(r.d, V(r.o.x, r.o.y, r.o.z));
    double C = vsqlen(r.o) - lt->radius * lt->radius;
    double D = B * B - 4. * A * C;
    if (D <= 0. + PT_EPS) return 0;
    double rootD = sqrt(D);
    double q = (B < 0.) ? -.5 * (B - rootD) : -.5 * (B + rootD);
    double t0 = q / A, t1 = C / q;
    if (t0 > t1) { double tmp = t0; t0 = t1; t1 = tmp; }
    if (t0 > r.maxt || t1 < r.mint) return 0;
    double th = t0;
    if (t0 < r.mint) { th = t1; if (th > r.maxt) return 0; }
    *thit = th;
    if (phit_obj) *phit_obj = vadd(r.o, vmul(r.d, th));
    return 1;
}

/* dg.nn of a sphere hit (sphere.py:157-224 + diffgeom.py:30-38), full sphere */
static v3 sphere_hit_normal(const pt_light *lt, v3 phit) {
    const double phiMax = 2. * PT_PI, thetaMin = PT_PI, thetaMax = 0.;
    if (phit.x == 0. && phit.y == 0.) phit.x = PT_EPS * lt->radius;
    double theta = acos(clipd(phit.z / lt->radius, -1., 1.));
    double delta_theta = thetaMax - thetaMin;
    v3 dpdu = V(-phiMax * phit.y, phiMax * phit.x, 0.);
    double zrad = sqrt(phit.x * phit.x + phit.y * phit.y);
    double inv_zrad = 1. / zrad;
    double cphi = phit.x * inv_zrad, sphi = phit.y * inv_zrad;
    v3 dpdv = vmul(V(phit.z * cphi, phit.z * sphi, -lt->radius * sin(theta)), delta_theta);
    v3 nn = vnormalize(vcross(xf_vec(lt->o2w, dpdu), xf_vec(lt->o2w, dpdv)));
    /* swaps_handedness: det of the 3x3 < 0 (transform/__init__.py:266-267) */
    const double *m = lt->o2w;
    double det = m[0] * (m[5] * m[10] - m[6] * m[9]) - m[1] * (m[4] * m[10] - m[6] * m[8]) +
                 m[2] * (m[4] * m[9] - m[5] * m[8]);
    if ((lt->ro != 0) ^ (det < 0.)) nn = vmul(nn, -1.);
    return nn;
}

typedef struct { v3 Li, wi; double pdf; oray shadow; } olsample;

/* DiffuseAreaLight.sample_l (light.py:514-521) → ShapeSet.sample_p (light/utility.py:104-108,
 * one shape ⇒ index 0) → Sphere.sample_p (sphere.py:78-110), ShapeSet.pdf → Sphere.pdf_p
 * (sphere.py:112-123), VisibilityTester.set_segment (light/utility.py:34-42). */
static void sphere_light_sample_l(const pt_light *lt, v3 p, double pEps, double u1, double u2, olsample *o) {
    v3 ctr = V(lt->pos[0], lt->pos[1], lt->pos[2]);
    v3 wc = vnormalize(vsub(ctr, p));
    v3 wcx, wcy;
    coordinate_system(wc, &wcx, &wcy);
    v3 dcp = vsub(p, ctr);
    double sqd = dcp.x * dcp.x + dcp.y * dcp.y + dcp.z * dcp.z;     /* Point.sq_dist */
    v3 ps, ns;
    int inside = (sqd - lt->radius * lt->radius < PT_EPS);
    if (inside) {
        /* Sphere.sample AS WRITTEN (sphere.py:43-60; Q7), full sphere */
        const double phiMax = 2. * PT_PI, thetaMin = PT_PI, thetaMax = 0.;
        double z = 1. - 2. * u1;
        double r = sqrt(max0(1. - z * z));
        double ph = 2. * PT_PI * u2;
        v3 v = V(r * cos(ph), r * sin(ph), z);
        double sth = acos(clipd(v.z, -1., 1.));                      /* spherical_theta */
        double phi = sth * phiMax * PT_INV_2PI;
        double theta = thetaMin + sth * (thetaMax - thetaMin);
        double s = sin(theta), c = cos(theta);
        v = vmul(V(s * cos(phi), s * sin(phi), c), lt->radius);
        v.z = -lt->radius + v.z * (lt->radius - (-lt->radius));
        ns = vnormalize(xf_normal(lt->w2o, v));
        if (lt->ro) ns = vmul(ns, -1.);
        ps = xf_point(lt->o2w, v);
    } else {
        double st_max_sq = lt->radius * lt->radius / sqd;
        double ct_max = sqrt(max0(1. - st_max_sq));
        /* uniform_sample_cone with a basis, montecarlo/utility.py:222-226 */
        double ct = (1. - u1) * ct_max + u1 * 1.;
        double st = sqrt(1. - ct * ct);
        double phi = u2 * 2. * PT_PI;
        v3 dir = vadd(vadd(vmul(wcx, cos(phi) * st), vmul(wcy, sin(phi) * st)), vmul(wc, ct));
        oray r; r.o = p; r.d = dir; r.mint = PT_EPS; r.maxt = PT_INF;
        double thit;
        if (!sphere_intersect(lt, &r, &thit, 0)) thit = vdot(vsub(ctr, p), vnormalize(r.d));
        ps = vadd(r.o, vmul(r.d, thit));
        ns = vnormalize(vsub(ps, ctr));
        if (lt->ro) ns = vmul(ns, -1.);
    }
    v3 wi = vnormalize(vsub(ps, p));
    double pdf;
    if (inside) {
        /* Shape.pdf_p, shape/__init__.py:122-140 */
        oray r; r.o = p; r.d = wi; r.mint = PT_EPS; r.maxt = PT_INF;
        double thit; v3 ph;
        if (!sphere_intersect(lt, &r, &thit, &ph)) pdf = 0.;
        else {
            v3 nn = sphere_hit_normal(lt, ph);
            v3 hitp = vadd(r.o, vmul(r.d, thit));
            double area = (2. * PT_PI) * lt->radius * (lt->radius - (-lt->radius));
            pdf = vsqlen(vsub(p, hitp)) / (fabs(vdot(nn, vneg(wi))) * area);
        }
    } else {
        double st_max_sq = lt->radius * lt->radius / sqd;
        double ct_max = sqrt(max0(1. - st_max_sq));
        pdf = 1. / (2. * PT_PI * (1. - ct_max));                     /* uniform_cone_pdf */
    }
    /* DiffuseAreaLight.l, light.py:507-508 */
    if (vdot(ns, vneg(wi)) > 0.) o->Li = V(lt->emit[0], lt->emit[1], lt->emit[2]);
    else o->Li = V(0, 0, 0);
    o->wi = wi; o->pdf = pdf;
    double dist = vlen(vsub(p, ps));
    o->shadow.o = p; o->shadow.d = vdiv(vsub(ps, p), dist);
    o->shadow.mint = pEps; o->shadow.maxt = dist * (1. - PT_EPS);
}

/* SpotLight.sample_l (light.py:190-196, falloff :173-188; divides by the falloff, no 1/r², Q9)
 * and PointLight.sample_l (light.py:129-135) */
static void delta_light_sample_l(const pt_light *lt, v3 p, double pEps, olsample *o) {
    v3 pos = V(lt->pos[0], lt->pos[1], lt->pos[2]);
    v3 wi = vnormalize(vsub(pos, p));
    v3 I = V(lt->emit[0], lt->emit[1], lt->emit[2]);
    if (lt->kind == PT_LIGHT_SPOT) {
        v3 wl = vnormalize(xf_vec(lt->w2o, vneg(wi)));
        double ct = wl.z, fo;
        if (ct < lt->cos_width) fo = 0.;
        else if (ct > lt->cos_falloff) fo = 1.;
        else { double d = (ct - lt->cos_width) / (lt->cos_falloff - lt->cos_width); fo = d * d * d * d; }
        o->Li = vdiv(I, fo);
    } else {
        o->Li = vdiv(I, vsqlen(vsub(pos, p)));
    }
    o->wi = wi; o->pdf = 1.;
    double dist = vlen(vsub(p, pos));
    o->shadow.o = p; o->shadow.d = vdiv(vsub(pos, p), dist);
    o->shadow.mint = pEps; o->shadow.maxt = dist * (1. - 0.);
}

static void light_sample_l(const pt_light *lt, v3 p, double pEps, double u1, double u2, olsample *o) {
    if (lt->kind == PT_LIGHT_DIFFUSE_SPHERE) sphere_light_sample_l(lt, p, pEps, u1, u2, o);
    else delta_light_sample_l(lt, p, pEps, o);
}

/* Light.le(ray): emit for DiffuseAreaLight ("debugging purposes only", light.py:503-505), else 0 */
static v3 lights_le(const oscene *s) {
    v3 L = V(0, 0, 0);
    for (int i = 0; i < s->n_lights; i++)
        if (s->lights[i].kind == PT_LIGHT_DIFFUSE_SPHERE)
            L = vadd(L, V(s->lights[i].emit[0], s->lights[i].emit[1], s->lights[i].emit[2]));
    return L;
}

/* util.is_black on an array: all |c| < EPS (utility/utility.py:119-124) */
static inline int util_is_black(v3 s) { return fabs(s.x) < PT_EPS && fabs(s.y) < PT_EPS && fabs(s.z) < PT_EPS; }
/* CoefficientSpectrum.is_black: all c < EPS (spectral/spectrum.py:182-187) */
static inline int spec_is_black(v3 s) { return s.x < PT_EPS && s.y < PT_EPS && s.z < PT_EPS; }
/* RGBSpectrum.y, spectrum.py:587-589 */
static inline double spec_y(v3 s) { return (0.212671 * s.x + 0.715160 * s.y) + 0.072169 * s.z; }

typedef struct { const oscene *s; pt_stats *st; } octx;

static int occluded(octx *c, const oray *r) {
    oray rr = *r; ohit h;
    uint64_t nv = 0, nt = 0;
    int hit = bvh_traverse(c->s, &rr, 1, &h, &nv, &nt);
    if (c->st) { c->st->rays_any++; c->st->node_visits += nv; c->st->tri_tests += nt;
                 c->st->node_visits_any += nv; c->st->tri_tests_any += nt; }
    return hit == 1;
}
static int closest(octx *c, oray *r, ohit *h) {
    uint64_t nv = 0, nt = 0;
    int hit = bvh_traverse(c->s, r, 0, h, &nv, &nt);
    if (c->st) { c->st->rays_closest++; c->st->node_visits += nv; c->st->tri_tests += nt; }
    return hit == 1;
}

/* estimate_direct AS IT EXECUTES (integrator/utility.py:203-257): only the light-sampling
 * half contributes; the BSDF-sampling block is unreachable for non-specular flags (Q3). */
static v3 estimate_direct(octx *c, const pt_light *lt, const overtex *vx, v3 wo, double ul0, double ul1) {
    const unsigned flags = BDF_ALL & ~BDF_SPECULAR;
    v3 Ld = V(0, 0, 0);
    olsample ls;
    light_sample_l(lt, vx->p, vx->rEps, ul0, ul1, &ls);
    if (ls.pdf > 0. && !util_is_black(ls.Li)) {
        v3 f = bsdf_f(vx, wo, ls.wi, flags);
        if (!util_is_black(f) && !occluded(c, &ls.shadow)) {
            double adot = fabs(vdot(ls.wi, vx->nn));
            if (lt->kind != PT_LIGHT_DIFFUSE_SPHERE) {
                Ld = vadd(Ld, vmul(vmulv(f, ls.Li), (adot / ls.pdf)));
            } else {
                double bp = bsdf_pdf(vx, wo, ls.wi, flags);
                double ff = 1 * ls.pdf, g = 1 * bp;
                double weight = ff * ff / (ff * ff + g * g);       /* power_heuristic, montecarlo/utility.py:33-42 */
                Ld = vadd(Ld, vmul(vmulv(f, ls.Li), (adot * weight / ls.pdf)));
            }
        }
    }
    return Ld;
}

/* sample-table layout: row = [imageX,imageY,lens_u,lens_v,time, oneD arrays..., twoD arrays...]
 * (tools/oracle/gen_fixtures.py:flat_sample; sampler/sample.py:36-76) */
typedef struct {
    int dims;
    int n_lights;
    /* direct: per light offsets of LightSample com/pos and BSDFSample com/dir (direct.py:46-62) */
    int d_lcom[16], d_lpos[16], d_bcom[16], d_bdir[16], d_ns[16];
    /* path: per sampled depth (path.py:34-42) */
    int p_lcom[3], p_lpos[3], p_lnum[3], p_bcom[3], p_bdir[3], p_pcom[3], p_pdir[3];
} olayout;

static int make_layout(const oscene *s, const pt_integrator_desc *ig, olayout *L) {
    memset(L, 0, sizeof(*L));
    L->n_lights = s->n_lights;
    if (ig->kind == PT_INTEG_DIRECT) {
        if (s->n_lights > 16) return PT_ERR_CAPACITY;
        int n1 = 0, n2 = 0;
        for (int i = 0; i < s->n_lights; i++) { n1 += 2 * s->lights[i].ns; n2 += 2 * s->lights[i].ns; }
        int o1 = 5, o2 = 5 + n1;
        for (int i = 0; i < s->n_lights; i++) {
            int ns = s->lights[i].ns;
            L->d_ns[i] = ns;
            L->d_lcom[i] = o1; o1 += ns; L->d_lpos[i] = o2; o2 += 2 * ns;
            L->d_bcom[i] = o1; o1 += ns; L->d_bdir[i] = o2; o2 += 2 * ns;
        }
        L->dims = 5 + n1 + 2 * n2;
    } else if (ig->kind == PT_INTEG_WHITTED) {
        L->dims = 5;                                   /* no request_samples (whitted.py) */
    } else {
        int o1 = 5, o2 = 5 + 12;
        for (int i = 0; i < 3; i++) {
            L->p_lcom[i] = o1++; L->p_lpos[i] = o2; o2 += 2;
            L->p_lnum[i] = o1++;
            L->p_bcom[i] = o1++; L->p_bdir[i] = o2; o2 += 2;
            L->p_pcom[i] = o1++; L->p_pdir[i] = o2; o2 += 2;
        }
        L->dims = 35;
    }
    return PT_OK;
}

int pto_table_dims(void *scene, const pt_integrator_desc *ig) {
    olayout L;
    if (make_layout((const oscene *)scene, ig, &L) != PT_OK) return -1;
    return L.dims;
}

/* ---- production sample stream (no reference counterpart: np.random's global state cannot
 * be reproduced on a wavefront machine).  Same distributions as StratifiedSampler.generate
 * (stratified.py:45-89): jittered xs×ys image strata in sample order, uniform lens/time
 * (decorrelated by construction), Latin-hypercube 1D/2D tables with a hashed Fisher–Yates
 * permutation per (sample, array).  Every value is u(sid_mixed, 0x10000 + k). --------- */
static inline uint32_t mix_seed(uint32_t sid, uint32_t seed) { return sid + seed * 0x9E3779B1u; }

static void lhs_perm(int n, uint32_t sid, uint32_t key, int *perm) {
    for (int i = 0; i < n; i++) perm[i] = i;
    for (int i = n - 1; i > 0; i--) {
        int j = (int)floor(hash_uniform(sid, key + (uint32_t)i) * (i + 1));
        if (j > i) j = i;
        int t = perm[i]; perm[i] = perm[j]; perm[j] = t;
    }
}

static void gen_sample_row(const pt_sampler_desc *sp, const olayout *L, const pt_integrator_desc *ig,
                           int px, int py, int sidx, uint32_t sid_raw, double *row) {
    uint32_t sid = mix_seed(sid_raw, sp->seed);
    int ix = sidx % sp->xs, iy = sidx / sp->xs;
    double jx = sp->jitter ? hash_uniform(sid, 0x10000u + 0) : .5;
    double jy = sp->jitter ? hash_uniform(sid, 0x10000u + 1) : .5;
    row[0] = (ix + jx) / sp->xs + px;
    row[1] = (iy + jy) / sp->ys + py;
    row[2] = hash_uniform(sid, 0x10000u + 2);
    row[3] = hash_uniform(sid, 0x10000u + 3);
    row[4] = 0.;   /* time = lerp(t, s_open, s_close) with s_open == s_close == 0 in every config */
    if (ig->kind == PT_INTEG_WHITTED) return;
    if (ig->kind == PT_INTEG_PATH) {
        for (int k = 5; k < L->dims; k++) row[k] = hash_uniform(sid, 0x10000u + (uint32_t)k);
        return;
    }
    int perm[64];
    for (int i = 0; i < L->n_lights; i++) {
        int n = L->d_ns[i];
        const int off1[2] = {L->d_lcom[i], L->d_bcom[i]};
        const int off2[2] = {L->d_lpos[i], L->d_bdir[i]};
        for (int a = 0; a < 2; a++) {
            lhs_perm(n, sid, 0x20000u + (uint32_t)off1[a] * 64u, perm);
            for (int k = 0; k < n; k++)
                row[off1[a] + k] = (perm[k] + hash_uniform(sid, 0x10000u + (uint32_t)(off1[a] + k))) / n;
            lhs_perm(n, sid, 0x20000u + (uint32_t)off2[a] * 64u, perm);
            for (int k = 0; k < n; k++) {
                row[off2[a] + 2 * k] = (k + hash_uniform(sid, 0x10000u + (uint32_t)(off2[a] + 2 * k))) / n;
                row[off2[a] + 2 * k + 1] =
                    (perm[k] + hash_uniform(sid, 0x10000u + (uint32_t)(off2[a] + 2 * k + 1))) / n;
            }
        }
    }
}

/* ---- camera: PerspectiveCamera.generate_ray_differential (camera/projective.py:178-232);
 * only o and d are produced (ray differentials feed textures only; every supported
 * texture is constant).  Transform.__call__ on Point / Vector: transform/__init__.py:54-64. */
static inline double ne_unity(double x) { return x < 1. - PT_EPS || x > 1. + PT_EPS; }
static v3 xf4_point(const double *m, v3 p) {
    double r0 = (m[0] * p.x + m[1] * p.y + m[2] * p.z) + m[3];
    double r1 = (m[4] * p.x + m[5] * p.y + m[6] * p.z) + m[7];
    double r2 = (m[8] * p.x + m[9] * p.y + m[10] * p.z) + m[11];
    double r3 = (m[12] * p.x + m[13] * p.y + m[14] * p.z) + m[15];
    v3 r = V(r0, r1, r2);
    if (ne_unity(r3)) r = vdiv(r, r3);
    return r;
}

/* concentric_sample_disk, montecarlo/utility.py:122-168 */
static void concentric_sample_disk(double u1, double u2, double *ox, double *oy) {
    double sx = 2. * u1 - 1., sy = 2. * u2 - 1., r, theta;
    if (sx == 0. && sy == 0.) { *ox = 0.; *oy = 0.; return; }
    if (sx > -sy) {
        if (sx > sy) { r = sx; theta = sy / sx; if (sy <= 0.) theta += 8.; }
        else { r = sy; theta = 2. - sx / sy; }
    } else {
        if (sx > sy) { r = -sy; theta = 6. + sx / sy; }
        else { r = -sx; theta = 4. - sy / sx; }
    }
    theta *= PT_PI / 4.;
    *ox = r * cos(theta); *oy = r * sin(theta);
}

static void camera_ray(const pt_camera *cam, const double *row, oray *r) {
    v3 p_cam = xf4_point(cam->r2c, V(row[0], row[1], 0.));
    v3 o = V(0, 0, 0);
    v3 d = vnormalize(p_cam);
    if (cam->kind == PT_CAM_ORTHO) {               /* OrthoCamera.generate_ray, camera/projective.py:80-108 */
        o = p_cam;
        d = V(0., 0., 1.);
    }
    if (cam->lens_rad > 0.) {
        double lu, lv;
        concentric_sample_disk(row[2], row[3], &lu, &lv);
        lu *= cam->lens_rad; lv *= cam->lens_rad;
        double ft = cam->focal_dist / d.z;
        v3 Pfoc = vadd(o, vmul(d, ft));
        o = V(lu, lv, 0.);
        d = vnormalize(vsub(Pfoc, o));
    }
    r->o = xf4_point(cam->c2w, o);
    r->d = xf_vec(cam->c2w, d);
    if (cam->kind == PT_CAM_ORTHO) {
        /* OrthoCamera.generate_ray_differential (projective.py:110-125) transforms the ray that
         * generate_ray already took to world space by c2w a SECOND time (Q14) */
        r->o = xf4_point(cam->c2w, r->o);
        r->d = xf_vec(cam->c2w, r->d);
    }
    r->mint = 0.; r->maxt = PT_INF;
}

/* ----------------------------------------------------------------------------------
 * integrators
 * -------------------------------------------------------------------------------- */

/* DirectLightingIntegrator.li (integrator/surface/direct.py:70-105) with SAMPLE_ALL_UNIFORM
 * (the only strategy that runs, Q10) written as a loop over the specular-reflection chain:
 * specular_transmit contributes exactly 0 (it unpacks the transmittance slot, Q5) and its
 * subtree is therefore not traced; specular_reflect (integrator/utility.py:120-155) recurses
 * through renderer.li (renderer.py:58-75), which adds Σ light.le on a miss.  The sample
 * tables are the SAME at every recursion depth (the offsets index one Sample). */
static v3 li_direct(octx *c, const pt_integrator_desc *ig, const olayout *Lo, const double *row,
                    oray ray, ohit hit) {
    const oscene *s = c->s;
    v3 L = V(0, 0, 0), T = V(1, 1, 1);
    for (int depth = 0;; depth++) {
        overtex vx;
        make_vertex(s, &ray, &hit, &vx);
        v3 wo = vneg(ray.d);
        /* isect.le(wo): no primitive of a supported scene carries an area light → 0 */
        v3 Lsum = V(0, 0, 0);
        for (int i = 0; i < s->n_lights; i++) {          /* uniform_sample_all_lights, utility.py:45-77 */
            int ns = Lo->d_ns[i];
            v3 Ld = V(0, 0, 0);
            for (int j = 0; j < ns; j++)
                Ld = vadd(Ld, estimate_direct(c, &s->lights[i], &vx, wo, row[Lo->d_lpos[i] + 2 * j],
                                              row[Lo->d_lpos[i] + 2 * j + 1]));
            Lsum = vadd(Lsum, vdiv(Ld, (double)ns));
        }
        L = vadd(L, vmulv(T, Lsum));
        if (!(depth + 1 < ig->max_depth)) break;
        /* specular_reflect: BSDF.sample_f(wo, rng-sample, REFLECTION|SPECULAR) */
        v3 wi, f; unsigned tp;
        /* the rng() draws only choose among >= 2 specular-reflection lobes; a material has at most one */
        double pdf = bsdf_sample_f(&vx, wo, 0., 0., 0., BDF_REFLECTION | BDF_SPECULAR, &wi, &tp, &f);
        if (!(pdf > 0. && !util_is_black(f) && fabs(vdot(wi, vx.nn)) != 0.)) break;
        T = vmulv(T, vdiv(vmul(f, fabs(vdot(wi, vx.nn))), pdf));
        ray.o = vx.p; ray.d = wi; ray.mint = vx.rEps; ray.maxt = PT_INF;
        if (!closest(c, &ray, &hit)) { L = vadd(L, vmulv(T, lights_le(s))); break; }
    }
    return L;
}

/* WhittedIntegrator.li (integrator/surface/whitted.py:26-59) as a loop over the specular-reflection
 * chain, like li_direct.  Every light is sampled once with LightSample.from_rand(rng) (3 draws, the
 * third unused), no MIS; specular_reflect / specular_transmit each draw 3 more (integrator/
 * utility.py:128,164).  Along the reflect chain — the only part that contributes (Q5) — the vertex
 * at depth d therefore starts at draw d * (3 nL + 3).  bsdf.f runs with the default flags (ALL). */
static v3 li_whitted(octx *c, const pt_integrator_desc *ig, uint32_t sid, oray ray, ohit hit) {
    const oscene *s = c->s;
    v3 L = V(0, 0, 0), T = V(1, 1, 1);
    for (int depth = 0;; depth++) {
        overtex vx;
        make_vertex(s, &ray, &hit, &vx);
        v3 wo = vneg(ray.d);
        uint32_t k = (uint32_t)depth * (3u * (uint32_t)s->n_lights + 3u);
        v3 Lsum = V(0, 0, 0);
        for (int i = 0; i < s->n_lights; i++, k += 3) {
            olsample ls;
            light_sample_l(&s->lights[i], vx.p, vx.rEps, hash_uniform(sid, k), hash_uniform(sid, k + 1), &ls);
            if (spec_is_black(ls.Li) || ls.pdf == 0.) continue;
            v3 f = bsdf_f(&vx, wo, ls.wi, BDF_ALL);
            if (!spec_is_black(f) && !occluded(c, &ls.shadow))
                Lsum = vadd(Lsum, vdiv(vmul(vmul(vmulv(f, ls.Li), fabs(vdot(ls.wi, vx.nn))), 1.), ls.pdf));
        }
        L = vadd(L, vmulv(T, Lsum));
        if (!(depth + 1 < ig->max_depth)) break;
        v3 wi, f; unsigned tp;
        double pdf = bsdf_sample_f(&vx, wo, 0., 0., 0., BDF_REFLECTION | BDF_SPECULAR, &wi, &tp, &f);
        if (!(pdf > 0. && !util_is_black(f) && fabs(vdot(wi, vx.nn)) != 0.)) break;
        T = vmulv(T, vdiv(vmul(f, fabs(vdot(wi, vx.nn))), pdf));
        ray.o = vx.p; ray.d = wi; ray.mint = vx.rEps; ray.maxt = PT_INF;
        if (!closest(c, &ray, &hit)) { L = vadd(L, vmulv(T, lights_le(s))); break; }
    }
    return L;
}

/* PathIntegrator.li AS IT EXECUTES (integrator/surface/path.py:44-144): `specular_bounce`
 * is always True after the first sample_f (Q4), so every escaped continuation ray adds
 * T·Σ light.le; emitted terms at vertices are 0 (no emissive primitives); transmittance ≡ 1. */
static v3 li_path(octx *c, const pt_integrator_desc *ig, const olayout *Lo, const double *row, uint32_t sid,
                  oray ray, ohit hit) {
    const oscene *s = c->s;
    v3 T = V(1, 1, 1), L = V(0, 0, 0);
    uint32_t k = 0;                                   /* rng() draw counter */
    for (int b = 0;; b++) {
        overtex vx;
        make_vertex(s, &ray, &hit, &vx);
        v3 wo = vneg(ray.d);
        /* uniform_sample_one_light, integrator/utility.py:80-117 */
        if (s->n_lights > 0) {
            double ulnum, ul0, ul1;
            if (b < 3) { ulnum = row[Lo->p_lnum[b]]; ul0 = row[Lo->p_lpos[b]]; ul1 = row[Lo->p_lpos[b] + 1]; }
            else {
                ulnum = hash_uniform(sid, k++);
                ul0 = hash_uniform(sid, k++); ul1 = hash_uniform(sid, k++); k++;   /* LightSample.from_rand */
                k += 3;                                                            /* BSDFSample.from_rand */
            }
            int ln = (int)floor(ulnum * s->n_lights);
            if (ln > s->n_lights - 1) ln = s->n_lights - 1;
            v3 Ld = estimate_direct(c, &s->lights[ln], &vx, wo, ul0, ul1);
            L = vadd(L, vmulv(T, vmul(Ld, (double)s->n_lights)));
        }
        double u0, u1, uc;
        if (b < 3) { u0 = row[Lo->p_pdir[b]]; u1 = row[Lo->p_pdir[b] + 1]; uc = row[Lo->p_pcom[b]]; }
        else { u0 = hash_uniform(sid, k++); u1 = hash_uniform(sid, k++); uc = hash_uniform(sid, k++); }
        v3 wi, f; unsigned tp;
        double pdf = bsdf_sample_f(&vx, wo, u0, u1, uc, BDF_ALL, &wi, &tp, &f);
        if (spec_is_black(f) || pdf == 0.) break;
        T = vmulv(T, vdiv(vmul(f, fabs(vdot(wi, vx.nn))), pdf));
        ray.o = vx.p; ray.d = wi; ray.mint = vx.rEps; ray.maxt = PT_INF;
        if (b > 3) {
            double ty = spec_y(T);
            double cont_prob = ty < .5 ? ty : .5;                  /* min(.5, T.y()) */
            if (hash_uniform(sid, k++) > cont_prob) break;
            T = vdiv(T, cont_prob);
        }
        if (b == ig->max_depth) break;
        if (!closest(c, &ray, &hit)) { L = vadd(L, vmulv(T, lights_le(s))); break; }
    }
    return L;
}

/* ----------------------------------------------------------------------------------
 * film: ImageFilm.add_sample, film/film.py:137-198 (film_accum[y][x][4] = Lxyz, weight_sum)
 * -------------------------------------------------------------------------------- */
static void film_add_sample(const pt_film_desc *fm, double imageX, double imageY, v3 L, double *acc) {
    double dX = imageX - .5, dY = imageY - .5;
    int x0 = (int)ceil(dX - fm->xw), x1 = (int)floor(dX + fm->xw);
    int y0 = (int)ceil(dY - fm->yw), y1 = (int)floor(dY + fm->yw);
    if (x0 < fm->x_start) x0 = fm->x_start;
    if (x1 > fm->x_start + fm->x_cnt - 1) x1 = fm->x_start + fm->x_cnt - 1;
    if (y0 < fm->y_start) y0 = fm->y_start;
    if (y1 > fm->y_start + fm->y_cnt - 1) y1 = fm->y_start + fm->y_cnt - 1;
    if (x1 - x0 < 0 || y1 - y0 < 0) return;
    /* rgb2xyz, spectral/spectrum.py:31-34 */
    double xyz[3] = {(0.412453 * L.x + 0.357580 * L.y) + 0.180423 * L.z,
                     (0.212671 * L.x + 0.715160 * L.y) + 0.072169 * L.z,
                     (0.019334 * L.x + 0.119193 * L.y) + 0.950227 * L.z};
    double xwInv = 1. / fm->xw, ywInv = 1. / fm->yw;
    for (int y = y0; y <= y1; y++) {
        int ify = (int)floor(fabs((y - dY) * 16 * ywInv));
        if (ify > 15) ify = 15;
        for (int x = x0; x <= x1; x++) {
            int ifx = (int)floor(fabs((x - dX) * 16 * xwInv));
            if (ifx > 15) ifx = 15;
            double wt = fm->table[ify * 16 + ifx];
            double *px = acc + 4 * ((size_t)(y - fm->y_start) * fm->x_cnt + (x - fm->x_start));
            px[0] += wt * xyz[0]; px[1] += wt * xyz[1]; px[2] += wt * xyz[2]; px[3] += wt;
        }
    }
}

/* ImageFilm.write_image pixel loop, film/film.py:243-256 (xyz2rgb spectrum.py:25-28) */
int pto_film_resolve(const double *acc, int32_t x_cnt, int32_t y_cnt, double *rgb) {
    for (size_t i = 0; i < (size_t)x_cnt * y_cnt; i++) {
        const double *a = acc + 4 * i;
        double r = (3.240479 * a[0] - 1.537150 * a[1]) - 0.498535 * a[2];
        double g = (-0.969256 * a[0] + 1.875991 * a[1]) + 0.041556 * a[2];
        double b = (0.055648 * a[0] - 0.204043 * a[1]) + 1.057311 * a[2];
        if (!(a[3] == 0.)) {
            r = r / a[3]; g = g / a[3]; b = b / a[3];
            r = r > 0. ? r : 0.; g = g > 0. ? g : 0.; b = b > 0. ? b : 0.;   /* np.maximum(0., ·) */
        }
        rgb[3 * i] = r; rgb[3 * i + 1] = g; rgb[3 * i + 2] = b;
    }
    return PT_OK;
}

/* The array conversion of imageio.write_image (utility/imageio.py:60-65): per channel
 * coef = np.max(rgb[:,:,c]) (NaN propagates), `if not coef == 0.: rgb = (255. * rgb) / coef`, then
 * rgb.astype('uint8'), which on x86-64 is the C conversion through a 32-bit integer (low byte kept;
 * out of range / NaN → 0x80000000 → 0). */
int pto_film_to_u8(const double *rgb, int32_t x_cnt, int32_t y_cnt, uint8_t *out) {
    const size_t n = (size_t)x_cnt * y_cnt;
    for (int c = 0; c < 3; c++) {
        double coef = n ? rgb[c] : 0.;
        for (size_t i = 1; i < n; i++) {
            const double v = rgb[3 * i + c];
            if (v != v || v > coef) { if (!(coef != coef)) coef = v; }
        }
        for (size_t i = 0; i < n; i++) {
            double v = rgb[3 * i + c];
            if (!(coef == 0.)) v = (255. * v) / coef;
            int32_t q = (v > -2147483649. && v < 2147483648.) ? (int32_t)v : (int32_t)0x80000000u;
            out[3 * i + c] = (uint8_t)(q & 255);
        }
    }
    return PT_OK;
}

/* ----------------------------------------------------------------------------------
 * SamplerRendererTask.__call__ (renderer/renderer.py:132-198): same argument meaning as
 * pt_render, host pointers.
 * -------------------------------------------------------------------------------- */
int pto_render(void *scene, const pt_camera *cam, const pt_film_desc *film, const pt_sampler_desc *sp,
               const pt_integrator_desc *ig, const pt_tile *tile, const double *sample_table,
               int32_t table_dims, double *film_accum, double *sample_L, pt_stats *stats) {
    const oscene *s = (const oscene *)scene;
    if (!s || !cam || !film || !sp || !ig || !tile) return PT_ERR_ARG;
    olayout Lo;
    int rc = make_layout(s, ig, &Lo);
    if (rc != PT_OK) return rc;
    if (sample_table && table_dims != Lo.dims) return PT_ERR_ARG;
    if (Lo.dims > 512) return PT_ERR_CAPACITY;
    int spp = sp->xs * sp->ys;
    int W = sp->x_end - sp->x_start;
    octx c; c.s = s; c.st = stats;
    double rowbuf[512];
    int64_t out_i = 0;
    const int y_step = tile->y_step > 1 ? tile->y_step : 1;      /* interleaved rows (include/pytracer_hip.h pt_tile) */
    for (int py = tile->y0 + (y_step > 1 ? tile->y_phase : 0); py < tile->y1; py += y_step)
        for (int px = tile->x0; px < tile->x1; px++)
            for (int si = tile->s0; si < tile->s1; si++, out_i++) {
                uint32_t sid = (uint32_t)((((int64_t)(py - sp->y_start) * W + (px - sp->x_start)) * spp) + si);
                const double *row;
                if (sample_table) row = sample_table + (size_t)sid * Lo.dims;
                else { gen_sample_row(sp, &Lo, ig, px, py, si, sid, rowbuf); row = rowbuf; }
                oray ray;
                camera_ray(cam, row, &ray);
                if (stats) stats->camera_samples++;
                v3 L = V(0, 0, 0);
                int hit = 1;
                if (ig->pretest) hit = occluded(&c, &ray);           /* renderer.py:168 */
                ohit h;
                if (hit && closest(&c, &ray, &h)) {                   /* renderer.py:64 */
                    const uint32_t rsid = sample_table ? sid : mix_seed(sid, sp->seed);
                    if (ig->kind == PT_INTEG_DIRECT) L = li_direct(&c, ig, &Lo, row, ray, h);
                    else if (ig->kind == PT_INTEG_WHITTED) L = li_whitted(&c, ig, rsid, ray, h);
                    else L = li_path(&c, ig, &Lo, row, rsid, ray, h);
                }
                /* guards, renderer.py:185-193 */
                if (isnan(L.x) || isnan(L.y) || isnan(L.z)) { L = V(0, 0, 0); if (stats) stats->guard_nan++; }
                else if (spec_y(L) < -PT_EPS) { L = V(0, 0, 0); if (stats) stats->guard_neg++; }
                else if (spec_y(L) == PT_INF) { L = V(0, 0, 0); if (stats) stats->guard_inf++; }
                if (sample_L) { sample_L[3 * out_i] = L.x; sample_L[3 * out_i + 1] = L.y; sample_L[3 * out_i + 2] = L.z; }
                if (film_accum) film_add_sample(film, row[0], row[1], L, film_accum);
            }
    return PT_OK;
}

/* ---- unit-level entry points for the golden-log checks -------------------------------- */

/* camera rays for table rows: out[n][6] = o, d */
int pto_camera_rays(const pt_camera *cam, const double *table, int32_t dims, int64_t n, double *out) {
    for (int64_t i = 0; i < n; i++) {
        oray r;
        camera_ray(cam, table + i * dims, &r);
        out[6 * i] = r.o.x; out[6 * i + 1] = r.o.y; out[6 * i + 2] = r.o.z;
        out[6 * i + 3] = r.d.x; out[6 * i + 4] = r.d.y; out[6 * i + 5] = r.d.z;
    }
    return PT_OK;
}

/* hit geometry of a closest hit: out[15] = p, nn, dpdu, dpdv, u, v, rEps */
int pto_hit_geometry(void *scene, const double *ray8, int32_t prim, double t, double b1, double b2, double *out) {
    const oscene *s = (const oscene *)scene;
    oray r; r.o = V(ray8[0], ray8[1], ray8[2]); r.d = V(ray8[3], ray8[4], ray8[5]); r.mint = ray8[6]; r.maxt = t;
    ohit h; h.prim = prim; h.t = t; h.b1 = b1; h.b2 = b2;
    overtex vx; make_vertex(s, &r, &h, &vx);
    double o[15] = {vx.p.x, vx.p.y, vx.p.z, vx.nn.x, vx.nn.y, vx.nn.z, vx.dpdu.x, vx.dpdu.y, vx.dpdu.z,
                    vx.dpdv.x, vx.dpdv.y, vx.dpdv.z, vx.u, vx.v, vx.rEps};
    memcpy(out, o, sizeof(o));
    return PT_OK;
}

/* light.sample_l: out[18] = Li, wi, pdf, shadow o, d, mint, maxt */
/* One texture at one point: out[3] = tex_eval (textures 0-based; p = the shading point, (u, v) its surface
 * coordinates).  For tests that compare with the reference's Texture.__call__. */
int pto_texture_eval(void *scene, int32_t tex, const double *p, double u, double v, double *out) {
    const oscene *s = (const oscene *)scene;
    if (!s || tex < 0 || tex >= s->n_textures) return PT_ERR_ARG;
    overtex vx;
    memset(&vx, 0, sizeof(vx));
    vx.p = V(p[0], p[1], p[2]);
    vx.u = u; vx.v = v;
    v3 c = tex_eval(s->textures + tex, &vx);
    out[0] = c.x; out[1] = c.y; out[2] = c.z;
    return PT_OK;
}

int pto_sample_l(void *scene, int32_t light, const double *p, double pEps, double u1, double u2, double *out) {
    const oscene *s = (const oscene *)scene;
    olsample ls;
    light_sample_l(&s->lights[light], V(p[0], p[1], p[2]), pEps, u1, u2, &ls);
    double o[18] = {ls.Li.x, ls.Li.y, ls.Li.z, ls.wi.x, ls.wi.y, ls.wi.z, ls.pdf,
                    ls.shadow.o.x, ls.shadow.o.y, ls.shadow.o.z, ls.shadow.d.x, ls.shadow.d.y, ls.shadow.d.z,
                    ls.shadow.mint, ls.shadow.maxt, 0, 0, 0};
    memcpy(out, o, sizeof(o));
    return PT_OK;
}

/* one lobe of a material, in its local frame: out[11] = f(wo,wi)[3], pdf(wo,wi), then sample_f(wo,u1,u2) = pdf, wi[3], f[3] */
int pto_lobe_eval(void *scene, int32_t mat, int32_t lobe, const double *wo, const double *wi, double u1, double u2,
                  double *out) {
    const oscene *s = (const oscene *)scene;
    if (mat < 0 || mat >= s->n_materials || lobe < 0 || lobe >= s->materials[mat].n_lobes) return PT_ERR_ARG;
    const pt_lobe *l = &s->materials[mat].lobes[lobe];
    v3 o = V(wo[0], wo[1], wo[2]), i = V(wi[0], wi[1], wi[2]), swi = V(0, 0, 0), sf = V(0, 0, 0);
    v3 f = lobe_f(l, o, i);
    double pdf = lobe_pdf(l, o, i);
    double spdf = lobe_sample_f(l, o, u1, u2, &swi, &sf);
    double r[11] = {f.x, f.y, f.z, pdf, spdf, swi.x, swi.y, swi.z, sf.x, sf.y, sf.z};
    memcpy(out, r, sizeof(r));
    return PT_OK;
}
