// pt_device.h — device-side value types and the per-vertex shading math of the HIP back end.
//
// gfx950 only.  All arithmetic is binary64 and this translation unit is compiled with
// -ffp-contract=off: the reference computes in numpy.float64 and hit/miss decisions must
// reproduce it bit-exactly (include/pytracer_hip.h), so expressions are evaluated as
// written, in the reference's operation order.  Each function cites the reference
// file:line it restates (paths relative to the reference root).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/pytracer_hip.h"

#define PT_EPS 1e-5
#define PT_PI 3.14159265358979323846
#define PT_INV_PI (1.0 / PT_PI)
#define PT_INV_2PI (1.0 / (2.0 * PT_PI))

// BDFType bits, reflection/bdf/bdf.py:60-70
#define BDF_REFLECTION 1u
#define BDF_TRANSMISSION 2u
#define BDF_DIFFUSE 4u
#define BDF_GLOSSY 8u
#define BDF_SPECULAR 16u
#define BDF_ALL 31u

template <typename R> struct v3 {
    R x, y, z;
};
typedef v3<double> d3;

template <typename R> __device__ __forceinline__ v3<R> mkv(R x, R y, R z) { v3<R> r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ d3 mk3(double x, double y, double z) { return mkv<double>(x, y, z); }
template <typename R, typename S> __device__ __forceinline__ v3<R> cvt3(v3<S> a) { return mkv<R>((R)a.x, (R)a.y, (R)a.z); }
template <typename R> __device__ __forceinline__ v3<R> operator+(v3<R> a, v3<R> b) { return mkv<R>(a.x + b.x, a.y + b.y, a.z + b.z); }
template <typename R> __device__ __forceinline__ v3<R> operator-(v3<R> a, v3<R> b) { return mkv<R>(a.x - b.x, a.y - b.y, a.z - b.z); }
template <typename R> __device__ __forceinline__ v3<R> operator-(v3<R> a) { return mkv<R>(-a.x, -a.y, -a.z); }
template <typename R> __device__ __forceinline__ v3<R> operator*(v3<R> a, R s) { return mkv<R>(a.x * s, a.y * s, a.z * s); }
template <typename R> __device__ __forceinline__ v3<R> operator*(v3<R> a, v3<R> b) { return mkv<R>(a.x * b.x, a.y * b.y, a.z * b.z); }
__device__ __forceinline__ d3 operator/(d3 a, double s) { return mk3(a.x / s, a.y / s, a.z / s); }
__device__ __forceinline__ v3<float> operator/(v3<float> a, float s) {       // production variant: one reciprocal (1 ulp)
    const float r = __frcp_rn(s);
    return mkv<float>(a.x * r, a.y * r, a.z * r);
}
// ndarray.dot on 3-vectors: left-to-right accumulation
template <typename R> __device__ __forceinline__ R dot3(v3<R> a, v3<R> b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
// Vector.cross, geometry/__init__.py:79-82
template <typename R> __device__ __forceinline__ v3<R> cross3(v3<R> a, v3<R> b) {
    return mkv<R>(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
// sq_length / length, geometry/__init__.py:84-88
template <typename R> __device__ __forceinline__ R sqlen3(v3<R> a) { return a.x * a.x + a.y * a.y + a.z * a.z; }
__device__ __forceinline__ double len3(d3 a) { return sqrt(sqlen3(a)); }
__device__ __forceinline__ float len3(v3<float> a) { return sqrtf(sqlen3(a)); }
// geometry/utility.py:55-73
template <typename R> __device__ __forceinline__ v3<R> normalize3(v3<R> a) {
    R l = len3(a);
    if (!(l == R(0))) return a / l;
    return a;
}
template <typename R> __device__ __forceinline__ R clipd(R x, R lo, R hi) { return x < lo ? lo : (x > hi ? hi : x); }
template <typename R> __device__ __forceinline__ R max0(R x) { return x > R(0) ? x : R(0); }

// counter-based uniform in [0,1): splitmix64 finaliser of (sid, k); DESIGN.md "sample streams"
__device__ __forceinline__ double hash_uniform(uint32_t sid, uint32_t k) {
    uint64_t z = ((uint64_t)sid << 32) | (uint64_t)k;
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
    return (double)(z >> 11) * (1.0 / 9007199254740992.0);
}
__device__ __forceinline__ uint32_t mix_seed(uint32_t sid, uint32_t seed) { return sid + seed * 0x9E3779B1u; }

// ---------------------------------------------------------------------------------------
// device scene layout (HBM).  Nodes and intersection triangles are 16-byte-aligned AoS
// records fetched with dwordx4 loads by the one lane that walks them; everything indexed
// by ray / path slot is component-major SoA so that a wave's 64 lanes read 512 contiguous
// bytes per component.
// ---------------------------------------------------------------------------------------
struct __attribute__((aligned(16))) GNode {   // 64 B; bvh.py:60-66 _LinearNode
    double bmin[3];
    double bmax[3];
    int32_t offset;      // interior: second child; leaf: first primitive
    int32_t nprim_axis;  // n_prim | axis << 16
    double _pad;
};

struct __attribute__((aligned(16))) GTri {    // 80 B; p1 and the two edges of triangle.py:183-184
    double p1[3];
    double e1[3];        // p2 - p1 (same subtraction, done once on the host in binary64)
    double e2[3];        // p3 - p1
    int32_t leaf_nprim;  // on the FIRST triangle of a leaf: that leaf's n_prim and node index (0 elsewhere); the
    int32_t leaf_node;   // children-pair walk reaches a leaf by its first primitive alone (GPair::ref)
};

// Children-pair record of an interior node (k_trace_pair): ONE 128-byte line brings the boxes of both children,
// so a step of the walk tests two boxes per dependent fetch instead of one (the visit ORDER, every comparison and
// the census stay those of bvh.py:322-410: the far child's verdict is finished when it is popped).
struct __attribute__((aligned(128))) GPair {
    double b0[6];        // first child (node self + 1): bmin xyz, bmax xyz
    double b1[6];        // second child (node nodes[self].offset)
    int32_t ref0, ref1;  // child reference: >= 0 pair index of an interior child; < 0: ~(first primitive) of a leaf
    int32_t axis;        // split axis of THIS node (dir_is_neg[node.axis], bvh.py:350)
    int32_t self;        // this node's index in DScene::nodes (the exact re-test of a popped entry reads its box there)
    int32_t node0, node1;// node indices of the two children
    int32_t _pad[2];
};
static_assert(sizeof(GPair) == 128, "one cache line");

// The same record at half the size (k_trace_pair_refill_h): the child boxes as binary32 OUTER boxes (minima rounded towards
// -inf, maxima towards +inf) and `w`, an upper bound of the distance any of the twelve planes moved.  The walk decides a
// box test on the outer box when it can prove that the reference's binary64 test on the exact box gives the same answer
// (a missed outer box is a missed box; a hit whose margins exceed what w can change is a hit) and fetches the exact GPair
// otherwise — so results and census stay bit-identical while a step moves 64 bytes instead of 128.
struct __attribute__((aligned(64))) GPairH {
    float b0[6];         // first child: bmin xyz (rounded down), bmax xyz (rounded up)
    float b1[6];         // second child
    int32_t ref0, ref1;  // as in GPair
    int32_t axis;
    float w;             // >= |outer plane - exact plane| for all twelve planes (rounded up; inf: always fetch the exact record)
};
static_assert(sizeof(GPairH) == 64, "half a cache line");

// Top-of-tree record staged in LDS by k_trace_top (pt_kernels.hip): a GNode whose padding holds
// the references of its two children.  ref >= 0: index into DScene::nodes; ref < 0: LDS slot ~ref.
struct __attribute__((aligned(16))) TopNode {   // 64 B
    double bmin[3];
    double bmax[3];
    int32_t offset;
    int32_t nprim_axis;
    int32_t ref0, ref1;  // first (idx+1) and second (offset) child
};

// One measured BRDF table (PT_LOBE_MEASURED_IRISO): the samples sorted into a uniform grid over their bounding
// box, x fastest, so the ball queries of IrIsotropicBRDF.f (reflection/bdf/measured.py:51-66; a scipy cKDTree
// in the reference) scan the cells a ball overlaps.  Built by pt_scene_upload as ONE position-independent blob
// ([BrdfGrid x tables][p, v][cell starts][start rounds], offsets in bytes from the blob's first byte), which
// k_shade<KIND, 2> copies into LDS when it fits (DScene::brdf_lds_bytes).
struct BrdfGrid {
    double bmin[3], inv[3];    // cell of a point = clamp(floor((p - bmin) * inv), 0, g - 1) per axis
    double cw[3];              // cell widths (0 on an axis where all samples coincide)
    double radius[16];         // .01 * 1.414^k as the reference's loop produces them (radius[15] is the first > 1.5)
    int32_t g[3], n;
    int32_t off_cells;         // int32 [g0*g1*g2 + 1] first sample of every cell
    int32_t off_kguess;        // int32 [g0*g1*g2] the round at which a query at the cell's centre ends (a guess, not a bound)
    int32_t off_p, off_v;      // double [n][3] sample positions (brdf_remap space) and RGB values, in cell order
};

struct DScene {
    const GNode *nodes;
    const GPair *pairs;        // one record per interior node, numbered in depth-first order (root = 0), (k_tri_frames at upload)
    const GPairH *pairs_h;     // the same records with binary32 outer boxes (nullptr: not built)
    float pair_w_max;          // the largest GPairH::w of the scene
    const TopNode *top;        // the first n_top nodes in breadth-first order, (k_tri_frames at upload)
    int32_t n_top;
    int32_t exact_slab;        // a node box with min > max (or NaN): only the reference's own comparison order is exact
    const double2 *node_box;   // [n_nodes][3] the boxes alone at 48-byte stride (bmin.xy | bmin.z bmax.x | bmax.yz), or null
    const int2 *node_meta;     // [n_nodes] (offset, n_prim | axis << 16) of every node, packed: what k_trace_meta stages in LDS
    const GTri *tris;
    const double *tri_verts;   // [n_tris][9] p1,p2,p3 for shading geometry
    const double *tri_frames;  // [n_tris][9] nn, sn, tn: the shading frame of every triangle (tri_frame), (k_tri_frames at upload)
    const int32_t *tri_mat;
    const int32_t *tri_flip;
    const pt_material *mats;
    const pt_light *lights;
    const pt_texture *textures;   // referenced by pt_lobe.tex (k > 0: textures[k-1])
    pt_material *tex_scratch;     // one pt_material per shading thread: the textured material at its vertex
    const double *tri_uv;         // [n_tris][6] corner (u,v) of meshes with uvs, or nullptr (triangle.py:125-131)
    const char *brdf;             // blob of the measured lobes' tables (the device copy of such a lobe has a = table index)
    int32_t brdf_lds_bytes;       // size of that blob if k_shade stages it in LDS (it fits in 64 KB), else 0
    int32_t n_nodes, n_tris, n_mats, n_lights;
    double le_sum[3];          // Σ light.le over lights (DiffuseAreaLight.le = emit, light.py:503-505)
};

struct Ray {
    d3 o, d;
    double mint, maxt;
};

// ---------------------------------------------------------------------------------------
// shading vertex: Triangle.intersect tail (shape/triangle.py:209-243, default uvs
// (0,0),(1,0),(1,1) :125-127), DifferentialGeometry.__init__ (geometry/diffgeom.py:21-42),
// BSDF.__init__ (reflection/bsdf.py:78-94).  Shading geometry == geometric (triangle.py:323-326).
// ---------------------------------------------------------------------------------------
struct Vertex {
    d3 p, nn, sn, tn;
    double rEps;
    const pt_material *mat;
    const char *brdf;          // the BrdfGrid blob, in HBM or staged in LDS (read by k_shade<KIND, 2> only)
    bool brdf_in_lds;
    int *hits;                 // this thread's hit list in LDS (measured_f; k_shade<KIND, 2> only)
};

// The shading frame of a triangle: nn, sn, tn of BSDF.__init__ from dpdu, dpdv of Triangle.intersect.  It depends on the
// triangle only (vertices, uvs, orientation flag): pt_scene_upload runs this once per triangle ON THE DEVICE
// (k_tri_frames: the same instructions as a per-hit evaluation would execute, so the same bits) and make_vertex reads the
// 72-byte result instead of 72 bytes of vertices + 8 divisions and 2 square roots per hit.
// EXT: the scene may have mesh uvs; without them the default uvs are compile-time constants (same arithmetic).
template <bool EXT>
__device__ inline void tri_frame(const DScene &sc, int prim, d3 &nn, d3 &sn, d3 &tn) {
    const double *tv = sc.tri_verts + 9 * (size_t)prim;
    d3 p1 = mk3(tv[0], tv[1], tv[2]), p2 = mk3(tv[3], tv[4], tv[5]), p3 = mk3(tv[6], tv[7], tv[8]);
    d3 dpdu, dpdv;
    if (EXT && sc.tri_uv) {
        const double *q = sc.tri_uv + 6 * (size_t)prim;
        const double du1 = q[0] - q[4], du2 = q[2] - q[4], dv1 = q[1] - q[5], dv2 = q[3] - q[5];
        d3 dp1 = p1 - p3, dp2 = p2 - p3;
        const double det = du1 * dv2 - du2 * dv1;
        if (det == 0.) {
            // triangle.py:219-221: _, dpdu, dpdv = coordinate_system(normalize(e2 x e1)), geometry/utility.py:40-52
            d3 v1 = normalize3(cross3(p3 - p1, p2 - p1));
            if (sqrt(dot3(v1, v1)) == 0.) { dpdu = mk3(0., 1., 0.); dpdv = mk3(0., 0., 1.); }
            else {
                if (fabs(v1.x) > fabs(v1.y)) {
                    const double inv_len = 1. / sqrt(v1.x * v1.x + v1.z * v1.z);
                    dpdu = mk3(-v1.z * inv_len, 0., v1.x * inv_len);
                } else {
                    const double inv_len = 1. / sqrt(v1.y * v1.y + v1.z * v1.z);
                    dpdu = mk3(0., -v1.z * inv_len, v1.y * inv_len);
                }
                dpdv = cross3(v1, dpdu);
            }
        } else {
            const double detInv = 1.0 / det;
            dpdu = ((dp1 * dv2) - (dp2 * dv1)) * detInv;
            dpdv = ((dp1 * (-du2)) + (dp2 * du1)) * detInv;
        }
    } else {
        const double du1 = 0. - 1., du2 = 1. - 1., dv1 = 0. - 1., dv2 = 0. - 1.;
        d3 dp1 = p1 - p3, dp2 = p2 - p3;
        double det = du1 * dv2 - du2 * dv1;
        double detInv = 1.0 / det;
        dpdu = ((dp1 * dv2) - (dp2 * dv1)) * detInv;
        dpdv = ((dp1 * (-du2)) + (dp2 * du1)) * detInv;
    }
    nn = normalize3(cross3(dpdu, dpdv));
    if (sc.tri_flip[prim]) nn = nn * -1.0;
    sn = normalize3(dpdu);
    tn = cross3(nn, sn);
}

template <bool EXT>
__device__ inline void make_vertex(const DScene &sc, const Ray &r, int prim, double t, Vertex &vx) {
    const double *f = sc.tri_frames + 9 * (size_t)prim;
    vx.nn = mk3(f[0], f[1], f[2]); vx.sn = mk3(f[3], f[4], f[5]); vx.tn = mk3(f[6], f[7], f[8]);
    vx.p = r.o + r.d * t;
    vx.rEps = 1e-3 * t;
    vx.mat = sc.mats + sc.tri_mat[prim];
    vx.brdf = sc.brdf;
    vx.hits = nullptr;
    vx.brdf_in_lds = false;
}

// ---- textures (texture/texture/__init__.py) over constant children, evaluated where Material.get_bsdf
// evaluates Kd / Ks (material/material.py:131,175-176) on a copy of the shading geometry whose differentials are
// zero (dg_s.copy(), geometry/diffgeom.py:21-42,47-48): Chekcerboard2DTexture's closed-form antialiasing then
// takes its "inside one check" branch (texture/__init__.py:156-166) = point sampling.
__device__ inline d3 xform_point(const double *m, d3 p) {       // Transform.__call__(Point), transform/__init__.py:55-60
    double x = ((m[0] * p.x + m[1] * p.y) + m[2] * p.z) + m[3];
    double y = ((m[4] * p.x + m[5] * p.y) + m[6] * p.z) + m[7];
    double z = ((m[8] * p.x + m[9] * p.y) + m[10] * p.z) + m[11];
    double w = ((m[12] * p.x + m[13] * p.y) + m[14] * p.z) + m[15];
    if (w < 1. - PT_EPS || w > 1. + PT_EPS) { x /= w; y /= w; z /= w; }     // util.ne_unity
    return mk3(x, y, z);
}
__device__ inline void tex_map2d(const pt_texture &t, d3 p, double u, double v, double &s_out, double &t_out) {
    const double *m = t.m;
    if (t.mapping == PT_MAP_UV) {                  // texturemap:45-47
        s_out = m[0] * u + m[2];
        t_out = m[1] * v + m[3];
    } else if (t.mapping == PT_MAP_PLANAR) {       // texturemap:142-145
        s_out = m[6] + dot3(p, mk3(m[0], m[1], m[2]));
        t_out = m[7] + dot3(p, mk3(m[3], m[4], m[5]));
    } else {                                       // PT_MAP_SPHERICAL, texturemap:59-68
        d3 w = normalize3(xform_point(m, p));
        double cz = w.z < -1. ? -1. : (w.z > 1. ? 1. : w.z);
        double theta = acos(cz);                   // geometry/utility.py:106-112
        double ph = atan2(w.y, w.x);
        if (!(ph > 0.)) ph = ph + 2. * PT_PI;      // geometry/utility.py:121-122
        s_out = theta * PT_INV_PI;
        t_out = ph * PT_INV_2PI;
    }
}
// Perlin's reference permutation (Improved Noise, 2002), the table of pytracer/data/noise.py (there doubled to 512
// entries; indices are reduced modulo 256 here, which is the same look-up).
__device__ const unsigned char PT_NOISE_PERM[256] = {
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
// _grad AS WRITTEN, texture/utility.py:104-114: `if h & 1: u = -1` sets u to the constant -1 (not -u)
__device__ __forceinline__ double noise_grad(int x, int y, int z, double dx, double dy, double dz) {
    int h = PT_NOISE_PERM[(PT_NOISE_PERM[(PT_NOISE_PERM[x & 255] + y) & 255] + z) & 255];
    h &= 15;
    double u = (h < 8 || h == 12 || h == 13) ? dx : dy;
    double v = (h < 4 || h == 12 || h == 13) ? dy : dz;
    if (h & 1) u = -1.;
    if (h & 2) v = -v;
    return u + v;
}
__device__ __forceinline__ double pt_lerp(double t, double v1, double v2) { return (1. - t) * v1 + t * v2; }   // util.lerp
// noise() AS WRITTEN, texture/utility.py:121-164 (y1 is interpolated with wz, not wy)
__device__ inline double perlin_noise(double px, double py, double pz) {
    const double fx = floor(px), fy = floor(py), fz = floor(pz);
    const double dx = px - fx, dy = py - fy, dz = pz - fz;
    // util.ftoi then `& (NOISE_PERM_SIZE - 1)` on a Python int: the low 8 bits of the two's complement
    const int xi = (int)((long long)fx & 255), yi = (int)((long long)fy & 255), zi = (int)((long long)fz & 255);
    const double w000 = noise_grad(xi, yi, zi, dx, dy, dz), w100 = noise_grad(xi + 1, yi, zi, dx - 1, dy, dz);
    const double w010 = noise_grad(xi, yi + 1, zi, dx, dy - 1, dz), w110 = noise_grad(xi + 1, yi + 1, zi, dx - 1, dy - 1, dz);
    const double w001 = noise_grad(xi, yi, zi + 1, dx, dy, dz - 1), w101 = noise_grad(xi + 1, yi, zi + 1, dx - 1, dy, dz - 1);
    const double w011 = noise_grad(xi, yi + 1, zi + 1, dx, dy - 1, dz - 1);
    const double w111 = noise_grad(xi + 1, yi + 1, zi + 1, dx - 1, dy - 1, dz - 1);
    const double wx = dx * dx * dx * (10. - 15. * dx + 6. * dx * dx);
    const double wy = dy * dy * dy * (10. - 15. * dy + 6. * dy * dy);
    const double wz = dz * dz * dz * (10. - 15. * dz + 6. * dz * dz);
    const double x00 = pt_lerp(wx, w000, w100), x10 = pt_lerp(wx, w010, w110);
    const double x01 = pt_lerp(wx, w001, w101), x11 = pt_lerp(wx, w011, w111);
    const double y0 = pt_lerp(wy, x00, x10), y1 = pt_lerp(wz, x01, x11);
    return pt_lerp(wz, y0, y1);
}
// turbulence() AS IT EXECUTES for a material's texture look-up, texture/utility.py:78-101: the differentials are zero
// (Q16), so s2 = 0, log2(0) = -inf and foct = min(max_octaves, +inf) = max_octaves; smooth_step(.3, .7, par) clips par
// INTO [.3, .7] (utility.py:43-50), so the partial octave always contributes (weight .216 when par = 0).
__device__ inline double turbulence(d3 p, double omega, double max_octaves) {
    const double foct = max_octaves;
    const double fo = floor(foct);
    const int octaves = fo > 64. ? 64 : (fo > 0. ? (int)fo : 0);      // range(octaves); PT_TEX_WRINKLED allows at most 64
    double sum = 0., l = 1., o = 1.;
    for (int i = 0; i < octaves; i++) {
        sum += o * fabs(perlin_noise(l * p.x, l * p.y, l * p.z));
        l *= 1.99;
        o *= omega;
    }
    const double par = foct - fo;
    const double v = par < .3 ? .3 : (par > .7 ? .7 : par);
    sum += o * (v * v * (3. - 2. * v)) * fabs(perlin_noise(l * p.x, l * p.y, l * p.z));
    return sum;
}

__device__ inline d3 tex_eval(const pt_texture &t, d3 p, double u, double v) {
    double s = 0., tt = 0.;
    switch (t.kind) {
    case PT_TEX_WRINKLED: {                        // WrinkledTexture, texture/__init__.py:252-266
        const double w = t.v[1][0] * turbulence(xform_point(t.m, p), t.v[0][0], t.v[0][1]);   // v[1][0]: a ScaleTexture's constant, else 1
        return mk3(w, w, w);
    }
    case PT_TEX_CHECKER2D:                         // :147-152 / :162-166
        tex_map2d(t, p, u, v, s, tt);
        return ((((long long)floor(s) + (long long)floor(tt)) & 1) == 0) ? mk3(t.v[0][0], t.v[0][1], t.v[0][2])
                                                                          : mk3(t.v[1][0], t.v[1][1], t.v[1][2]);
    case PT_TEX_CHECKER3D: {                       // :191-196
        d3 q = xform_point(t.m, p);
        return ((((long long)floor(q.x) + (long long)floor(q.y) + (long long)floor(q.z)) & 1) == 0)
                   ? mk3(t.v[0][0], t.v[0][1], t.v[0][2]) : mk3(t.v[1][0], t.v[1][1], t.v[1][2]);
    }
    case PT_TEX_BILERP: {                          // :101-104
        tex_map2d(t, p, u, v, s, tt);
        d3 a = mk3(t.v[0][0], t.v[0][1], t.v[0][2]) * ((1. - s) * (1. - tt));
        d3 b = mk3(t.v[1][0], t.v[1][1], t.v[1][2]) * ((1. - s) * tt);
        d3 c = mk3(t.v[2][0], t.v[2][1], t.v[2][2]) * ((1. - tt) * s);
        d3 e = mk3(t.v[3][0], t.v[3][1], t.v[3][2]) * (s * tt);
        return ((a + b) + c) + e;
    }
    default:                                       // PT_TEX_UV :114-116
        tex_map2d(t, p, u, v, s, tt);
        return mk3(s - floor(s), tt - floor(tt), 0.);
    }
}
// A material with textured lobes is evaluated ONCE per vertex into the calling thread's scratch record
// (lobe colour = R * util.clip(texture(dgs)), material.py:131) and the vertex points at that copy: the BSDF code
// reads lobes through vx.mat and never knows.  Materials without textures (tex == 0 in every lobe) pay one test.
__device__ inline void apply_textures(const DScene &sc, Vertex &vx, int prim, double b1, double b2, pt_material *scratch) {
    const pt_material &m = *vx.mat;
    int any = 0;
    for (int k = 0; k < m.n_lobes; k++) any |= m.lobes[k].tex;
    if (!any) return;
    // Triangle.intersect, triangle.py:228-231 with the default uvs (0,0),(1,0),(1,1)
    const double b0 = 1.0 - b1 - b2;
    double u = (b0 * 0. + b1 * 1.) + b2 * 1., v = (b0 * 0. + b1 * 0.) + b2 * 1.;
    if (sc.tri_uv) {
        const double *q = sc.tri_uv + 6 * (size_t)prim;
        u = (b0 * q[0] + b1 * q[2]) + b2 * q[4];
        v = (b0 * q[1] + b1 * q[3]) + b2 * q[5];
    }
    int n = 0;
    for (int k = 0; k < m.n_lobes; k++) {
        pt_lobe l = m.lobes[k];
        const int ti = l.tex & 0xFFFF;
        if (ti) {
            d3 c = tex_eval(sc.textures[ti - 1], vx.p, u, v);
            l.R[0] *= max0(c.x); l.R[1] *= max0(c.y); l.R[2] *= max0(c.z);
            // `if not k.is_black()` (material.py:217-219,289-295,343-362): the lobe is not pushed at this hit
            if ((l.tex & PT_LOBE_DROP_IF_BLACK) && l.R[0] < PT_EPS && l.R[1] < PT_EPS && l.R[2] < PT_EPS) continue;
        }
        const int pi = PT_LOBE_PARAM_TEX(l.tex);
        if (pi) {
            // the lobe's float parameter from a FLOAT texture, per hit
            const double f = tex_eval(sc.textures[pi - 1], vx.p, u, v).x;
            if (l.type == PT_LOBE_OREN_NAYAR) {
                // MatteMaterial, material.py:132-139: sigma = np.clip(self.sigma(dgs), 0., 90.); Lambertian when it is 0,
                // else OrenNayar.__init__ (orennayar.py:36-42)
                const double sigma = f < 0. ? 0. : (f > 90. ? 90. : f);      // NaN stays NaN, as in np.clip
                if (sigma == 0.) { l.type = PT_LOBE_LAMBERT; l.a = 0.; l.b = 0.; }
                else {
                    const double sig = sigma * (PT_PI / 180.);               // np.deg2rad
                    const double sig_sq = sig * sig;
                    l.a = 1. - (sig_sq / (2. * (sig_sq + 0.33)));
                    l.b = .45 * sig_sq / (sig_sq + .09);
                }
            } else {
                // Plastic / Uber, material.py:178-179,349-350: Blinn(1. / rough); Blinn.__init__ clips at 10000 (torrance.py:66-67)
                const double e = 1. / f;
                l.c = e > 10000. ? 10000. : e;
            }
        }
        l.tex = 0;
        scratch->lobes[n++] = l;
    }
    scratch->n_lobes = n;
    vx.mat = scratch;
}

// ---------------------------------------------------------------------------------------
// BSDF and light arithmetic, generic in the real type R.  R = double restates the reference operation for
// operation (the parity kernels); R = float is the production-only fast variant (k_shade<KIND, EXT, float>):
// directions, colours, pdfs and the transcendental functions in binary32, while POINTS — the vertex, the point
// sampled on a light, everything a ray is built from — and every discrete decision that geometry feeds (inside /
// outside the light sphere, the sphere intersection) stay binary64, as do throughput, radiance and the film.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ double r_sqrt(double x) { return sqrt(x); }
__device__ __forceinline__ float r_sqrt(float x) { return sqrtf(x); }
__device__ __forceinline__ double r_abs(double x) { return fabs(x); }
__device__ __forceinline__ float r_abs(float x) { return fabsf(x); }
__device__ __forceinline__ double r_pow(double x, double y) { return pow(x, y); }
__device__ __forceinline__ float r_pow(float x, float y) { return __powf(x, y); }          // exp2(y log2 x): ~2e-6 relative here
__device__ __forceinline__ void r_sincos(double x, double *s, double *c) { sincos(x, s, c); }
__device__ __forceinline__ void r_sincos(float x, float *s, float *c) { __sincosf(x, s, c); }   // arguments in [0, 2 pi]
// binary64: IEEE division, as numpy divides.  binary32 (production variant): multiplication by v_rcp_f32 (1 ulp)
__device__ __forceinline__ double r_div(double a, double b) { return a / b; }
__device__ __forceinline__ float r_div(float a, float b) { return a * __frcp_rn(b); }
__device__ __forceinline__ double r_acos(double x) { return acos(x); }
__device__ __forceinline__ float r_acos(float x) { return acosf(x); }
__device__ __forceinline__ double r_sin(double x) { return sin(x); }
__device__ __forceinline__ float r_sin(float x) { return sinf(x); }
__device__ __forceinline__ double r_cos(double x) { return cos(x); }
__device__ __forceinline__ float r_cos(float x) { return cosf(x); }

template <typename R> struct VertexT {
    d3 p;                      // the shading point: always binary64
    v3<R> nn, sn, tn;
    double rEps;
    const pt_material *mat;
    const char *brdf;          // the BrdfGrid blob, in HBM or staged in LDS (read by k_shade<KIND, 2> only)
    bool brdf_in_lds;
    int *hits;                 // this thread's hit list in LDS (measured_f; k_shade<KIND, 2> only)
};
template <typename R>
__device__ __forceinline__ VertexT<R> vertex_as(const Vertex &vx) {
    VertexT<R> v;
    v.p = vx.p;
    v.nn = cvt3<R>(vx.nn); v.sn = cvt3<R>(vx.sn); v.tn = cvt3<R>(vx.tn);
    v.rEps = vx.rEps; v.mat = vx.mat; v.brdf = vx.brdf; v.brdf_in_lds = vx.brdf_in_lds; v.hits = vx.hits;
    return v;
}

// BSDF.w2l, bsdf.py:187-194
template <typename R>
__device__ __forceinline__ v3<R> w2l(const VertexT<R> &vx, v3<R> v) { return mkv<R>(dot3(v, vx.sn), dot3(v, vx.tn), dot3(v, vx.nn)); }
// BSDF.l2w AS WRITTEN, bsdf.py:196-205 (each basis vector scaled by the same component)
template <typename R>
__device__ __forceinline__ v3<R> l2w(const VertexT<R> &vx, v3<R> v) {
    return mkv<R>(vx.sn.x * v.x + vx.tn.x * v.x + vx.nn.x * v.x, vx.sn.y * v.y + vx.tn.y * v.y + vx.nn.y * v.y,
                  vx.sn.z * v.z + vx.tn.z * v.z + vx.nn.z * v.z);
}

__device__ __forceinline__ unsigned lobe_flags(int type) {
    // Lambertian/OrenNayar: REFLECTION|DIFFUSE; Microfacet: REFLECTION|GLOSSY;
    // SpecularReflection: REFLECTION|SPECULAR; SpecularTransmission: TRANSMISSION|SPECULAR
    // (the conductor microfacet lobe, type 5, is REFLECTION|GLOSSY like type 2)
    // (so is the measured lobe, type 6: measured.py:47)
    const unsigned long long tab = (5ull) | (5ull << 5) | (9ull << 10) | (17ull << 15) | (18ull << 20) | (9ull << 25) | (9ull << 30);
    return (unsigned)(tab >> (5 * type)) & 31u;
}
// BDF.match_flag, bdf.py:166-170
__device__ __forceinline__ bool match_flag(int type, unsigned flags) {
    unsigned t = lobe_flags(type);
    return (t & flags) == t;
}

// fr_diel AS WRITTEN, reflection/utility.py:46-57 (Q8)
template <typename R>
__device__ __forceinline__ R fr_diel(R cos_i, R cos_t, R eta_i, R eta_t) {
    R r_para = ((eta_t * cos_i) - (eta_i * cos_t)) / ((eta_t * cos_i) + (eta_t * cos_t));
    R r_perp = ((eta_i * cos_i) - (eta_t * cos_t)) / ((eta_i * cos_i) + (eta_t * cos_t));
    return R(.5) * (r_para * r_para + r_perp * r_perp);
}
// FresnelDielectric.__call__, reflection/fresnel.py:54-72
template <typename R>
__device__ inline R fresnel_dielectric(R cosi, R eta_i, R eta_t) {
    R ci = clipd(cosi, R(-1), R(1));
    R ei = eta_i, et = eta_t;
    if (cosi < R(0)) { R tmp = ei; ei = et; et = tmp; }
    R st = ei / et * r_sqrt(max0(R(1) - ci * ci));
    if (st >= R(1)) return R(1);
    R ct = r_sqrt(max0(R(1) - st * st));
    return fr_diel(r_abs(ci), ct, ei, et);
}

// fr_cond, reflection/utility.py:60-73, one channel; FresnelConductor.__call__ passes |cos| (fresnel.py:40-41)
template <typename R>
__device__ __forceinline__ R fr_cond(R cos_i, R eta, R k) {
    R tmp = (eta * eta + k * k) * cos_i * cos_i;
    R r_para_sq = (tmp - (R(2) * eta * cos_i) + R(1)) / (tmp + (R(2) * eta * cos_i) + R(1));
    tmp = eta * eta + k * k;
    R r_perp_sq = (tmp - (R(2) * eta * cos_i) + cos_i * cos_i) / (tmp + (R(2) * eta * cos_i) + cos_i * cos_i);
    return R(.5) * (r_para_sq + r_perp_sq);
}

// reflection/utility.py:31-42
template <typename R> __device__ __forceinline__ R sin_theta(v3<R> w) { return r_sqrt(max0(R(1) - w.z * w.z)); }
template <typename R> __device__ __forceinline__ R cos_phi(v3<R> w) {
    R s2 = max0(R(1) - w.z * w.z);
    return s2 == R(0) ? R(1) : clipd(w.x / r_sqrt(s2), R(-1), R(1));
}
template <typename R> __device__ __forceinline__ R sin_phi(v3<R> w) {
    R s2 = max0(R(1) - w.z * w.z);
    return s2 == R(0) ? R(0) : clipd(w.y / r_sqrt(s2), R(-1), R(1));
}

// Blinn.pdf, reflection/bdf/torrance.py:95-102
template <typename R>
__device__ inline R blinn_pdf(R e, v3<R> wo, v3<R> wi) {
    v3<R> wh = normalize3(wo + wi);
    R ct = r_abs(wh.z);
    R dwh = dot3(wo, wh);
    if (dwh <= R(0)) return R(0);
    return ((e + R(1)) * r_pow(ct, e)) / (R(2) * R(PT_PI) * R(4) * dwh);
}

// geometry/utility.py:115-122
__device__ __forceinline__ double spherical_phi(d3 v) {
    double p = atan2(v.y, v.x);
    return p > 0. ? p : p + 2 * PT_PI;
}
// brdf_remap, reflection/utility.py:76-90
__device__ inline d3 brdf_remap(d3 wo, d3 wi) {
    double dphi = spherical_phi(wi) - spherical_phi(wo);
    if (dphi < 0.) dphi += 2. * PT_PI;
    if (dphi > 2. * PT_PI) dphi -= 2. * PT_PI;
    if (dphi > PT_PI) dphi = 2. * PT_PI - dphi;
    return mk3(sin_theta(wi) * sin_theta(wo), dphi * PT_INV_PI, wi.z * wo.z);
}
__device__ __forceinline__ int grid_cell(double x, double bmin, double inv, int g) {
    const double c = floor((x - bmin) * inv);
    return c >= (double)(g - 1) ? g - 1 : (c > 0. ? (int)c : 0);     // NaN -> 0
}
__device__ __forceinline__ double interval_dist(double x, double lo, double hi) {
    return x < lo ? lo - x : (x > hi ? x - hi : 0.);
}
// IrIsotropicBRDF.f, reflection/bdf/measured.py:51-66: the ball around m grows from .01 by 1.414 until it holds
// more than two samples or its radius exceeds 1.5; the result is the mean of the samples in it, weighted
// exp(-100 d^2).  query_ball_point (cKDTree, p = 2) = squared distance <= r*r.  Here the samples come from the grid,
// every candidate is tested exactly, and the sums run in cell order (measured_f_as below has the algorithm).
// One lane = one query, 64 different balls per wave, so the walk is divergent and VALU-bound (rocprofv3: 445 k VALU
// against 27 k memory instructions per wave).  The samples found inside a ball are LISTED during a scan
// (`hits`: PT_MEAS_HITS ints per thread in LDS, thread-strided) and weighted afterwards, so exp() runs
// max-over-lanes(samples in the ball) times per wave instead of once per scan step (some lane is inside its ball at
// nearly every step).  Tried and rejected: one flat loop in which a lane either tests a candidate or steps to its
// next row (rows hold 2-5 candidates, so the ~100-instruction row set-up ran at every step: shade 363 -> 497 ms).
#define PT_MEAS_HITS 24
#define PT_MEAS_LDS_MAX (78 * 1024)   // dynamic LDS of k_shade<KIND, 2>: hit lists + table blob; two blocks per CU (160 KB)
#define PT_MEAS_STRIDE 256   // = PT_BLOCK (pt_kernels.h checks): entry j of a thread's list is hits[j * PT_MEAS_STRIDE]
// Pointer types of the blob: explicit LDS pointers (ds_read with 32-bit addresses) when k_shade staged it, else
// global memory.  Through generic pointers every access is a flat instruction with 64-bit address arithmetic.
#define PT_AS_LDS __attribute__((address_space(3)))
template <bool LDS> struct MeasPtr {
    typedef const double *D;
    typedef const int32_t *I;
    typedef const BrdfGrid *G;
    typedef const char *C;
};
template <> struct MeasPtr<true> {
    typedef PT_AS_LDS const double *D;
    typedef PT_AS_LDS const int32_t *I;
    typedef PT_AS_LDS const BrdfGrid *G;
    typedef PT_AS_LDS const char *C;
};
typedef PT_AS_LDS int *lds_int_ptr;

template <bool LDS>
__device__ __forceinline__ void measured_flush(typename MeasPtr<LDS>::D P, typename MeasPtr<LDS>::D V, d3 m, lds_int_ptr hits, int n,
                                               double r2, double &sum_wt, d3 &v) {
    for (int j = 0; j < n; j++) {
        const int i = hits[j * PT_MEAS_STRIDE];
        const d3 d = mk3(P[3 * i], P[3 * i + 1], P[3 * i + 2]) - m;
        const double d2 = d.x * d.x + d.y * d.y + d.z * d.z;
        if (!(d2 <= r2)) continue;           // a list made at a larger radius (phase 1)
        const double wt = exp(d2 * -100.);
        sum_wt += wt;
        v = v + mk3(V[3 * i], V[3 * i + 1], V[3 * i + 2]) * wt;
    }
}
// One pass over the cells around m at radius r: visits every sample in the (y, z) rows of cells the ball touches,
// within the x range of the ball's chord in that row (padded by more than the rounding of the cell assignment), so
// every sample with d^2 <= r^2 is visited, some others too.  `visit(i, d2)` gets the sample index and its squared distance.
template <bool LDS, class F>
__device__ __forceinline__ void measured_scan(typename MeasPtr<LDS>::G gp, typename MeasPtr<LDS>::I cells, typename MeasPtr<LDS>::D P,
                                              d3 m, double r, F visit) {
    const double bx = gp->bmin[0], by = gp->bmin[1], bz = gp->bmin[2], ix = gp->inv[0], iy = gp->inv[1], iz = gp->inv[2];
    const double cwy = gp->cw[1], cwz = gp->cw[2];
    const int gx = gp->g[0], gy = gp->g[1], gz = gp->g[2];
    const double rp = r * (1. + 1e-9) + 1e-12, rp2 = rp * rp;
    const int y0 = grid_cell(m.y - rp, by, iy, gy), y1 = grid_cell(m.y + rp, by, iy, gy);
    const int z1 = grid_cell(m.z + rp, bz, iz, gz);
    for (int z = grid_cell(m.z - rp, bz, iz, gz); z <= z1; z++) {
        const double dz = interval_dist(m.z, bz + z * cwz, bz + (z + 1) * cwz);
        for (int y = y0; y <= y1; y++) {
            const double dy = interval_dist(m.y, by + y * cwy, by + (y + 1) * cwy);
            const double h2 = rp2 - dz * dz - dy * dy;
            if (!(h2 >= 0.)) continue;       // the ball misses this row of cells
            const double h = sqrt(h2) * (1. + 1e-9) + 1e-12;
            const int row = (z * gy + y) * gx;
            const int e = cells[row + grid_cell(m.x + h, bx, ix, gx) + 1];
            for (int i = cells[row + grid_cell(m.x - h, bx, ix, gx)]; i < e; i++) {
                const d3 d = mk3(P[3 * i], P[3 * i + 1], P[3 * i + 2]) - m;
                visit(i, d.x * d.x + d.y * d.y + d.z * d.z);
            }
        }
    }
}

// The reference's loop (ball radius .01, x 1.414 per round, until more than 2 samples are inside or the radius
// exceeds 1.5) ends at the first radius that reaches the THIRD-NEAREST sample.  So instead of one scan per round:
// phase 1 finds the squared distance of the third-nearest sample with one branch-free scan at a guessed radius
// (kguess of m's cell, computed at upload; the scan is repeated one round further out in the rare case that the guess
// holds fewer than 3 samples), which gives the round kf the reference stops at by the same `d2 <= r*r` comparisons;
// phase 2 scans ball kf once, lists its samples and weights them.
template <bool LDS>
__device__ __forceinline__ d3 measured_f_as(const char *base_generic, int table, d3 wo, d3 wi, int *hits_generic) {
    typedef MeasPtr<LDS> MP;
    const typename MP::C base = (typename MP::C)base_generic;
    const lds_int_ptr hits = (lds_int_ptr)hits_generic;
    const typename MP::G gp = (typename MP::G)base + table;
    const typename MP::I cells = (typename MP::I)(base + gp->off_cells);
    const typename MP::D P = (typename MP::D)(base + gp->off_p), V = (typename MP::D)(base + gp->off_v);
    const d3 m = brdf_remap(wo, wi);
    int kg = ((typename MP::I)(base + gp->off_kguess))[
        (grid_cell(m.z, gp->bmin[2], gp->inv[2], gp->g[2]) * gp->g[1] + grid_cell(m.y, gp->bmin[1], gp->inv[1], gp->g[1])) * gp->g[0] +
        grid_cell(m.x, gp->bmin[0], gp->inv[0], gp->g[0])];
    // phase 1: a0 <= a1 <= a2 = the three smallest squared distances among the visited samples; once a2 <= r_kg^2
    // they are the three smallest of the whole table (an unvisited sample lies outside ball kg).  The samples inside
    // ball kg are listed on the way: ball kf lies inside it, so if the list did not overflow phase 2 needs no scan.
    double a2;
    int nl;
    for (;; kg++) {
        const double r = gp->radius[kg], r2g = r * r;
        double a0 = __builtin_inf(), a1 = __builtin_inf();
        a2 = __builtin_inf();
        nl = 0;
        measured_scan<LDS>(gp, cells, P, m, r, [&](int i, double d2) {
            if (d2 <= r2g) {
                if (nl < PT_MEAS_HITS) hits[nl * PT_MEAS_STRIDE] = i;
                nl++;
            }
            double t = d2 == d2 ? d2 : __builtin_inf();      // a NaN query never counts a sample
            const double l0 = fmin(a0, t);
            t = fmax(a0, t);
            a0 = l0;
            const double l1 = fmin(a1, t);
            t = fmax(a1, t);
            a1 = l1;
            a2 = fmin(a2, t);
        });
        if (a2 <= r2g || r > 1.5) break;             // radius[15] is the first > 1.5
    }
    int kf = 0;
    while (!(a2 <= gp->radius[kf] * gp->radius[kf]) && !(gp->radius[kf] > 1.5)) kf++;
    // phase 2: the samples of ball kf (kf <= kg), weighted in cell order
    const double max_dist = gp->radius[kf], r2 = max_dist * max_dist;
    double sum_wt = 0.;
    d3 v = mk3(0, 0, 0);
    if (nl <= PT_MEAS_HITS) {
        measured_flush<LDS>(P, V, m, hits, nl, r2, sum_wt, v);
    } else {                                 // more than a list of samples in ball kg: scan ball kf
        int nlist = 0;
        measured_scan<LDS>(gp, cells, P, m, max_dist, [&](int i, double d2) {
            if (d2 <= r2) {
                if (nlist == PT_MEAS_HITS) {     // a dense ball: weight what is listed, go on listing
                    measured_flush<LDS>(P, V, m, hits, nlist, r2, sum_wt, v);
                    nlist = 0;
                }
                hits[nlist++ * PT_MEAS_STRIDE] = i;
            }
        });
        measured_flush<LDS>(P, V, m, hits, nlist, r2, sum_wt, v);
    }
    return mk3(max0(v.x), max0(v.y), max0(v.z)) / sum_wt;   // v.clip() / sum_wt
}
__device__ __forceinline__ d3 measured_f(const char *base, bool in_lds, int table, d3 wo, d3 wi, int *hits) {
    return in_lds ? measured_f_as<true>(base, table, wo, wi, hits) : measured_f_as<false>(base, table, wo, wi, hits);
}

// LOBES (here and below): the lobe types the calling kernel is compiled for.  k_shade is instantiated per LOBE SET:
// hits are listed by the class of their material (pt_scene_upload: the lobe-type mask of every material), and the hits
// on materials without Blinn / Fresnel / specular lobes — Matte, i.e. the head scene's backdrop and half of the
// 10^6-triangle soup — are shaded by a kernel in which those lobes do not exist, at a fraction of the registers.
// lobe_type<LOBES> returns the lobe's type restricted to the set, so that every switch below folds; for a material
// of the class it is the identity, hence the same operations in the same order and the same bits.
#define PT_LOBES_ALL 0x7F
#define PT_LOBES_DIFFUSE ((1 << PT_LOBE_LAMBERT) | (1 << PT_LOBE_OREN_NAYAR))
template <int LOBES> __device__ __forceinline__ int lobe_type(const pt_lobe &l) {
    if (LOBES == PT_LOBES_DIFFUSE) return l.type == PT_LOBE_OREN_NAYAR ? PT_LOBE_OREN_NAYAR : PT_LOBE_LAMBERT;
    return l.type;
}

// EXT (here and below): the shade kernels that scenes with measured lobes run (k_shade<KIND, 2>, pt_launch_shade);
// the measured lobe is compiled into those only, so that the other kernels keep their registers
template <bool EXT, typename R, int LOBES = PT_LOBES_ALL>
__device__ inline v3<R> lobe_f(const pt_lobe &l, v3<R> wo, v3<R> wi, const VertexT<R> &vx) {
    v3<R> Rc = mkv<R>((R)l.R[0], (R)l.R[1], (R)l.R[2]);
    const int type = lobe_type<LOBES>(l);
    if (EXT && type == PT_LOBE_MEASURED_IRISO)      // the table walk is binary64 in every variant
        return cvt3<R>(measured_f(vx.brdf, vx.brdf_in_lds, (int)l.a, cvt3<double>(wo), cvt3<double>(wi), vx.hits));
    switch (type) {
    case PT_LOBE_LAMBERT:   // bdf.py:327-332
        return Rc * R(PT_INV_PI);
    case PT_LOBE_OREN_NAYAR: {   // orennayar.py:44-62
        R st_i = sin_theta(wi), st_o = sin_theta(wo);
        R mc = R(0);
        if (st_i > R(PT_EPS) && st_o > R(PT_EPS)) {
            R dcos = cos_phi(wi) * cos_phi(wo) + sin_phi(wi) * sin_phi(wo);
            mc = max0(dcos);
        }
        R sa, tb;
        if (r_abs(wi.z) > r_abs(wo.z)) { sa = st_o; tb = st_i / r_abs(wi.z); }
        else { sa = st_i; tb = st_o / r_abs(wo.z); }
        return (Rc * R(PT_INV_PI)) * ((R)l.a + (R)l.b * mc * sa * tb);
    }
    case PT_LOBE_MICROFACET_BLINN:
    case PT_LOBE_MICROFACET_BLINN_COND: {   // torrance.py:200-216, Blinn.D :69-70
        R ct_i = wi.z, ct_o = wo.z;
        if (ct_i == R(0) || ct_o == R(0)) return mkv<R>(R(0), R(0), R(0));
        v3<R> wh = normalize3(wi + wo);
        R ct_h = dot3(wi, wh);
        R awh = r_abs(wh.z), adot = r_abs(dot3(wo, wh));
        R g1 = R(2) * awh * r_abs(wo.z) / adot, g2 = R(2) * awh * r_abs(wi.z) / adot;
        R gm = (g2 < g1) ? g2 : g1;
        R G = (gm < R(1)) ? gm : R(1);
        R D = ((R)l.c + R(2)) * R(PT_INV_2PI) * r_pow(awh, (R)l.c);
        if (type == PT_LOBE_MICROFACET_BLINN_COND) {   // Spectrum-valued Fresnel term (MetalMaterial)
            const R ch = r_abs(ct_h);
            v3<R> F = mkv<R>(fr_cond(ch, (R)l.eta[0], (R)l.k[0]), fr_cond(ch, (R)l.eta[1], (R)l.k[1]),
                             fr_cond(ch, (R)l.eta[2], (R)l.k[2]));
            return (((Rc * D) * G) * F) / (R(4) * ct_i * ct_o);
        }
        R F = fresnel_dielectric(ct_h, (R)l.a, (R)l.b);
        return (((Rc * D) * G) * F) / (R(4) * ct_i * ct_o);
    }
    default:   // specular lobes evaluate to 0, bdf.py:241-246,274-279
        return mkv<R>(R(0), R(0), R(0));
    }
}

// BDF.pdf (bdf.py:97-106), Microfacet.pdf (torrance.py:225-228), specular → 0
template <typename R, int LOBES = PT_LOBES_ALL>
__device__ inline R lobe_pdf(const pt_lobe &l, v3<R> wo, v3<R> wi) {
    switch (lobe_type<LOBES>(l)) {
    case PT_LOBE_LAMBERT:
    case PT_LOBE_OREN_NAYAR:
    case PT_LOBE_MEASURED_IRISO:
        return (wo.z * wi.z > R(0)) ? r_abs(wi.z) * R(PT_INV_PI) : R(0);
    case PT_LOBE_MICROFACET_BLINN:
    case PT_LOBE_MICROFACET_BLINN_COND:
        if (wi.z * wo.z <= R(0)) return R(0);
        return blinn_pdf((R)l.c, wo, wi);
    default:
        return R(0);
    }
}

template <bool EXT, typename R, int LOBES = PT_LOBES_ALL>
__device__ inline R lobe_sample_f(const pt_lobe &l, v3<R> wo, R u1, R u2, v3<R> &wi_out, v3<R> &f_out,
                                  const VertexT<R> &vx) {
    v3<R> Rc = mkv<R>((R)l.R[0], (R)l.R[1], (R)l.R[2]);
    switch (lobe_type<LOBES>(l)) {
    case PT_LOBE_LAMBERT:
    case PT_LOBE_OREN_NAYAR:
    case PT_LOBE_MEASURED_IRISO: {
        // BDF.sample_f (bdf.py:108-126); cosine_sample_hemisphere AS WRITTEN
        // (montecarlo/utility.py:171-184 on top of uniform_sample_sphere :83-95; Q6)
        R z = R(1) - R(2) * u1;
        R r = r_sqrt(max0(R(1) - z * z));
        R phi = R(2) * R(PT_PI) * u2, sphi, cphi;
        r_sincos(phi, &sphi, &cphi);
        v3<R> wi = mkv<R>(r * cphi, r * sphi, z);
        // sqrt(1 - x^2 - y^2) is |z| up to rounding; in binary32 the subtraction cancels for grazing directions
        // (relative error 6e-8 / z^2: 1e-4 already at |z| = 0.025, 2.5 % of these samples), so that variant takes |z|
        if (sizeof(R) == 4) wi.z = r_abs(z);
        else wi.z = r_sqrt(max0(R(1) - wi.x * wi.x - wi.y * wi.y));
        if (wo.z < R(0)) wi.z *= R(-1);
        wi_out = wi;
        f_out = lobe_f<EXT, R, LOBES>(l, wo, wi, vx);
        return lobe_pdf<R, LOBES>(l, wo, wi);
    }
    case PT_LOBE_MICROFACET_BLINN:
    case PT_LOBE_MICROFACET_BLINN_COND: {   // torrance.py:218-223, Blinn.sample_f :72-93
        R e = (R)l.c;
        R ct = r_pow(u1, R(1) / (e + R(1)));
        R st = r_sqrt(max0(R(1) - ct * ct));
        R phi = u2 * R(2) * R(PT_PI), sphi, cphi;
        r_sincos(phi, &sphi, &cphi);
        v3<R> wh = mkv<R>(st * cphi, st * sphi, ct);
        if (wo.z * wh.z <= R(0)) wh = wh * R(-1);
        R dwh = dot3(wo, wh);
        v3<R> wi = (-wo) + (wh * (R(2) * dwh));
        R pdf = ((e + R(1)) * r_pow(ct, e)) / (R(2) * R(PT_PI) * R(4) * dwh);
        if (dwh <= R(0)) pdf = R(0);
        wi_out = wi;
        if (wi.z * wo.z <= R(0)) f_out = mkv<R>(R(0), R(0), R(0));
        else f_out = lobe_f<EXT, R, LOBES>(l, wo, wi, vx);
        return pdf;
    }
    case PT_LOBE_SPEC_REFLECT: {   // bdf.py:251-256
        v3<R> wi = mkv<R>(-wo.x, -wo.y, wo.z);
        R F = (l.c != 0.) ? R(1) : fresnel_dielectric(wo.z, (R)l.a, (R)l.b);
        wi_out = wi;
        f_out = (Rc * F) / r_abs(wi.z);
        return R(1);
    }
    default: {   // SpecularTransmission.sample_f, bdf.py:284-307
        R ei = (R)l.a, et = (R)l.b;
        if (wo.z > R(0)) { R tmp = ei; ei = et; et = tmp; }
        R si_sq = max0(R(1) - wo.z * wo.z);
        R eta = ei / et;
        R st_sq = eta * eta * si_sq;
        if (st_sq >= R(1)) return R(0);
        R ct = r_sqrt(max0(R(1) - st_sq));
        if (wo.z > R(0)) ct = -ct;
        v3<R> wi = mkv<R>(eta * (-wo.x), eta * (-wo.y), ct);
        R F = fresnel_dielectric(wo.z, (R)l.a, (R)l.b);
        wi_out = wi;
        f_out = (Rc * ((et * et) / (ei * ei) * (R(1) - F))) / r_abs(wi.z);
        return R(1);
    }
    }
}

// BSDF.f, bsdf.py:207-224
template <bool EXT, typename R, int LOBES = PT_LOBES_ALL>
__device__ inline v3<R> bsdf_f(const VertexT<R> &vx, v3<R> wo_w, v3<R> wi_w, unsigned flags) {
    v3<R> wi = w2l(vx, wi_w), wo = w2l(vx, wo_w);
    if (dot3(wi_w, vx.nn) * dot3(wo_w, vx.nn) > R(0)) flags &= ~BDF_TRANSMISSION;
    else flags &= ~BDF_REFLECTION;
    v3<R> f = mkv<R>(R(0), R(0), R(0));
    for (int i = 0; i < vx.mat->n_lobes; i++)
        if (match_flag(lobe_type<LOBES>(vx.mat->lobes[i]), flags)) f = f + lobe_f<EXT, R, LOBES>(vx.mat->lobes[i], wo, wi, vx);
    return f;
}

// BSDF.pdf, bsdf.py:156-167
template <typename R, int LOBES = PT_LOBES_ALL>
__device__ inline R bsdf_pdf(const VertexT<R> &vx, v3<R> wo_w, v3<R> wi_w, unsigned flags) {
    if (vx.mat->n_lobes == 0) return R(0);
    v3<R> wo = w2l(vx, wo_w), wi = w2l(vx, wi_w);
    R pdf = R(0);
    int n_match = 0;
    for (int i = 0; i < vx.mat->n_lobes; i++)
        if (match_flag(lobe_type<LOBES>(vx.mat->lobes[i]), flags)) { n_match++; pdf += lobe_pdf<R, LOBES>(vx.mat->lobes[i], wo, wi); }
    return n_match > 0 ? pdf / n_match : R(0);
}

// BSDF.f and BSDF.pdf for the same pair of directions (estimate_direct evaluates both,
// integrator/utility.py:221-231).  Same values as bsdf_f / bsdf_pdf; a Blinn microfacet lobe computes
// its half vector and |cos θh|^e once for both (pow is the most expensive call of the vertex).
template <bool EXT, typename R, int LOBES = PT_LOBES_ALL>
__device__ inline v3<R> bsdf_f_pdf(const VertexT<R> &vx, v3<R> wo_w, v3<R> wi_w, unsigned flags, R &pdf_out) {
    v3<R> wi = w2l(vx, wi_w), wo = w2l(vx, wo_w);
    unsigned fflags = flags;
    if (dot3(wi_w, vx.nn) * dot3(wo_w, vx.nn) > R(0)) fflags &= ~BDF_TRANSMISSION;
    else fflags &= ~BDF_REFLECTION;
    v3<R> f = mkv<R>(R(0), R(0), R(0));
    R pdf = R(0);
    int n_match = 0;
    for (int i = 0; i < vx.mat->n_lobes; i++) {
        const pt_lobe &l = vx.mat->lobes[i];
        const int type = lobe_type<LOBES>(l);
        const bool mf = match_flag(type, fflags), mp = match_flag(type, flags);
        if (type == PT_LOBE_MICROFACET_BLINN && mf && mp && wi.z != R(0) && wo.z != R(0) && wi.z * wo.z > R(0)) {
            // lobe_f and blinn_pdf share wh = normalize(wo + wi) (addition commutes) and pow(|wh.z|, e)
            v3<R> Rc = mkv<R>((R)l.R[0], (R)l.R[1], (R)l.R[2]);
            v3<R> wh = normalize3(wi + wo);
            R ct_h = dot3(wi, wh);
            R F = fresnel_dielectric(ct_h, (R)l.a, (R)l.b);
            R awh = r_abs(wh.z), dwh = dot3(wo, wh), adot = r_abs(dwh);
            R g1 = R(2) * awh * r_abs(wo.z) / adot, g2 = R(2) * awh * r_abs(wi.z) / adot;
            R gm = (g2 < g1) ? g2 : g1;
            R G = (gm < R(1)) ? gm : R(1);
            R pw = r_pow(awh, (R)l.c);
            R D = ((R)l.c + R(2)) * R(PT_INV_2PI) * pw;
            f = f + (((Rc * D) * G) * F) / (R(4) * wi.z * wo.z);
            n_match++;
            if (dwh > R(0)) pdf += (((R)l.c + R(1)) * pw) / (R(2) * R(PT_PI) * R(4) * dwh);
            continue;
        }
        if (mf) f = f + lobe_f<EXT, R, LOBES>(l, wo, wi, vx);
        if (mp) { n_match++; pdf += lobe_pdf<R, LOBES>(l, wo, wi); }
    }
    pdf_out = n_match > 0 ? pdf / n_match : R(0);
    return f;
}

// BSDF.sample_f AS IT EXECUTES, bsdf.py:100-154: `bdf.type & SPECULAR` is an object, always
// truthy, so the pdf sum (:132-135) and the f re-evaluation (:141-152) never run (Q2).
// u_com picks the lobe: binary64 in every variant (a discrete decision).
template <bool EXT, typename R, int LOBES = PT_LOBES_ALL>
__device__ inline R bsdf_sample_f(const VertexT<R> &vx, v3<R> wo_w, R u0, R u1, double u_com, unsigned flags,
                                  v3<R> &wi_w, v3<R> &f) {
    const pt_material *m = vx.mat;
    int n_match = 0;
    for (int i = 0; i < m->n_lobes; i++) n_match += match_flag(lobe_type<LOBES>(m->lobes[i]), flags) ? 1 : 0;
    f = mkv<R>(R(0), R(0), R(0));
    wi_w = mkv<R>(R(0), R(0), R(0));
    if (n_match == 0) return R(0);
    int cnt = (int)floor(u_com * n_match);
    if (cnt > n_match - 1) cnt = n_match - 1;
    int pick = -1;
    for (int i = 0; i < m->n_lobes; i++)
        if (match_flag(lobe_type<LOBES>(m->lobes[i]), flags)) {
            if (cnt == 0) { pick = i; break; }
            cnt--;
        }
    if (pick < 0) return R(0);
    v3<R> wo = w2l(vx, wo_w), wi, fl;
    R pdf = lobe_sample_f<EXT, R, LOBES>(m->lobes[pick], wo, u0, u1, wi, fl, vx);
    if (pdf == R(0)) return R(0);
    wi_w = l2w(vx, wi);
    if (n_match > 1) pdf /= n_match;
    f = fl;
    return pdf;
}

// ---------------------------------------------------------------------------------------
// lights
// ---------------------------------------------------------------------------------------
// Transform.__call__ on Point (affine rows) / Vector / Normal, transform/__init__.py:54-69
__device__ __forceinline__ d3 xf_point(const double *m, d3 p) {
    return mk3((m[0] * p.x + m[1] * p.y + m[2] * p.z) + m[3], (m[4] * p.x + m[5] * p.y + m[6] * p.z) + m[7],
               (m[8] * p.x + m[9] * p.y + m[10] * p.z) + m[11]);
}
__device__ __forceinline__ d3 xf_vec(const double *m, d3 v) {
    return mk3(m[0] * v.x + m[1] * v.y + m[2] * v.z, m[4] * v.x + m[5] * v.y + m[6] * v.z,
               m[8] * v.x + m[9] * v.y + m[10] * v.z);
}
__device__ __forceinline__ d3 xf_normal(const double *minv, d3 n) {
    return mk3(minv[0] * n.x + minv[4] * n.y + minv[8] * n.z, minv[1] * n.x + minv[5] * n.y + minv[9] * n.z,
               minv[2] * n.x + minv[6] * n.y + minv[10] * n.z);
}

// geometry/utility.py:25-52
template <typename R>
__device__ inline void coordinate_system(v3<R> v1, v3<R> &v2, v3<R> &v3_) {
    if (len3(v1) == R(0)) { v2 = mkv<R>(R(0), R(1), R(0)); v3_ = mkv<R>(R(0), R(0), R(1)); return; }
    if (r_abs(v1.x) > r_abs(v1.y)) {
        R inv_len = R(1) / r_sqrt(v1.x * v1.x + v1.z * v1.z);
        v2 = mkv<R>(-v1.z * inv_len, R(0), v1.x * inv_len);
    } else {
        R inv_len = R(1) / r_sqrt(v1.y * v1.y + v1.z * v1.z);
        v2 = mkv<R>(R(0), -v1.z * inv_len, v1.y * inv_len);
    }
    v3_ = cross3(v1, v2);
}

// Sphere.intersect for a full sphere, shape/sphere.py:125-157 (np.roots → closed form).  Binary64 in every
// variant: it takes a point and decides hit / miss.
__device__ inline bool sphere_intersect(const pt_light &lt, const Ray &rw, double &thit, d3 &phit_obj) {
    d3 o = xf_point(lt.w2o, rw.o), d = xf_vec(lt.w2o, rw.d);
    double A = sqlen3(d);
    double B = 2 * dot3(d, o);
    double C = sqlen3(o) - lt.radius * lt.radius;
    double D = B * B - 4. * A * C;
    if (D <= 0. + PT_EPS) return false;
    double rootD = sqrt(D);
    double q = (B < 0.) ? -.5 * (B - rootD) : -.5 * (B + rootD);
    double t0 = q / A, t1 = C / q;
    if (t0 > t1) { double tmp = t0; t0 = t1; t1 = tmp; }
    if (t0 > rw.maxt || t1 < rw.mint) return false;
    double th = t0;
    if (t0 < rw.mint) { th = t1; if (th > rw.maxt) return false; }
    thit = th;
    phit_obj = o + d * th;
    return true;
}

// dg.nn of a full-sphere hit, sphere.py:186-224 + diffgeom.py:30-38
__device__ inline d3 sphere_hit_normal(const pt_light &lt, d3 phit) {
    const double phiMax = 2. * PT_PI, thetaMin = PT_PI, thetaMax = 0.;
    if (phit.x == 0. && phit.y == 0.) phit.x = PT_EPS * lt.radius;
    double theta = acos(clipd(phit.z / lt.radius, -1., 1.));
    double delta_theta = thetaMax - thetaMin;
    d3 dpdu = mk3(-phiMax * phit.y, phiMax * phit.x, 0.);
    double zrad = sqrt(phit.x * phit.x + phit.y * phit.y);
    double inv_zrad = 1. / zrad;
    double cphi = phit.x * inv_zrad, sphi = phit.y * inv_zrad;
    d3 dpdv = mk3(phit.z * cphi, phit.z * sphi, -lt.radius * sin(theta)) * delta_theta;
    d3 nn = normalize3(cross3(xf_vec(lt.o2w, dpdu), xf_vec(lt.o2w, dpdv)));
    const double *m = lt.o2w;
    double det = m[0] * (m[5] * m[10] - m[6] * m[9]) - m[1] * (m[4] * m[10] - m[6] * m[8]) +
                 m[2] * (m[4] * m[9] - m[5] * m[8]);
    if ((lt.ro != 0) != (det < 0.)) nn = nn * -1.;
    return nn;
}

template <typename R> struct LightSampleT {
    v3<R> Li, wi;
    R pdf;
    Ray shadow;                // a ray: binary64
};

// DiffuseAreaLight.sample_l (light/light.py:514-521) → Sphere.sample_p (shape/sphere.py:78-110),
// Sphere.pdf_p (:112-123), VisibilityTester.set_segment (light/utility.py:34-42).
// R = float: the cone frame, the sampled direction and the pdf are binary32; the inside / outside decision, the
// intersection with the sphere, the sampled point and the shadow segment are binary64.
template <typename R>
__device__ inline void sphere_light_sample_l(const pt_light &lt, d3 p, double pEps, R u1, R u2, LightSampleT<R> &o) {
    d3 ctr = mk3(lt.pos[0], lt.pos[1], lt.pos[2]);
    d3 dcp = p - ctr;
    double sqd = dcp.x * dcp.x + dcp.y * dcp.y + dcp.z * dcp.z;
    bool inside = (sqd - lt.radius * lt.radius < PT_EPS);
    d3 ps, ns;
    R pdf;
    if (inside) {
        // Sphere.sample AS WRITTEN, sphere.py:43-60 (Q7).  Rare (a vertex inside the light): binary64 throughout.
        const double phiMax = 2. * PT_PI, thetaMin = PT_PI, thetaMax = 0.;
        double z = 1. - 2. * (double)u1;
        double r = sqrt(max0(1. - z * z));
        double ph = 2. * PT_PI * (double)u2;
        d3 v = mk3(r * cos(ph), r * sin(ph), z);
        double sth = acos(clipd(v.z, -1., 1.));
        double phi = sth * phiMax * PT_INV_2PI;
        double theta = thetaMin + sth * (thetaMax - thetaMin);
        double s = sin(theta), c = cos(theta);
        v = mk3(s * cos(phi), s * sin(phi), c) * lt.radius;
        v.z = -lt.radius + v.z * (lt.radius - (-lt.radius));
        ns = normalize3(xf_normal(lt.w2o, v));
        if (lt.ro) ns = ns * -1.;
        ps = xf_point(lt.o2w, v);
    } else {
        v3<R> wc = cvt3<R>(normalize3(ctr - p));
        v3<R> wcx, wcy;
        coordinate_system(wc, wcx, wcy);
        R st_max_sq = (R)(lt.radius * lt.radius / sqd);
        R ct_max = r_sqrt(max0(R(1) - st_max_sq));
        // uniform_sample_cone with a basis, montecarlo/utility.py:222-226
        R ct = (R(1) - u1) * ct_max + u1 * R(1);
        R st = r_sqrt(R(1) - ct * ct);
        R phi = u2 * R(2) * R(PT_PI), sphi, cphi;
        r_sincos(phi, &sphi, &cphi);
        v3<R> dir = ((wcx * (cphi * st)) + (wcy * (sphi * st))) + (wc * ct);
        Ray r;
        r.o = p; r.d = cvt3<double>(dir); r.mint = PT_EPS; r.maxt = __builtin_inf();
        double thit;
        d3 dummy;
        if (!sphere_intersect(lt, r, thit, dummy)) thit = dot3(ctr - p, normalize3(r.d));
        ps = r.o + r.d * thit;
        ns = normalize3(ps - ctr);
        if (lt.ro) ns = ns * -1.;
    }
    d3 wi_d = normalize3(ps - p);
    if (inside) {
        // Shape.pdf_p, shape/__init__.py:122-140
        Ray r;
        r.o = p; r.d = wi_d; r.mint = PT_EPS; r.maxt = __builtin_inf();
        double thit;
        d3 ph;
        if (!sphere_intersect(lt, r, thit, ph)) pdf = R(0);
        else {
            d3 nn = sphere_hit_normal(lt, ph);
            d3 hitp = r.o + r.d * thit;
            double area = (2. * PT_PI) * lt.radius * (lt.radius - (-lt.radius));
            pdf = (R)(sqlen3(p - hitp) / (fabs(dot3(nn, -wi_d)) * area));
        }
    } else {
        // 1 - cos(theta_max) cancels (a light of radius 3 at distance 60: 1 - 0.99875): binary64 in every variant
        double st_max_sq = lt.radius * lt.radius / sqd;
        double ct_max = sqrt(max0(1. - st_max_sq));
        pdf = (R)(1. / (2. * PT_PI * (1. - ct_max)));
    }
    // DiffuseAreaLight.l, light.py:507-508
    if (dot3(ns, -wi_d) > 0.) o.Li = mkv<R>((R)lt.emit[0], (R)lt.emit[1], (R)lt.emit[2]);
    else o.Li = mkv<R>(R(0), R(0), R(0));
    o.wi = cvt3<R>(wi_d);
    o.pdf = pdf;
    double dist = len3(p - ps);
    o.shadow.o = p;
    o.shadow.d = (ps - p) / dist;
    o.shadow.mint = pEps;
    o.shadow.maxt = dist * (1. - PT_EPS);
}

// SpotLight.sample_l (light.py:190-196; divides by the falloff, no 1/r², Q9) and
// PointLight.sample_l (light.py:129-135).  Points and the cone decision in binary64.
template <typename R>
__device__ inline void delta_light_sample_l(const pt_light &lt, d3 p, double pEps, LightSampleT<R> &o) {
    d3 pos = mk3(lt.pos[0], lt.pos[1], lt.pos[2]);
    d3 wi = normalize3(pos - p);
    d3 I = mk3(lt.emit[0], lt.emit[1], lt.emit[2]);
    d3 Li;
    if (lt.kind == PT_LIGHT_SPOT) {
        d3 wl = normalize3(xf_vec(lt.w2o, -wi));
        double ct = wl.z, fo;
        if (ct < lt.cos_width) fo = 0.;
        else if (ct > lt.cos_falloff) fo = 1.;
        else { double d = (ct - lt.cos_width) / (lt.cos_falloff - lt.cos_width); fo = d * d * d * d; }
        Li = I / fo;
    } else {
        Li = I / sqlen3(pos - p);
    }
    o.Li = cvt3<R>(Li);
    o.wi = cvt3<R>(wi);
    o.pdf = R(1);
    double dist = len3(p - pos);
    o.shadow.o = p;
    o.shadow.d = (pos - p) / dist;
    o.shadow.mint = pEps;
    o.shadow.maxt = dist * (1. - 0.);
}

// util.is_black on an array: all |c| < EPS (utility/utility.py:119-124)
template <typename R> __device__ __forceinline__ bool util_is_black(v3<R> s) { return r_abs(s.x) < R(PT_EPS) && r_abs(s.y) < R(PT_EPS) && r_abs(s.z) < R(PT_EPS); }
// CoefficientSpectrum.is_black: all c < EPS (spectral/spectrum.py:182-187)
template <typename R> __device__ __forceinline__ bool spec_is_black(v3<R> s) { return s.x < R(PT_EPS) && s.y < R(PT_EPS) && s.z < R(PT_EPS); }
// RGBSpectrum.y, spectrum.py:587-589
__device__ __forceinline__ double spec_y(d3 s) { return (0.212671 * s.x + 0.715160 * s.y) + 0.072169 * s.z; }

// ---------------------------------------------------------------------------------------
// wave64 helpers
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned lane_prefix(unsigned long long mask) {
    return __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
}

// Append `pred` lanes of this wave to a queue: one atomic per wave (ballot + prefix sum).
// Every lane of the wave must call it (convergent).  Returns the lane's position or -1.
__device__ __forceinline__ int wave_append(bool pred, int *counter) {
    unsigned long long mask = __ballot(pred);
    if (mask == 0ull) return -1;
    unsigned n = (unsigned)__popcll(mask);
    unsigned rank = lane_prefix(mask);
    int leader = __ffsll((long long)mask) - 1;
    int base = 0;
    if ((int)(threadIdx.x & 63) == leader) base = atomicAdd(counter, (int)n);
    base = __shfl(base, leader);
    return pred ? base + (int)rank : -1;
}

__device__ __forceinline__ unsigned long long wave_sum_u64(unsigned long long v) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
    return v;
}
