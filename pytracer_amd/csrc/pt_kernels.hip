// pt_kernels.hip — wavefront kernels of the MI355X back end (gfx950, wave64).
//
// Stages (SURVEY.md §8(a')): K1 raygen → K3 trace_closest → K4 shade (+K5 light sampling)
// → K2 trace_any (+K6 shadow resolve) → … → K7 film splat, K8 film resolve.
// No MFMA anywhere: ray–box / ray–triangle tests and BSDF evaluation are not dense
// contractions; the bounding resource is cache/HBM bandwidth and divergent fp64 VALU.
#include "pt_kernels.h"
#include <type_traits>

// =========================================================================================
// BVH traversal.  One lane per ray; the per-lane todo stack of bvh.py:332 lives in LDS as
// stack[level][thread] so that lane l always hits bank l (conflict-free ds_read/write_b32).
// =========================================================================================

// Triangle.intersect / intersect_p up to the range test (shape/triangle.py:173-207,245-279)
__device__ __forceinline__ bool tri_math(double2 a0, double2 a1, double2 a2, double2 a3, double2 a4, const Ray &r,
                                         double &t_out, double &b1_out, double &b2_out) {
    d3 p1 = mk3(a0.x, a0.y, a1.x), e1 = mk3(a1.y, a2.x, a2.y), e2 = mk3(a3.x, a3.y, a4.x);
    d3 s1 = cross3(r.d, e2);
    double div = dot3(s1, e1);
    if (div == 0.0) return false;
    double divInv = 1.0 / div;
    d3 d = r.o - p1;
    double b1 = dot3(d, s1) * divInv;
    if (b1 < 0.0 || b1 > 1.0) return false;
    d3 s2 = cross3(d, e1);
    double b2 = dot3(r.d, s2) * divInv;
    if (b2 < 0.0 || (b1 + b2) > 1.0) return false;
    double t = dot3(e2, s2) * divInv;
    if (t < r.mint || t > r.maxt) return false;
    t_out = t; b1_out = b1; b2_out = b2;
    return true;
}
__device__ __forceinline__ bool tri_test(const GTri *tp, const Ray &r, double &t_out, double &b1_out, double &b2_out) {
    const double2 *q = reinterpret_cast<const double2 *>(tp);
    return tri_math(q[0], q[1], q[2], q[3], q[4], r, t_out, b1_out, b2_out);
}
// BVH._intersect_p (bvh.py:295-320): the slab test of a node box in the reference's comparison order.
// PT_FAST_SLAB: when every 1/d is finite and the box is ordered (min <= max), each axis' two plane distances are
// ordered by the monotonicity of rounding, none is NaN, and the reference's nested tests reduce to
//   max(near_x, near_y, near_z) <= min(far_x, far_y, far_z)  and  near < maxt  and  far > mint
// with near/far = min/max of the axis' two products: the same six subtractions and products, so the same decision
// bit for bit, without the sign selects and the nested branches.  Rays with a zero (or subnormal, or NaN)
// direction component — where 0 x inf = NaN makes the reference's outcome depend on its comparison order — and
// scenes with unordered boxes keep the literal form.
#ifndef PT_FAST_SLAB
#define PT_FAST_SLAB 1
#endif
struct SlabRay {
    double invx, invy, invz;
    bool negx, negy, negz, fast;
    int negmask;               // bit a = direction component a is negative
};
__device__ __forceinline__ SlabRay slab_ray(const DScene &sc, const Ray &r) {
    SlabRay v;
    v.invx = 1.0 / r.d.x; v.invy = 1.0 / r.d.y; v.invz = 1.0 / r.d.z;
    v.negx = r.d.x < 0.0; v.negy = r.d.y < 0.0; v.negz = r.d.z < 0.0;
    v.negmask = (v.negx ? 1 : 0) | (v.negy ? 2 : 0) | (v.negz ? 4 : 0);
    v.fast = PT_FAST_SLAB && !sc.exact_slab && isfinite(v.invx) && isfinite(v.invy) && isfinite(v.invz);
    return v;
}
template <bool FAST>
__device__ __forceinline__ bool slab_hit(const double2 n0, const double2 n1, const double2 n2, const Ray &r,
                                         const SlabRay &v) {
    // n0 = bmin.xy, n1 = bmin.z bmax.x, n2 = bmax.yz
    if (FAST) {
        const double x0 = (n0.x - r.o.x) * v.invx, x1 = (n1.y - r.o.x) * v.invx;
        const double y0 = (n0.y - r.o.y) * v.invy, y1 = (n2.x - r.o.y) * v.invy;
        const double z0 = (n1.x - r.o.z) * v.invz, z1 = (n2.y - r.o.z) * v.invz;
        const double tnear = fmax(fmax(fmin(x0, x1), fmin(y0, y1)), fmin(z0, z1));
        const double tfar = fmin(fmin(fmax(x0, x1), fmax(y0, y1)), fmax(z0, z1));
        return (tnear <= tfar) && (tnear < r.maxt) && (tfar > r.mint);
    }
    double tmin = ((v.negx ? n1.y : n0.x) - r.o.x) * v.invx;
    double tmax = ((v.negx ? n0.x : n1.y) - r.o.x) * v.invx;
    const double tmin_y = ((v.negy ? n2.x : n0.y) - r.o.y) * v.invy;
    const double tmax_y = ((v.negy ? n0.y : n2.x) - r.o.y) * v.invy;
    if (tmin > tmax_y || tmin_y > tmax) return false;
    if (tmin_y > tmin) tmin = tmin_y;
    if (tmax_y < tmax) tmax = tmax_y;
    const double tmin_z = ((v.negz ? n2.y : n1.x) - r.o.z) * v.invz;
    const double tmax_z = ((v.negz ? n1.x : n2.y) - r.o.z) * v.invz;
    if (tmin > tmax_z || tmin_z > tmax) return false;
    if (tmin_z > tmin) tmin = tmin_z;
    if (tmax_z < tmax) tmax = tmax_z;
    return (tmin < r.maxt) && (tmax > r.mint);
}

// BVH.intersect (bvh.py:322-365) / BVH.intersect_p (:367-410) with BVH._intersect_p (:295-320).
// Returns the primitive index (closest; -1 = miss) or 0/1 (any hit).  r.maxt is shrunk by
// every accepted closest hit (aggregate/primitive.py:107).
template <bool ANY, bool FAST>
__device__ __forceinline__ int traverse_loop(const DScene &sc, Ray &r, const SlabRay &sv, int *stack, double &b1o,
                                             double &b2o, unsigned &nvis, unsigned &ntri, const long STRIDE = PT_BLOCK) {
    int result = ANY ? 0 : -1;
    int todo_idx = 0, node_idx = 0;
    for (;;) {
        const double2 *q = reinterpret_cast<const double2 *>(sc.nodes + node_idx);
        const double2 n0 = q[0], n1 = q[1], n2 = q[2];   // bmin.xy | bmin.z bmax.x | bmax.yz
        const int2 meta = *reinterpret_cast<const int2 *>(&q[3]);
        nvis++;
        const bool hitbox = slab_hit<FAST>(n0, n1, n2, r, sv);
        const int nprim = meta.y & 0xFFFF;
        if (hitbox && nprim > 0) {
            for (int i = 0; i < nprim; i++) {
                double t, b1, b2;
                ntri++;
                if (tri_test(sc.tris + meta.x + i, r, t, b1, b2)) {
                    if (ANY) return 1;
                    result = meta.x + i;
                    b1o = b1; b2o = b2;
                    r.maxt = t;
                }
            }
        }
        // next node: one straight line for descending and popping lanes (an if/else here makes the compiler nest
        // a descend-only loop inside the walk and park the popping lanes: closest hit 44 -> 63 ms)
        const bool inner = hitbox && nprim == 0;
        const bool neg = (sv.negmask >> (meta.y >> 16)) & 1;          // dir_is_neg[node.axis], bvh.py:350
        const int next = node_idx + 1;
        if (!inner && todo_idx == 0) break;
        int *slot = stack + (inner ? todo_idx : todo_idx - 1) * STRIDE;
        if (inner) *slot = neg ? next : meta.x;
        const int popped = *slot;
        node_idx = inner ? (neg ? meta.x : next) : popped;
        todo_idx += inner ? 1 : -1;
    }
    return result;
}

// The two forms of the slab test are two loops (not a branch per visit): the literal one runs for the rare lanes
// that need it after the others are done.
template <bool ANY>
__device__ __forceinline__ int traverse(const DScene &sc, Ray &r, int *stack, double &b1o, double &b2o,
                                        unsigned &nvis, unsigned &ntri) {
    if (sc.n_nodes == 0) return ANY ? 0 : -1;
    const SlabRay sv = slab_ray(sc, r);
    if (sv.fast) return traverse_loop<ANY, true>(sc, r, sv, stack, b1o, b2o, nvis, ntri);
    return traverse_loop<ANY, false>(sc, r, sv, stack, b1o, b2o, nvis, ntri);
}

// Census counters are sharded ([PT_CNT_SHARDS][CNT_COUNT], summed on the host): thousands of
// waves adding to ONE word serialise at ~88 atomics/us (MI355X_MICROARCH.md "dequeue"), which
// cost every launch hundreds of microseconds before sharding.  Waves that traced nothing skip.
template <bool ANY>
__device__ __forceinline__ void trace_count(unsigned long long *counters, unsigned nvis, unsigned ntri, unsigned nray) {
    unsigned long long x = wave_sum_u64(nray);
    x = __shfl(x, 0);
    if (x == 0ull) return;
    unsigned long long v = wave_sum_u64(nvis), w = wave_sum_u64(ntri);
    if ((threadIdx.x & 63) == 0) {
        unsigned long long *c = counters + (size_t)(blockIdx.x % PT_CNT_SHARDS) * CNT_COUNT;
        atomicAdd(c + CNT_NODE_VISITS, v);
        atomicAdd(c + CNT_TRI_TESTS, w);
        atomicAdd(c + (ANY ? CNT_RAYS_ANY : CNT_RAYS_CLOSEST), x);
        if (ANY) {
            atomicAdd(c + CNT_NODE_VISITS_ANY, v);
            atomicAdd(c + CNT_TRI_TESTS_ANY, w);
        }
    }
}

// rays: component-major SoA [8][stride]; outputs per ray at the ray's own index.  ANY + rs.L:
// the shadow "resolve" (K6) is fused: an unoccluded ray adds its precomputed contribution to
// L[slot].
#ifndef PT_TRACE_WAVES
#define PT_TRACE_WAVES 5   // <= 96 VGPRs: the two loops of traverse<> spill at 80 (6 waves/SIMD measured slower)
#endif
__device__ __forceinline__ Ray load_ray(const double *__restrict__ rays, long stride, long i) {
    Ray r;
    r.o = mk3(rays[i], rays[stride + i], rays[2 * stride + i]);
    r.d = mk3(rays[3 * stride + i], rays[4 * stride + i], rays[5 * stride + i]);
    r.mint = rays[6 * stride + i];
    r.maxt = rays[7 * stride + i];
    return r;
}
// what every trace kernel does with a finished ray: hit record (closest) or the fused shadow resolve (any hit)
struct TraceOut {
    int *__restrict__ prim;
    double *__restrict__ t, *__restrict__ b1, *__restrict__ b2;
    ShadowResolve rs;
};
template <bool ANY>
__device__ __forceinline__ void store_result(const TraceOut &o, long i, int res, const Ray &r, double b1, double b2) {
    if (o.prim) o.prim[i] = res;
    if (!ANY) {
        if (o.t) o.t[i] = r.maxt;
        if (o.b1) o.b1[i] = b1;
        if (o.b2) o.b2[i] = b2;
    } else if (o.rs.L && res == 0) {
        const int slot = o.rs.slot[i];
        unsafeAtomicAdd(o.rs.L + slot, o.rs.c[i]);
        unsafeAtomicAdd(o.rs.L + o.rs.lstride + slot, o.rs.c[o.rs.cstride + i]);
        unsafeAtomicAdd(o.rs.L + 2 * o.rs.lstride + slot, o.rs.c[2 * o.rs.cstride + i]);
    }
}
// The rays one thread serves, in both input shapes (below); returns how many.
// DYN: how the kernel compiles the chunk-cursor loop (PT_TRACE_DYNAMIC in n_val; pt_kernels.h): 0 not at all,
// 1 with the claim for the NEXT chunk in flight while this one is traced (one VGPR held across the walk: its latency is
// not in the loop), 2 claimed when needed, kept in a scalar register (k_trace_pair: a VGPR more costs its closest-hit
// form the fifth wave per SIMD, 96 -> 103, and a chunk of its long walks takes 50x the claim's round trip).
// `order` (segmented input only, nullable): entry k of a region is traced as entry order[region start + k] of that region
// (k_sort_rays: the rays of a window binned by origin cell and direction, so that a wave's lanes and the waves of a CU
// walk neighbouring parts of a tree that is far larger than the caches); results go to the ray's own index.
template <int BLOCK, int DYN, class F>
__device__ __forceinline__ unsigned for_each_ray(const int *__restrict__ seg_counts, long seg, int n_val, F one,
                                                 const int *__restrict__ order = nullptr) {
    unsigned nray = 0;
    if (seg_counts) {
        const int lane = threadIdx.x & 63;
        const int wave = (int)((blockIdx.x * BLOCK + threadIdx.x) >> 6);
        const int nsub = (int)((gridDim.x * (BLOCK / 64)) / PT_SEG);
        const int r = wave % PT_SEG, sub = wave / PT_SEG;
        if (sub < nsub) {
            const int n = seg_counts[r * PT_CNT_STRIDE];
            const long first = (long)r * seg;
            // ONE loop for both assignments (the walk is inlined once): chunk k of the region, then either the next
            // claim or the static stride
            const bool dyn = DYN && (n_val & PT_TRACE_DYNAMIC);
            int *cursor = const_cast<int *>(seg_counts) + r * PT_CNT_STRIDE + PT_CNT_CURSOR;
            int claim = (dyn && lane == 0) ? atomicAdd(cursor, 64) : 0;
            int k = dyn ? __builtin_amdgcn_readfirstlane(claim) : sub * 64;
            while (k < n) {
                if (DYN == 1 && dyn) claim = lane == 0 ? atomicAdd(cursor, 64) : 0;
                if (k + lane < n) { one(first + (order ? order[first + k + lane] : k + lane)); nray++; }
                if (DYN == 2 && dyn) claim = lane == 0 ? atomicAdd(cursor, 64) : 0;
                k = dyn ? __builtin_amdgcn_readfirstlane(claim) : k + nsub * 64;
            }
        }
    } else {
        for (long i = (long)blockIdx.x * BLOCK + threadIdx.x; i < n_val; i += (long)gridDim.x * BLOCK) { one(i); nray++; }
    }
    return nray;
}

// Two input shapes:
//  - dense (seg_counts == nullptr): rays [0, n_val), static grid-stride (pt_intersect);
//  - segmented (the render queues, pt_kernels.h "segmented queues"): PT_SEG regions of `seg`
//    entries, region r filled to seg_counts[r]; wave w serves region w % PT_SEG together with
//    the other waves of the same residue.
// Scheduling alternatives measured and rejected (while-while, lane refill, software prefetch of the
// next node, scalar loads for wave-uniform nodes, permuted chunk assignment) are in the
// history of this file and in DESIGN.md "trace: what was tried".
template <int STACK, bool ANY>
__global__ __launch_bounds__(PT_BLOCK, PT_TRACE_WAVES) void k_trace(DScene sc, const double *__restrict__ rays, long stride,
                                                    const int *__restrict__ seg_counts, long seg, int n_val,
                                                    int *__restrict__ out_prim, double *__restrict__ out_t,
                                                    double *__restrict__ out_b1, double *__restrict__ out_b2,
                                                    ShadowResolve rs, unsigned long long *counters) {
    __shared__ int s_stack[STACK * PT_BLOCK];
    unsigned nvis = 0, ntri = 0;
    const TraceOut out = {out_prim, out_t, out_b1, out_b2, rs};
    const unsigned nray = for_each_ray<PT_BLOCK, 1>(seg_counts, seg, n_val, [&](long i) {
        Ray r = load_ray(rays, stride, i);
        double b1 = 0., b2 = 0.;
        const int res = traverse<ANY>(sc, r, s_stack + threadIdx.x, b1, b2, nvis, ntri);
        store_result<ANY>(out, i, res, r, b1, b2);
    });
    if (counters) trace_count<ANY>(counters, nvis, ntri, nray);
}

// ---- top of the tree in LDS ---------------------------------------------------------------
// k_trace_top: one 1024-thread workgroup per CU keeps the first n_top nodes of the tree in
// breadth-first order (every ray walks them; ~45 % of all node visits on the head mesh) in LDS
// next to the per-lane stacks, so those visits are ds_read_b128 (128 B/clk/CU, no TA, no L1 tag
// look-ups) instead of four 64-lane vector loads each.  References on the stack and in the
// staged records say where a node lives (TopNode, pt_device.h); the visit sequence, every
// comparison and every counter are those of traverse<>.
#define PT_TOP_BLOCK 1024

template <bool ANY, bool FAST>
__device__ __forceinline__ int traverse_top_loop(const DScene &sc, const TopNode *top, Ray &r, const SlabRay &sv,
                                                 int *stack, double &b1o, double &b2o, unsigned &nvis, unsigned &ntri) {
    int result = ANY ? 0 : -1;
    int todo_idx = 0, ref = ~0;   // the root is slot 0
    for (;;) {
        double2 n0, n1, n2;
        int offset, nprim_axis, ref0, ref1;
        if (ref < 0) {
            const double2 *q = reinterpret_cast<const double2 *>(top + ~ref);
            n0 = q[0]; n1 = q[1]; n2 = q[2];
            const int4 m = *reinterpret_cast<const int4 *>(&q[3]);
            offset = m.x; nprim_axis = m.y; ref0 = m.z; ref1 = m.w;
        } else {
            const double2 *q = reinterpret_cast<const double2 *>(sc.nodes + ref);
            n0 = q[0]; n1 = q[1]; n2 = q[2];
            const int2 m = *reinterpret_cast<const int2 *>(&q[3]);
            offset = m.x; nprim_axis = m.y; ref0 = ref + 1; ref1 = m.x;
        }
        nvis++;
        const bool hitbox = slab_hit<FAST>(n0, n1, n2, r, sv);
        const int nprim = nprim_axis & 0xFFFF;
        if (hitbox && nprim > 0) {
            for (int i = 0; i < nprim; i++) {
                double t, b1, b2;
                ntri++;
                if (tri_test(sc.tris + offset + i, r, t, b1, b2)) {
                    if (ANY) return 1;
                    result = offset + i;
                    b1o = b1; b2o = b2;
                    r.maxt = t;
                }
            }
        }
        const bool inner = hitbox && nprim == 0;                      // straight line, as in traverse_loop
        const bool neg = (sv.negmask >> (nprim_axis >> 16)) & 1;
        if (!inner && todo_idx == 0) break;
        int *slot = stack + (inner ? todo_idx : todo_idx - 1) * PT_TOP_BLOCK;
        if (inner) *slot = neg ? ref0 : ref1;
        const int popped = *slot;
        ref = inner ? (neg ? ref1 : ref0) : popped;
        todo_idx += inner ? 1 : -1;
    }
    return result;
}
template <bool ANY>
__device__ __forceinline__ int traverse_top(const DScene &sc, const TopNode *top, Ray &r, int *stack, double &b1o,
                                            double &b2o, unsigned &nvis, unsigned &ntri) {
    if (sc.n_nodes == 0) return ANY ? 0 : -1;
    const SlabRay sv = slab_ray(sc, r);
    if (sv.fast) return traverse_top_loop<ANY, true>(sc, top, r, sv, stack, b1o, b2o, nvis, ntri);
    return traverse_top_loop<ANY, false>(sc, top, r, sv, stack, b1o, b2o, nvis, ntri);
}

template <bool ANY>
__global__ __launch_bounds__(PT_TOP_BLOCK) void k_trace_top(DScene sc, const double *__restrict__ rays, long stride,
                                                            const int *__restrict__ seg_counts, long seg, int n_val,
                                                            int *__restrict__ out_prim, double *__restrict__ out_t,
                                                            double *__restrict__ out_b1, double *__restrict__ out_b2,
                                                            ShadowResolve rs, unsigned long long *counters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char s_mem[];
    TopNode *top = reinterpret_cast<TopNode *>(s_mem);
    int *stack = reinterpret_cast<int *>(s_mem + (size_t)sc.n_top * sizeof(TopNode)) + threadIdx.x;
    {
        const uint4 *src = reinterpret_cast<const uint4 *>(sc.top);
        uint4 *dst = reinterpret_cast<uint4 *>(s_mem);
        for (int k = threadIdx.x; k < sc.n_top * 4; k += PT_TOP_BLOCK) dst[k] = src[k];
    }
    __syncthreads();
    unsigned nvis = 0, ntri = 0;
    const TraceOut out = {out_prim, out_t, out_b1, out_b2, rs};
    const unsigned nray = for_each_ray<PT_TOP_BLOCK, 2>(seg_counts, seg, n_val, [&](long i) {
        Ray r = load_ray(rays, stride, i);
        double b1 = 0., b2 = 0.;
        const int res = traverse_top<ANY>(sc, top, r, stack, b1, b2, nvis, ntri);
        store_result<ANY>(out, i, res, r, b1, b2);
    });
    if (counters) trace_count<ANY>(counters, nvis, ntri, nray);
}

// ---- child/primitive words in LDS ----------------------------------------------------------
// k_trace_meta: for trees small enough (8 bytes per node beside the 1024 per-lane stacks: ~7 900 nodes at 24
// levels), the (offset, n_prim | axis) word of EVERY node is staged in LDS, so a visit is three 16-byte vector
// loads (the box) plus one ds_read_b64 instead of four vector loads — the vector-memory pipe, which bounds the
// walk (scripts/ta_micro.hip: 252 vs 210 G visits/s), carries a quarter less.  Same walk as
// traverse<>, bit for bit.
// the staged (offset, n_prim | axis << 16) word of a node: as it is, or packed into 32 bits (offset 20, n_prim 10, axis 2)
__device__ __forceinline__ int2 meta_at(const int2 *t, int i) { return t[i]; }
__device__ __forceinline__ int2 meta_at(const unsigned *t, int i) {
    const unsigned m = t[i];
    return make_int2((int)(m & 0xFFFFFu), (int)(((m >> 20) & 0x3FFu) | ((m >> 30) << 16)));
}
template <bool ANY, bool FAST, int BLOCK = PT_TOP_BLOCK, typename STK = int, typename TAB = int2>
__device__ __forceinline__ int traverse_meta_loop(const DScene &sc, const TAB *mlds, Ray &r, const SlabRay &sv,
                                                  STK *stack, double &b1o, double &b2o, unsigned &nvis, unsigned &ntri) {
    int result = ANY ? 0 : -1;
    int todo_idx = 0, node_idx = 0;
    for (;;) {
        // the boxes alone at 48-byte stride (DScene::node_box, built for every tree these kernels run on): a quarter less
        // cache footprint than the 64-byte node records whose last 16 bytes this walk never reads (closest 34.7 -> 33.2 ms)
        const double2 *q = sc.node_box + 3 * node_idx;
        const double2 n0 = q[0], n1 = q[1], n2 = q[2];
        const int2 meta = meta_at(mlds, node_idx);
        nvis++;
        const bool hitbox = slab_hit<FAST>(n0, n1, n2, r, sv);
        const int nprim = meta.y & 0xFFFF;
        if (hitbox && nprim > 0) {
            for (int i = 0; i < nprim; i++) {
                double t, b1, b2;
                ntri++;
                if (tri_test(sc.tris + meta.x + i, r, t, b1, b2)) {
                    if (ANY) return 1;
                    result = meta.x + i;
                    b1o = b1; b2o = b2;
                    r.maxt = t;
                }
            }
        }
        const bool inner = hitbox && nprim == 0;
        const bool neg = (sv.negmask >> (meta.y >> 16)) & 1;
        const int next = node_idx + 1;
        if (!inner && todo_idx == 0) break;
        STK *slot = stack + (inner ? todo_idx : todo_idx - 1) * BLOCK;
        if (inner) *slot = (STK)(neg ? next : meta.x);
        const int popped = (int)*slot;
        node_idx = inner ? (neg ? meta.x : next) : popped;
        todo_idx += inner ? 1 : -1;
    }
    return result;
}

template <bool ANY>
__global__ __launch_bounds__(PT_TOP_BLOCK) void k_trace_meta(DScene sc, const double *__restrict__ rays, long stride,
                                                             const int *__restrict__ seg_counts, long seg, int n_val,
                                                             int *__restrict__ out_prim, double *__restrict__ out_t,
                                                             double *__restrict__ out_b1, double *__restrict__ out_b2,
                                                             ShadowResolve rs, unsigned long long *counters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char s_mem[];
    int2 *mlds = reinterpret_cast<int2 *>(s_mem);
    int *stack = reinterpret_cast<int *>(s_mem + pt_trace_meta_table_bytes(sc.n_nodes)) + threadIdx.x;
    {   // from the packed copy (8 bytes per node, 16-byte loads): read out of the 64-byte node records the staging of
        // a 5.5 k-node tree fetched 354 KB per workgroup, 90 MB per launch
        const int4 *src = reinterpret_cast<const int4 *>(sc.node_meta);
        int4 *dst = reinterpret_cast<int4 *>(s_mem);
        for (int k = threadIdx.x; k < (sc.n_nodes + 1) / 2; k += PT_TOP_BLOCK) dst[k] = src[k];
    }
    __syncthreads();
    unsigned nvis = 0, ntri = 0;
    const TraceOut out = {out_prim, out_t, out_b1, out_b2, rs};
    const unsigned nray = for_each_ray<PT_TOP_BLOCK, 1>(seg_counts, seg, n_val, [&](long i) {
        Ray r = load_ray(rays, stride, i);
        double b1 = 0., b2 = 0.;
        int res = ANY ? 0 : -1;
        if (sc.n_nodes > 0) {
            const SlabRay sv = slab_ray(sc, r);
            res = sv.fast ? traverse_meta_loop<ANY, true>(sc, mlds, r, sv, stack, b1, b2, nvis, ntri)
                          : traverse_meta_loop<ANY, false>(sc, mlds, r, sv, stack, b1, b2, nvis, ntri);
        }
        store_result<ANY>(out, i, res, r, b1, b2);
    });
    if (counters) trace_count<ANY>(counters, nvis, ntri, nray);
}

// k_trace_meta_s: the same kernel for trees of fewer than 2^16 nodes with 16-bit todo entries, which leaves LDS for
// several small workgroups per CU (each with its own copy of the table) instead of one of 1024 threads: the same 4 waves
// per SIMD, 5-6 % faster (headline closest-hit, one stream, ms per step with the level-0 packet launch: 1024 x 1 with
// 4-byte or 2-byte entries 36.4-36.8; 512 x 2 34.8-35.3; 384 x 2 (3 waves) 38.6, 448 x 2 36.1, 576 x 2 38.7, 640 x 2
// (5 waves) 41.1, 768 x 1 37.4, 512 x 1 47.8).  PACKED: the node word in 32 bits (meta_at: offset 20, n_prim 10, axis 2
// bits; pt_api.hip checks that the tree fits), half the table, so FOUR workgroups of 256 threads fit: 34.4 (512 x 2
// packed 34.7).  pt_api.hip uses the smallest form that fits (PT_TRACE_META_SHORT=0: the 1024-thread form, 1: 512 x 2).
template <bool ANY, int BLOCK, bool PACKED>
__global__ __launch_bounds__(BLOCK, 4) void k_trace_meta_s(DScene sc, const double *__restrict__ rays, long stride,
                                                           const int *__restrict__ seg_counts, long seg, int n_val,
                                                           int *__restrict__ out_prim, double *__restrict__ out_t,
                                                           double *__restrict__ out_b1, double *__restrict__ out_b2,
                                                           ShadowResolve rs, unsigned long long *counters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char s_mem[];
    typedef typename std::conditional<PACKED, unsigned, int2>::type TAB;
    TAB *mlds = reinterpret_cast<TAB *>(s_mem);
    unsigned short *stack = reinterpret_cast<unsigned short *>(s_mem + pt_trace_meta_table_bytes(sc.n_nodes) / (PACKED ? 2 : 1)) + threadIdx.x;
    if (PACKED) {
        const int2 *src = reinterpret_cast<const int2 *>(sc.node_meta);
        unsigned *dst = reinterpret_cast<unsigned *>(s_mem);
        for (int k = threadIdx.x; k < sc.n_nodes; k += BLOCK) {
            const int2 m = src[k];
            dst[k] = (unsigned)m.x | ((unsigned)(m.y & 0x3FF) << 20) | ((unsigned)(m.y >> 16) << 30);
        }
    } else {
        const int4 *src = reinterpret_cast<const int4 *>(sc.node_meta);
        int4 *dst = reinterpret_cast<int4 *>(s_mem);
        for (int k = threadIdx.x; k < (sc.n_nodes + 1) / 2; k += BLOCK) dst[k] = src[k];
    }
    __syncthreads();
    unsigned nvis = 0, ntri = 0;
    const TraceOut out = {out_prim, out_t, out_b1, out_b2, rs};
    const unsigned nray = for_each_ray<BLOCK, 1>(seg_counts, seg, n_val, [&](long i) {
        Ray r = load_ray(rays, stride, i);
        double b1 = 0., b2 = 0.;
        int res = ANY ? 0 : -1;
        if (sc.n_nodes > 0) {
            const SlabRay sv = slab_ray(sc, r);
            res = sv.fast ? traverse_meta_loop<ANY, true, BLOCK, unsigned short, TAB>(sc, mlds, r, sv, stack, b1, b2, nvis, ntri)
                          : traverse_meta_loop<ANY, false, BLOCK, unsigned short, TAB>(sc, mlds, r, sv, stack, b1, b2, nvis, ntri);
        }
        store_result<ANY>(out, i, res, r, b1, b2);
    });
    if (counters) trace_count<ANY>(counters, nvis, ntri, nray);
}

// ---- children pairs -------------------------------------------------------------------------
// k_trace_pair: the same walk (bvh.py:322-410), fetched by PARENT: the record of an interior node (GPair, one
// 128-byte line) holds the boxes of both children, so one dependent fetch decides two visits.  What keeps it the
// reference's walk, visit for visit:
//  * the near child (dir_is_neg[axis], bvh.py:350) is visited at once with the ray as it is;
//  * the far child is visited when the reference pops it.  Of its box test, `tnear <= tfar` and `tfar > mint` can
//    never change; `tnear < maxt` can (closest hit only: accepted hits shrink maxt) and is finished at the pop.
//    The stack entry carries the HIGH WORD of the binary64 tnear: that bounds tnear between two doubles, which
//    settles the comparison unless maxt falls between them — then the child's own box is fetched and the
//    comparison made exactly (exact_tnear_lt_maxt).  A far child whose verdict is already "miss" is counted, not
//    pushed;
//  * any hit: maxt never changes, so only hit far children are pushed; the visits of the others are counted when
//    the reference would pop them (`pend`: missed far children above the top stack entry; each entry remembers
//    how many lie under it), because an any-hit walk returns at its first hit and never visits what is left.
// Rays that need the literal slab test (zero / NaN direction components, unordered boxes) walk the node array
// with traverse_loop<ANY, false> as before.
struct PairEnt { int ref; int aux; };   // aux: closest = high word of tnear; any = missed far children under the entry
#define PT_PAIR_POP (-0x7fffffff - 1)

__device__ __forceinline__ bool exact_tnear_lt_maxt(const DScene &sc, int ref, const Ray &r, double invx, double invy,
                                                    double invz) {
    const int node = ref >= 0 ? sc.pairs[ref].self : sc.tris[~ref].leaf_node;
    const double2 *q = reinterpret_cast<const double2 *>(sc.nodes + node);
    const double2 n0 = q[0], n1 = q[1], n2 = q[2];
    const double x0 = (n0.x - r.o.x) * invx, x1 = (n1.y - r.o.x) * invx;
    const double y0 = (n0.y - r.o.y) * invy, y1 = (n2.x - r.o.y) * invy;
    const double z0 = (n1.x - r.o.z) * invz, z1 = (n2.y - r.o.z) * invz;
    const double tnear = fmax(fmax(fmin(x0, x1), fmin(y0, y1)), fmin(z0, z1));
    return tnear < r.maxt;
}

// The todo stack is split: levels [0, L) in LDS (stack[level][thread]), levels from L on in a global spill area
// (spill[level - L][grid thread], coalesced across a wave's lanes) — deep trees would otherwise take 8 bytes x depth
// of LDS per lane and leave two or three waves per SIMD to hide a latency-bound walk.
struct PairStack {
    PairEnt *lds;        // + thread
    PairEnt *spill;      // + grid thread
    long spill_stride;   // grid threads
    int L;
};
template <int BLOCK>
__device__ __forceinline__ void pair_push(const PairStack &st, int level, PairEnt e) {
    if (level < st.L) st.lds[level * BLOCK] = e;
    else st.spill[(long)(level - st.L) * st.spill_stride] = e;
}
template <int BLOCK>
__device__ __forceinline__ PairEnt pair_pop(const PairStack &st, int level) {
    if (level < st.L) return st.lds[level * BLOCK];
    return st.spill[(long)(level - st.L) * st.spill_stride];
}

// The root: a visit like any other.  false = the ray misses the whole tree.
__device__ __forceinline__ bool pair_root(const DScene &sc, const Ray &r, const SlabRay &sv, int &ref, unsigned &nvis) {
    const double2 *q = reinterpret_cast<const double2 *>(sc.nodes);
    const double2 n0 = q[0], n1 = q[1], n2 = q[2];
    const int2 meta = *reinterpret_cast<const int2 *>(&q[3]);
    nvis++;
    if (!slab_hit<true>(n0, n1, n2, r, sv)) return false;
    ref = (meta.y & 0xFFFF) ? ~meta.x : 0;
    return true;
}

// One step of the walk: the visit `ref` stands for (a hit leaf's primitives, or a pair record = two child visits) and
// the pops that follow it.  true = the walk of this ray is over (`result` holds its outcome).
template <bool ANY, int BLOCK>
__device__ __forceinline__ bool pair_step(const DScene &sc, Ray &r, const SlabRay &sv, const PairStack &stack, int &ref,
                                          int &todo, unsigned &pend, int &result, double &b1o, double &b2o,
                                          unsigned &nvis, unsigned &ntri) {
    int next;
    if (ref < 0) {
        // a leaf whose box was hit: its primitives in order (bvh.py:339-345 / :384-390)
        const int first = ~ref;
        int n = 1;
        for (int i = 0; i < n; i++) {
            const double2 *q = reinterpret_cast<const double2 *>(sc.tris + first + i);
            const double2 a0 = q[0], a1 = q[1], a2 = q[2], a3 = q[3], a4 = q[4];
            if (i == 0) n = __double2loint(a4.y);          // GTri::leaf_nprim
            double t, b1, b2;
            ntri++;
            if (tri_math(a0, a1, a2, a3, a4, r, t, b1, b2)) {
                if (ANY) { result = 1; return true; }
                result = first + i;
                b1o = b1; b2o = b2;
                r.maxt = t;
            }
        }
        next = PT_PAIR_POP;
    } else {
        const double2 *q = reinterpret_cast<const double2 *>(sc.pairs + ref);
        const double2 c0 = q[0], c1 = q[1], c2 = q[2], d0 = q[3], d1 = q[4], d2 = q[5];
        const int4 m = *reinterpret_cast<const int4 *>(&q[6]);         // ref0, ref1, axis, self
        // both slab tests in the fast form (slab_hit<true>)
        const double ax0 = (c0.x - r.o.x) * sv.invx, ax1 = (c1.y - r.o.x) * sv.invx;
        const double ay0 = (c0.y - r.o.y) * sv.invy, ay1 = (c2.x - r.o.y) * sv.invy;
        const double az0 = (c1.x - r.o.z) * sv.invz, az1 = (c2.y - r.o.z) * sv.invz;
        const double bx0 = (d0.x - r.o.x) * sv.invx, bx1 = (d1.y - r.o.x) * sv.invx;
        const double by0 = (d0.y - r.o.y) * sv.invy, by1 = (d2.x - r.o.y) * sv.invy;
        const double bz0 = (d1.x - r.o.z) * sv.invz, bz1 = (d2.y - r.o.z) * sv.invz;
        const double tn0 = fmax(fmax(fmin(ax0, ax1), fmin(ay0, ay1)), fmin(az0, az1));
        const double tf0 = fmin(fmin(fmax(ax0, ax1), fmax(ay0, ay1)), fmax(az0, az1));
        const double tn1 = fmax(fmax(fmin(bx0, bx1), fmin(by0, by1)), fmin(bz0, bz1));
        const double tf1 = fmin(fmin(fmax(bx0, bx1), fmax(by0, by1)), fmax(bz0, bz1));
        const bool ok0 = (tn0 <= tf0) && (tn0 < r.maxt) && (tf0 > r.mint);
        const bool ok1 = (tn1 <= tf1) && (tn1 < r.maxt) && (tf1 > r.mint);
        const bool neg = (sv.negmask >> m.z) & 1;                       // near child = second child when negative
        const bool okN = neg ? ok1 : ok0, okF = neg ? ok0 : ok1;
        const int refN = neg ? m.y : m.x, refF = neg ? m.x : m.y;
        nvis++;                                                         // the near child's visit
        if (okN) {
            if (okF) {
                PairEnt e;
                e.ref = refF;
                e.aux = ANY ? (int)pend : __double2hiint(neg ? tn0 : tn1);
                pair_push<BLOCK>(stack, todo, e);
                todo++;
                pend = 0;
            } else if (ANY) pend++;
            else nvis++;                                                // closest: it will be popped and missed
            next = refN;
        } else {
            nvis++;                                                     // the far child is popped right away
            next = okF ? refF : PT_PAIR_POP;
        }
    }
    if (next == PT_PAIR_POP) {
        if (ANY) {
            nvis += pend;
            if (todo == 0) return true;
            todo--;
            const PairEnt e = pair_pop<BLOCK>(stack, todo);
            nvis++;
            pend = (unsigned)e.aux;
            next = e.ref;
        } else {
            for (;;) {
                if (todo == 0) return true;
                todo--;
                const PairEnt e = pair_pop<BLOCK>(stack, todo);
                nvis++;
                // tnear lies between the two doubles that share its high word
                const double lo = __hiloint2double(e.aux, 0), hi = __hiloint2double(e.aux, (int)0xFFFFFFFF);
                const bool sure_hit = (r.maxt > lo) && (r.maxt > hi);
                const bool sure_miss = (r.maxt <= lo) && (r.maxt <= hi);
                bool hit = sure_hit;
                if (!sure_hit && !sure_miss) hit = exact_tnear_lt_maxt(sc, e.ref, r, sv.invx, sv.invy, sv.invz);
                if (hit) { next = e.ref; break; }
            }
        }
    }
    ref = next;
    return false;
}

// The whole walk of one ray.  NOT written as `while (!pair_step(...))`, although it is that, statement for statement:
// from that form the compiler rotates the loop so that lanes going from one pair to the next spin in an inner loop
// while the lanes at leaves and pops wait for them (soup closest-hit traces 81 -> 111 ms; the same pathology as the
// if/else form of traverse_loop).  k_trace_pair_refill is safe from it: the ballot at the top of
// its loop is a convergent operation no lane may skip.
template <bool ANY, int BLOCK>
__device__ __forceinline__ int traverse_pair_loop(const DScene &sc, Ray &r, const SlabRay &sv, const PairStack &stack,
                                                  double &b1o, double &b2o, unsigned &nvis, unsigned &ntri) {
    int result = ANY ? 0 : -1;
    int ref;
    {   // the root: a visit like any other
        const double2 *q = reinterpret_cast<const double2 *>(sc.nodes);
        const double2 n0 = q[0], n1 = q[1], n2 = q[2];
        const int2 meta = *reinterpret_cast<const int2 *>(&q[3]);
        nvis++;
        if (!slab_hit<true>(n0, n1, n2, r, sv)) return result;
        ref = (meta.y & 0xFFFF) ? ~meta.x : 0;
    }
    int todo = 0;
    unsigned pend = 0;
    for (;;) {
        int next;
        if (ref < 0) {
            // a leaf whose box was hit: its primitives in order (bvh.py:339-345 / :384-390)
            const int first = ~ref;
            int n = 1;
            for (int i = 0; i < n; i++) {
                const double2 *q = reinterpret_cast<const double2 *>(sc.tris + first + i);
                const double2 a0 = q[0], a1 = q[1], a2 = q[2], a3 = q[3], a4 = q[4];
                if (i == 0) n = __double2loint(a4.y);          // GTri::leaf_nprim
                double t, b1, b2;
                ntri++;
                if (tri_math(a0, a1, a2, a3, a4, r, t, b1, b2)) {
                    if (ANY) return 1;
                    result = first + i;
                    b1o = b1; b2o = b2;
                    r.maxt = t;
                }
            }
            next = PT_PAIR_POP;
        } else {
            const double2 *q = reinterpret_cast<const double2 *>(sc.pairs + ref);
            const double2 c0 = q[0], c1 = q[1], c2 = q[2], d0 = q[3], d1 = q[4], d2 = q[5];
            const int4 m = *reinterpret_cast<const int4 *>(&q[6]);         // ref0, ref1, axis, self
            // both slab tests in the fast form (slab_hit<true>)
            const double ax0 = (c0.x - r.o.x) * sv.invx, ax1 = (c1.y - r.o.x) * sv.invx;
            const double ay0 = (c0.y - r.o.y) * sv.invy, ay1 = (c2.x - r.o.y) * sv.invy;
            const double az0 = (c1.x - r.o.z) * sv.invz, az1 = (c2.y - r.o.z) * sv.invz;
            const double bx0 = (d0.x - r.o.x) * sv.invx, bx1 = (d1.y - r.o.x) * sv.invx;
            const double by0 = (d0.y - r.o.y) * sv.invy, by1 = (d2.x - r.o.y) * sv.invy;
            const double bz0 = (d1.x - r.o.z) * sv.invz, bz1 = (d2.y - r.o.z) * sv.invz;
            const double tn0 = fmax(fmax(fmin(ax0, ax1), fmin(ay0, ay1)), fmin(az0, az1));
            const double tf0 = fmin(fmin(fmax(ax0, ax1), fmax(ay0, ay1)), fmax(az0, az1));
            const double tn1 = fmax(fmax(fmin(bx0, bx1), fmin(by0, by1)), fmin(bz0, bz1));
            const double tf1 = fmin(fmin(fmax(bx0, bx1), fmax(by0, by1)), fmax(bz0, bz1));
            const bool ok0 = (tn0 <= tf0) && (tn0 < r.maxt) && (tf0 > r.mint);
            const bool ok1 = (tn1 <= tf1) && (tn1 < r.maxt) && (tf1 > r.mint);
            const bool neg = (sv.negmask >> m.z) & 1;                       // near child = second child when negative
            const bool okN = neg ? ok1 : ok0, okF = neg ? ok0 : ok1;
            const int refN = neg ? m.y : m.x, refF = neg ? m.x : m.y;
            nvis++;                                                         // the near child's visit
            if (okN) {
                if (okF) {
                    PairEnt e;
                    e.ref = refF;
                    e.aux = ANY ? (int)pend : __double2hiint(neg ? tn0 : tn1);
                    pair_push<BLOCK>(stack, todo, e);
                    todo++;
                    pend = 0;
                } else if (ANY) pend++;
                else nvis++;                                                // closest: it will be popped and missed
                next = refN;
            } else {
                nvis++;                                                     // the far child is popped right away
                next = okF ? refF : PT_PAIR_POP;
            }
        }
        if (next == PT_PAIR_POP) {
            if (ANY) {
                nvis += pend;
                if (todo == 0) break;
                todo--;
                const PairEnt e = pair_pop<BLOCK>(stack, todo);
                nvis++;
                pend = (unsigned)e.aux;
                next = e.ref;
            } else {
                for (;;) {
                    if (todo == 0) { next = PT_PAIR_POP; break; }
                    todo--;
                    const PairEnt e = pair_pop<BLOCK>(stack, todo);
                    nvis++;
                    // tnear lies between the two doubles that share its high word
                    const double lo = __hiloint2double(e.aux, 0), hi = __hiloint2double(e.aux, (int)0xFFFFFFFF);
                    const bool sure_hit = (r.maxt > lo) && (r.maxt > hi);
                    const bool sure_miss = (r.maxt <= lo) && (r.maxt <= hi);
                    bool hit = sure_hit;
                    if (!sure_hit && !sure_miss) hit = exact_tnear_lt_maxt(sc, e.ref, r, sv.invx, sv.invy, sv.invz);
                    if (hit) { next = e.ref; break; }
                }
                if (next == PT_PAIR_POP) break;
            }
        }
        ref = next;
    }
    return result;
}

#define PT_PAIR_BLOCK 256
#ifndef PT_PAIR_WAVES
#define PT_PAIR_WAVES 5   // 96 VGPRs (12 bytes of scratch in the closest-hit form, outside the walk); 4 -> 101 VGPRs, same speed static, slower with chunk cursors (soup closest 82.0 vs 80.4 ms)
#endif
#ifndef PT_PAIR_DYN
#define PT_PAIR_DYN 2
#endif
template <bool ANY>
__global__ __launch_bounds__(PT_PAIR_BLOCK, PT_PAIR_WAVES) void k_trace_pair(DScene sc, const double *__restrict__ rays, long stride,
                                                             const int *__restrict__ seg_counts, long seg, int n_val,
                                                             int *__restrict__ out_prim, double *__restrict__ out_t,
                                                             double *__restrict__ out_b1, double *__restrict__ out_b2,
                                                             ShadowResolve rs, unsigned long long *counters,
                                                             PairEnt *__restrict__ spill, int lds_levels,
                                                             const int *__restrict__ order) {
    extern __shared__ __attribute__((aligned(16))) unsigned char s_mem[];
    PairStack stack;
    stack.lds = reinterpret_cast<PairEnt *>(s_mem) + threadIdx.x;            // stack[level][thread], 8-byte entries
    stack.spill = spill + ((long)blockIdx.x * PT_PAIR_BLOCK + threadIdx.x);
    stack.spill_stride = (long)gridDim.x * PT_PAIR_BLOCK;
    stack.L = lds_levels;
    unsigned nvis = 0, ntri = 0;
    const TraceOut out = {out_prim, out_t, out_b1, out_b2, rs};
    const unsigned nray = for_each_ray<PT_PAIR_BLOCK, PT_PAIR_DYN>(seg_counts, seg, n_val, [&](long i) {
        Ray r = load_ray(rays, stride, i);
        double b1 = 0., b2 = 0.;
        int res = ANY ? 0 : -1;
        if (sc.n_nodes > 0) {
            const SlabRay sv = slab_ray(sc, r);
            if (sv.fast) res = traverse_pair_loop<ANY, PT_PAIR_BLOCK>(sc, r, sv, stack, b1, b2, nvis, ntri);
            else   // the literal walk keeps its whole stack in the spill area (one int per level)
                res = traverse_loop<ANY, false>(sc, r, sv, reinterpret_cast<int *>(stack.spill), b1, b2, nvis, ntri,
                                                2 * stack.spill_stride);
        }
        store_result<ANY>(out, i, res, r, b1, b2);
    }, order);
    if (counters) trace_count<ANY>(counters, nvis, ntri, nray);
}

// ---- half-size pair records --------------------------------------------------------------------------------------------
// pair_step with the child boxes read from GPairH (binary32 outer boxes, 64 bytes).  For each child the reference's test
// (the fast slab form on the exact binary64 box) is
//     ok = (tnear <= tfar) && (tnear < maxt) && (tfar > mint),   tnear / tfar = max / min over the axes of the plane distances
//     t = fl(fl(b - o) * inv).
// The same expressions on the OUTER box give tnear_o <= tnear and tfar_o >= tfar (rounding is monotone), so a missed outer
// box is a missed box.  Every plane distance moves by at most
//     (w + 2^-48 |b_o - o|) |inv|        (w >= |b_o - b|; the second term covers the roundings of both evaluations)
// which is at most E = w A + R with two numbers per RAY: A = max |inv| over the axes, R = 2^-48 max (S_a |inv_a|),
// S_a = |o_a| + the largest |coordinate| of the root box on axis a (every box lies inside the root box); both carry a factor
// 1 + 2^-40, R an absolute 2^-1000.  The exact test is a hit whenever
//     tnear_o + E <= tfar_o - E,  tnear_o + E < maxt,  tfar_o - E > mint.
// Anything in between (a ray within a binary32 ulp of grazing the box, or of the current maxt) fetches the exact GPair and
// runs the exact test for both children.  The far child's stack entry keeps the high word of tnear_o - E, a lower bound of
// the exact tnear; at the pop the exact value lies between the smaller double with that high word and the larger one
// + 2 E_max (E_max = w_max A + R with the largest w of the scene): `tnear < maxt` is settled unless maxt falls in that
// interval, and then exact_tnear_lt_maxt re-tests on the exact box.
struct HalfBound { double A, R; };
__device__ __forceinline__ HalfBound half_bound(const DScene &sc, const Ray &r, const SlabRay &sv) {
    const GNode &root = sc.nodes[0];
    const double aix = fabs(sv.invx), aiy = fabs(sv.invy), aiz = fabs(sv.invz);
    const double sx = fabs(r.o.x) + fmax(fabs(root.bmin[0]), fabs(root.bmax[0]));
    const double sy = fabs(r.o.y) + fmax(fabs(root.bmin[1]), fabs(root.bmax[1]));
    const double sz = fabs(r.o.z) + fmax(fabs(root.bmin[2]), fabs(root.bmax[2]));
    HalfBound h;
    h.A = fmax(fmax(aix, aiy), aiz) * 1.0000000000009095;                                        // 1 + 2^-40
    h.R = 3.552713678800501e-15 * fmax(fmax(sx * aix, sy * aiy), sz * aiz) * 1.0000000000009095 + 9.332636185032189e-302;
    return h;
}
template <bool ANY, int BLOCK>
__device__ __forceinline__ bool pair_step_h(const DScene &sc, Ray &r, const SlabRay &sv, const HalfBound &hb, const PairStack &stack,
                                            int &ref, int &todo, unsigned &pend, int &result, double &b1o, double &b2o,
                                            unsigned &nvis, unsigned &ntri) {
    int next;
    if (ref < 0) {
        const int first = ~ref;
        int n = 1;
        for (int i = 0; i < n; i++) {
            const double2 *q = reinterpret_cast<const double2 *>(sc.tris + first + i);
            const double2 a0 = q[0], a1 = q[1], a2 = q[2], a3 = q[3], a4 = q[4];
            if (i == 0) n = __double2loint(a4.y);          // GTri::leaf_nprim
            double t, b1, b2;
            ntri++;
            if (tri_math(a0, a1, a2, a3, a4, r, t, b1, b2)) {
                if (ANY) { result = 1; return true; }
                result = first + i;
                b1o = b1; b2o = b2;
                r.maxt = t;
            }
        }
        next = PT_PAIR_POP;
    } else {
        const float4 *q = reinterpret_cast<const float4 *>(sc.pairs_h + ref);
        const float4 f0 = q[0], f1 = q[1], f2 = q[2];
        const int4 m = *reinterpret_cast<const int4 *>(&q[3]);         // ref0, ref1, axis, w
        // b0 = f0.x f0.y f0.z | f0.w f1.x f1.y     b1 = f1.z f1.w f2.x | f2.y f2.z f2.w
        const double ax0 = ((double)f0.x - r.o.x) * sv.invx, ax1 = ((double)f0.w - r.o.x) * sv.invx;
        const double ay0 = ((double)f0.y - r.o.y) * sv.invy, ay1 = ((double)f1.x - r.o.y) * sv.invy;
        const double az0 = ((double)f0.z - r.o.z) * sv.invz, az1 = ((double)f1.y - r.o.z) * sv.invz;
        const double bx0 = ((double)f1.z - r.o.x) * sv.invx, bx1 = ((double)f2.y - r.o.x) * sv.invx;
        const double by0 = ((double)f1.w - r.o.y) * sv.invy, by1 = ((double)f2.z - r.o.y) * sv.invy;
        const double bz0 = ((double)f2.x - r.o.z) * sv.invz, bz1 = ((double)f2.w - r.o.z) * sv.invz;
        double tn0 = fmax(fmax(fmin(ax0, ax1), fmin(ay0, ay1)), fmin(az0, az1));
        const double tf0 = fmin(fmin(fmax(ax0, ax1), fmax(ay0, ay1)), fmax(az0, az1));
        double tn1 = fmax(fmax(fmin(bx0, bx1), fmin(by0, by1)), fmin(bz0, bz1));
        const double tf1 = fmin(fmin(fmax(bx0, bx1), fmax(by0, by1)), fmax(bz0, bz1));
        bool ok0 = (tn0 <= tf0) && (tn0 < r.maxt) && (tf0 > r.mint);
        bool ok1 = (tn1 <= tf1) && (tn1 < r.maxt) && (tf1 > r.mint);
        double eb = 0.;
        if (ok0 || ok1) {       // an outer box was hit: is the exact box hit too?
            eb = (double)__int_as_float(m.w) * hb.A + hb.R;
            const bool sure0 = (tn0 + eb <= tf0 - eb) && (tn0 + eb < r.maxt) && (tf0 - eb > r.mint);
            const bool sure1 = (tn1 + eb <= tf1 - eb) && (tn1 + eb < r.maxt) && (tf1 - eb > r.mint);
            if ((ok0 && !sure0) || (ok1 && !sure1)) {
                // too close to call on the outer boxes: the exact record, the exact test (as pair_step)
                const double2 *x = reinterpret_cast<const double2 *>(sc.pairs + ref);
                const double2 c0 = x[0], c1 = x[1], c2 = x[2], d0 = x[3], d1 = x[4], d2 = x[5];
                const double xa0 = (c0.x - r.o.x) * sv.invx, xa1 = (c1.y - r.o.x) * sv.invx;
                const double ya0 = (c0.y - r.o.y) * sv.invy, ya1 = (c2.x - r.o.y) * sv.invy;
                const double za0 = (c1.x - r.o.z) * sv.invz, za1 = (c2.y - r.o.z) * sv.invz;
                const double xb0 = (d0.x - r.o.x) * sv.invx, xb1 = (d1.y - r.o.x) * sv.invx;
                const double yb0 = (d0.y - r.o.y) * sv.invy, yb1 = (d2.x - r.o.y) * sv.invy;
                const double zb0 = (d1.x - r.o.z) * sv.invz, zb1 = (d2.y - r.o.z) * sv.invz;
                tn0 = fmax(fmax(fmin(xa0, xa1), fmin(ya0, ya1)), fmin(za0, za1));
                const double xf0 = fmin(fmin(fmax(xa0, xa1), fmax(ya0, ya1)), fmax(za0, za1));
                tn1 = fmax(fmax(fmin(xb0, xb1), fmin(yb0, yb1)), fmin(zb0, zb1));
                const double xf1 = fmin(fmin(fmax(xb0, xb1), fmax(yb0, yb1)), fmax(zb0, zb1));
                ok0 = (tn0 <= xf0) && (tn0 < r.maxt) && (xf0 > r.mint);
                ok1 = (tn1 <= xf1) && (tn1 < r.maxt) && (xf1 > r.mint);
                eb = 0.;
            }
        }
        const bool neg = (sv.negmask >> m.z) & 1;                       // near child = second child when negative
        const bool okN = neg ? ok1 : ok0, okF = neg ? ok0 : ok1;
        const int refN = neg ? m.y : m.x, refF = neg ? m.x : m.y;
        nvis++;                                                         // the near child's visit
        if (okN) {
            if (okF) {
                PairEnt e;
                e.ref = refF;
                if (ANY) e.aux = (int)pend;
                else {
                    // the high word of a LOWER bound of the exact tnear; the pop adds the ray's largest possible 2 E on top
                    e.aux = __double2hiint((neg ? tn0 : tn1) - eb);
                }
                pair_push<BLOCK>(stack, todo, e);
                todo++;
                pend = 0;
            } else if (ANY) pend++;
            else nvis++;                                                // closest: it will be popped and missed
            next = refN;
        } else {
            nvis++;                                                     // the far child is popped right away
            next = okF ? refF : PT_PAIR_POP;
        }
    }
    if (next == PT_PAIR_POP) {
        if (ANY) {
            nvis += pend;
            if (todo == 0) return true;
            todo--;
            const PairEnt e = pair_pop<BLOCK>(stack, todo);
            nvis++;
            pend = (unsigned)e.aux;
            next = e.ref;
        } else {
            for (;;) {
                if (todo == 0) return true;
                todo--;
                const PairEnt e = pair_pop<BLOCK>(stack, todo);
                nvis++;
                // exact tnear in [the smaller of the two doubles with this high word, the larger one + 2 E], E <= w_max A + R
                const double lo = __hiloint2double(e.aux, 0), hi = __hiloint2double(e.aux, (int)0xFFFFFFFF);
                const double e2 = 2. * ((double)sc.pair_w_max * hb.A + hb.R);
                const bool sure_hit = (r.maxt > lo + e2) && (r.maxt > hi + e2);
                const bool sure_miss = (r.maxt <= lo) && (r.maxt <= hi);
                bool hit = sure_hit;
                if (!sure_hit && !sure_miss) hit = exact_tnear_lt_maxt(sc, e.ref, r, sv.invx, sv.invy, sv.invz);
                if (hit) { next = e.ref; break; }
            }
        }
    }
    ref = next;
    return false;
}

// k_trace_pair_refill: the same walk with the wave's lanes kept busy.  A chunk of 64 rays ends with its longest ray: on
// the 10^6-triangle soup the lanes of k_trace_pair are busy 31 % of a wave's steps (sum of visits / 64 x longest
// lane, measured), and that walk is bound by the latency of its fetches, not by their number — an idle lane is a
// fetch not in flight.  Here a lane that finishes its ray stores the result and waits; once PT_REFILL_MIN lanes wait
// (or the wave has nothing left), the wave claims that many rays from the region's cursor (one atomic) and the waiting
// lanes start on them.  Segmented input with PT_TRACE_DYNAMIC only (the cursor); per-ray results and the census
// cannot depend on which lane walks a ray.  A refill costs the wave a claim, the ray loads and the set-up code at a
// fraction of its lanes (~4 us): waiting for 4 / 8 / 16 / 24 / 32 / 48 lanes gives 74.4 / 71.0 / 67.8 / 66.1 / 65.5 /
// 66.9 ms for the soup's closest-hit traces (81.0 without refill).  On head_loop (352 k nodes, 46 visits per ray, lanes
// 59 % busy) it loses at every setting (closest 1.5 vs 1.3 ms): pt_api.hip turns it on for trees of 2^20 nodes or more.
#ifndef PT_REFILL_MIN
#define PT_REFILL_MIN 32
#endif
// waves per SIMD of the half-record kernels, measured on the 10^6-triangle soup (16 spp, one stream): closest-hit 4
// (109 VGPRs, no scratch: 55.5 ms; 5 -> 96 VGPRs + 48 B scratch: 57.3; 6 -> 86.8), any-hit 5 (27.3 ms; 4 -> 29.0; 6 -> 32.4)
#ifndef PT_PAIRH_WAVES
#define PT_PAIRH_WAVES(ANY) ((ANY) ? 5 : 4)
#endif
// HALF: the pair records are read at half size (GPairH, pair_step_h above).  SORT: rays are taken through `order`
// (k_sort_rays); a template argument because the pointer costs the default kernels 16-40 bytes of scratch.
template <bool ANY, bool HALF, bool SORT>
__global__ __launch_bounds__(PT_PAIR_BLOCK, HALF ? PT_PAIRH_WAVES(ANY) : PT_PAIR_WAVES) void k_trace_pair_refill(
    DScene sc, const double *__restrict__ rays, long stride, const int *__restrict__ seg_counts, long seg,
    int *__restrict__ out_prim, double *__restrict__ out_t, double *__restrict__ out_b1, double *__restrict__ out_b2,
    ShadowResolve rs, unsigned long long *counters, PairEnt *__restrict__ spill, int lds_levels, int refill_min,
    const int *__restrict__ order) {
    extern __shared__ __attribute__((aligned(16))) unsigned char s_mem[];
    PairStack stack;
    stack.lds = reinterpret_cast<PairEnt *>(s_mem) + threadIdx.x;
    stack.spill = spill + ((long)blockIdx.x * PT_PAIR_BLOCK + threadIdx.x);
    stack.spill_stride = (long)gridDim.x * PT_PAIR_BLOCK;
    stack.L = lds_levels;
    unsigned nvis = 0, ntri = 0, nray = 0;
    const TraceOut out = {out_prim, out_t, out_b1, out_b2, rs};
    const int lane = threadIdx.x & 63;
    const int wave = (int)((blockIdx.x * PT_PAIR_BLOCK + threadIdx.x) >> 6);
    const int nsub = (int)((gridDim.x * (PT_PAIR_BLOCK / 64)) / PT_SEG);
    const int region = wave % PT_SEG;
    if (wave / PT_SEG < nsub && sc.n_nodes > 0) {
        const int n = seg_counts[region * PT_CNT_STRIDE];
        const long first = (long)region * seg;
        int *cursor = const_cast<int *>(seg_counts) + region * PT_CNT_STRIDE + PT_CNT_CURSOR;
        bool more = n > 0;                 // wave-uniform: the region may still hold unclaimed rays
        bool have = false;
        int k = 0, ref = 0, todo = 0, result = 0;
        unsigned pend = 0;
        Ray r;
        SlabRay sv;
        HalfBound hb;
        hb.A = 0.; hb.R = 0.;
        double b1 = 0., b2 = 0.;
        r.o = r.d = mk3(0, 0, 0); r.mint = 0.; r.maxt = 0.;
        sv.invx = sv.invy = sv.invz = 0.; sv.negx = sv.negy = sv.negz = false; sv.negmask = 0; sv.fast = true;
        for (;;) {
            const unsigned long long waiting = __ballot(!have);
            const int nwait = __popcll(waiting);
            if (more && (nwait >= refill_min || nwait == 64)) {
                int base = lane == 0 ? atomicAdd(cursor, nwait) : 0;
                base = __builtin_amdgcn_readfirstlane(base);
                if (base + nwait >= n) more = false;
                if (!have) {
                    k = base + (int)lane_prefix(waiting);
                    if (k < n) {
                        if (SORT) k = order[first + k];       // the claimed position's ray (k_sort_rays); k < n still holds
                        r = load_ray(rays, stride, first + k);
                        b1 = 0.; b2 = 0.;
                        result = ANY ? 0 : -1;
                        todo = 0; pend = 0;
                        sv = slab_ray(sc, r);
                        if (HALF) hb = half_bound(sc, r, sv);
                        nray++;
                        bool walking;
                        if (sv.fast) walking = pair_root(sc, r, sv, ref, nvis);
                        else {     // the literal walk, to its end: rare (zero / NaN direction components)
                            result = traverse_loop<ANY, false>(sc, r, sv, reinterpret_cast<int *>(stack.spill), b1, b2, nvis,
                                                               ntri, 2 * stack.spill_stride);
                            walking = false;
                        }
                        if (walking) have = true;
                        else store_result<ANY>(out, first + k, result, r, b1, b2);
                    }
                }
            }
            if (__ballot(have) == 0ull) {
                if (!more) break;
                continue;                  // every claimed ray ended at once (missed the root): claim again
            }
            if (have && (HALF ? pair_step_h<ANY, PT_PAIR_BLOCK>(sc, r, sv, hb, stack, ref, todo, pend, result, b1, b2, nvis, ntri)
                              : pair_step<ANY, PT_PAIR_BLOCK>(sc, r, sv, stack, ref, todo, pend, result, b1, b2, nvis, ntri))) {
                store_result<ANY>(out, first + k, result, r, b1, b2);
                have = false;
            }
        }
    }
    if (counters) trace_count<ANY>(counters, nvis, ntri, nray);
}

size_t pt_trace_pair_lds(int lds_levels) { return (size_t)lds_levels * PT_PAIR_BLOCK * sizeof(PairEnt); }
size_t pt_trace_pair_spill_bytes(int levels, long grid_threads) { return (size_t)levels * (size_t)grid_threads * sizeof(PairEnt); }

int pt_trace_pair_occupancy(bool any, int lds_levels, bool half) {
    int nb = 0;
    if (half && any) (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_trace_pair_refill<true, true, false>, PT_PAIR_BLOCK, pt_trace_pair_lds(lds_levels));
    else if (half) (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_trace_pair_refill<false, true, false>, PT_PAIR_BLOCK, pt_trace_pair_lds(lds_levels));
    else if (any) (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_trace_pair<true>, PT_PAIR_BLOCK, pt_trace_pair_lds(lds_levels));
    else (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_trace_pair<false>, PT_PAIR_BLOCK, pt_trace_pair_lds(lds_levels));
    return nb > 0 ? nb : 2;
}

// `spill` must hold pt_trace_pair_spill_bytes(tree levels, grid x 256): every level for the rays that take the
// literal walk, the levels from lds_levels on for the others.
hipError_t pt_launch_trace_pair(bool any, int lds_levels, void *spill, int grid, hipStream_t stream, const DScene &sc,
                                const double *rays, long stride, const int *seg_counts, long seg, int n_val, int *out_prim,
                                double *out_t, double *out_b1, double *out_b2, const ShadowResolve &rs,
                                unsigned long long *counters, int refill_min, const int *order) {
    static bool attr_set[PT_MAX_DEVICES][2];
    const size_t lds = pt_trace_pair_lds(lds_levels);
    if (!seg_counts) order = nullptr;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= PT_MAX_DEVICES) return hipErrorInvalidDevice;
    if (!attr_set[dev][any ? 1 : 0]) {
        const void *fns[5] = {any ? reinterpret_cast<const void *>(&k_trace_pair<true>) : reinterpret_cast<const void *>(&k_trace_pair<false>),
                              any ? reinterpret_cast<const void *>(&k_trace_pair_refill<true, false, false>)
                                  : reinterpret_cast<const void *>(&k_trace_pair_refill<false, false, false>),
                              any ? reinterpret_cast<const void *>(&k_trace_pair_refill<true, true, false>)
                                  : reinterpret_cast<const void *>(&k_trace_pair_refill<false, true, false>),
                              any ? reinterpret_cast<const void *>(&k_trace_pair_refill<true, false, true>)
                                  : reinterpret_cast<const void *>(&k_trace_pair_refill<false, false, true>),
                              any ? reinterpret_cast<const void *>(&k_trace_pair_refill<true, true, true>)
                                  : reinterpret_cast<const void *>(&k_trace_pair_refill<false, true, true>)};
        for (const void *fn : fns) {
            hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) return e;
        }
        attr_set[dev][any ? 1 : 0] = true;
    }
    // lane refill needs the region cursors: segmented input of a launch with PT_TRACE_DYNAMIC
    const bool refill = seg_counts && (n_val & PT_TRACE_DYNAMIC) && (n_val & PT_TRACE_REFILL);
    if (refill_min < 1 || refill_min > 64) refill_min = PT_REFILL_MIN;
    const bool half = (n_val & PT_TRACE_HALF) && sc.pairs_h != nullptr;
#define PT_REFILL(A, H, S) hipLaunchKernelGGL((k_trace_pair_refill<A, H, S>), dim3(grid), dim3(PT_PAIR_BLOCK), lds, stream, sc, rays, stride, \
                               seg_counts, seg, out_prim, out_t, out_b1, out_b2, rs, counters, (PairEnt *)spill, lds_levels, refill_min, order)
#define PT_REFILL_S(A, H) do { if (order) PT_REFILL(A, H, true); else PT_REFILL(A, H, false); } while (0)
    if (refill) {
        if (any) { if (half) PT_REFILL_S(true, true); else PT_REFILL_S(true, false); }
        else { if (half) PT_REFILL_S(false, true); else PT_REFILL_S(false, false); }
#undef PT_REFILL_S
#undef PT_REFILL
    } else if (any)
        hipLaunchKernelGGL((k_trace_pair<true>), dim3(grid), dim3(PT_PAIR_BLOCK), lds, stream, sc, rays, stride, seg_counts,
                           seg, n_val, out_prim, out_t, out_b1, out_b2, rs, counters, (PairEnt *)spill, lds_levels, order);
    else
        hipLaunchKernelGGL((k_trace_pair<false>), dim3(grid), dim3(PT_PAIR_BLOCK), lds, stream, sc, rays, stride, seg_counts,
                           seg, n_val, out_prim, out_t, out_b1, out_b2, rs, counters, (PairEnt *)spill, lds_levels, order);
    return hipGetLastError();
}

// ---- ray binning for trees beyond the caches -----------------------------------------------------------------------
// k_sort_rays: one workgroup per window of PT_SORT_WINDOW consecutive entries of a queue region.  Every ray gets a key —
// the cell of its origin in a grid over the root box (Morton order) and the octant of its direction, arranged by `mode` —
// and the window's entries are sorted by (key, position) with a bitonic network in LDS; order[region start + k] = the
// region-local entry traced k-th.  Nothing moves: the trace kernels read rays and write results through `order`, the
// stages after them keep the original positions.  Which entry a lane traces is not observable (results and census are
// per ray), so the parity tests hold with the sort on and off.
#define PT_SORT_THREADS 256
__device__ __forceinline__ unsigned morton3(unsigned x, unsigned y, unsigned z) {       // 5 bits per axis
    unsigned m = 0;
    for (int b = 0; b < 5; b++) m |= (((x >> b) & 1u) << (3 * b)) | (((y >> b) & 1u) << (3 * b + 1)) | (((z >> b) & 1u) << (3 * b + 2));
    return m;
}
__global__ __launch_bounds__(PT_SORT_THREADS) void k_sort_rays(DScene sc, const double *__restrict__ rays, long stride,
                                                               const int *__restrict__ seg_counts, long seg, int window,
                                                               int mode, int *__restrict__ order) {
    extern __shared__ unsigned s_keys[];
    const int per_region = (int)((seg + window - 1) / window);
    const int region = blockIdx.x / per_region, w = blockIdx.x % per_region;
    const int n = seg_counts[region * PT_CNT_STRIDE];
    const int k0 = w * window;
    if (k0 >= n) return;                                   // wave-uniform: the whole workgroup leaves
    const int m = min(window, n - k0);
    const long first = (long)region * seg;
    // grid over the root box: 2^bits cells per axis
    const int bits = (mode >> 4) ? (mode >> 4) : 4;
    const double G = (double)(1 << bits);
    const double bx = sc.nodes[0].bmin[0], by = sc.nodes[0].bmin[1], bz = sc.nodes[0].bmin[2];
    const double ix = G / (sc.nodes[0].bmax[0] - bx), iy = G / (sc.nodes[0].bmax[1] - by), iz = G / (sc.nodes[0].bmax[2] - bz);
    const int idx_bits = 32 - __clz(window - 1);           // window is a power of two
    for (int k = threadIdx.x; k < window; k += PT_SORT_THREADS) {
        unsigned key = 0xFFFFFFFFu;
        if (k < m) {
            const long i = first + k0 + k;
            const double ox = rays[i], oy = rays[stride + i], oz = rays[2 * stride + i];
            const double dx = rays[3 * stride + i], dy = rays[4 * stride + i], dz = rays[5 * stride + i];
            const int hi = (1 << bits) - 1;
            const double fx = (ox - bx) * ix, fy = (oy - by) * iy, fz = (oz - bz) * iz;
            const unsigned cx = (unsigned)(fx > 0. ? (fx < (double)hi ? (int)fx : hi) : 0);     // NaN -> 0
            const unsigned cy = (unsigned)(fy > 0. ? (fy < (double)hi ? (int)fy : hi) : 0);
            const unsigned cz = (unsigned)(fz > 0. ? (fz < (double)hi ? (int)fz : hi) : 0);
            const unsigned mo = morton3(cx, cy, cz);
            const unsigned oct = (dx < 0. ? 1u : 0u) | (dy < 0. ? 2u : 0u) | (dz < 0. ? 4u : 0u);
            switch (mode & 15) {
            case 1: key = (oct << (3 * bits)) | mo; break;          // octant major
            case 2: key = mo; break;                                // origin only
            case 3: key = oct; break;                               // octant only
            default: key = (mo << 3) | oct; break;                  // origin cell, then octant
            }
            key = (key << idx_bits) | (unsigned)k;
        }
        s_keys[k] = key;
    }
    __syncthreads();
    for (int size = 2; size <= window; size <<= 1)
        for (int j = size >> 1; j > 0; j >>= 1) {
            for (int t = threadIdx.x; t < window / 2; t += PT_SORT_THREADS) {
                const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1)), l = i | j;
                const unsigned a = s_keys[i], b = s_keys[l];
                const bool up = (i & size) == 0;
                if ((a > b) == up) { s_keys[i] = b; s_keys[l] = a; }
            }
            __syncthreads();
        }
    for (int k = threadIdx.x; k < m; k += PT_SORT_THREADS)
        order[first + k0 + k] = k0 + (int)(s_keys[k] & (unsigned)(window - 1));
}
hipError_t pt_launch_sort_rays(hipStream_t stream, const DScene &sc, const double *rays, long stride, const int *seg_counts,
                               long seg, int window, int mode, int *order) {
    if (window < 64 || window > 16384 || (window & (window - 1))) return hipErrorInvalidValue;
    const int bits = (mode >> 4) ? (mode >> 4) : 4;
    if (bits < 1 || bits > 5 || 3 * bits + 3 + (32 - __builtin_clz(window - 1)) > 32) return hipErrorInvalidValue;
    const long per_region = (seg + window - 1) / window;
    hipLaunchKernelGGL(k_sort_rays, dim3((unsigned)(per_region * PT_SEG)), dim3(PT_SORT_THREADS), (size_t)window * sizeof(unsigned),
                       stream, sc, rays, stride, seg_counts, seg, window, mode, order);
    return hipGetLastError();
}

size_t pt_trace_meta_lds(int stack, int n_nodes) {
    return pt_trace_meta_table_bytes(n_nodes) + (size_t)stack * PT_TOP_BLOCK * sizeof(int);
}
size_t pt_trace_meta_s_lds(int stack, int n_nodes, bool packed) {
    return pt_trace_meta_table_bytes(n_nodes) / (packed ? 2 : 1) + (size_t)stack * (packed ? 256 : 512) * sizeof(unsigned short);
}

// packed: four 256-thread workgroups per CU, else two of 512; the grid in whole sets of PT_SEG waves
hipError_t pt_launch_trace_meta_s(bool any, bool packed, int stack, int grid, hipStream_t stream, const DScene &sc, const double *rays,
                                  long stride, const int *seg_counts, long seg, int n_val, int *out_prim, double *out_t,
                                  double *out_b1, double *out_b2, const ShadowResolve &rs, unsigned long long *counters) {
    static bool attr_set[PT_MAX_DEVICES];
    const size_t lds = pt_trace_meta_s_lds(stack, sc.n_nodes, packed);
    if (sc.n_nodes > 65535) return hipErrorInvalidValue;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= PT_MAX_DEVICES) return hipErrorInvalidDevice;
    if (!attr_set[dev]) {
        const void *fns[4] = {reinterpret_cast<const void *>(&k_trace_meta_s<true, 512, false>), reinterpret_cast<const void *>(&k_trace_meta_s<false, 512, false>),
                              reinterpret_cast<const void *>(&k_trace_meta_s<true, 256, true>), reinterpret_cast<const void *>(&k_trace_meta_s<false, 256, true>)};
        for (const void *fn : fns) {
            hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) return e;
        }
        attr_set[dev] = true;
    }
    if (seg_counts) grid = grid < 128 ? 64 : (grid / 64) * 64;
#define PT_METAS(A, B, P) hipLaunchKernelGGL((k_trace_meta_s<A, B, P>), dim3(grid), dim3(B), lds, stream, sc, rays, stride, seg_counts, \
                                             seg, n_val, out_prim, out_t, out_b1, out_b2, rs, counters)
    if (packed) { if (any) PT_METAS(true, 256, true); else PT_METAS(false, 256, true); }
    else { if (any) PT_METAS(true, 512, false); else PT_METAS(false, 512, false); }
#undef PT_METAS
    return hipGetLastError();
}

hipError_t pt_launch_trace_meta(bool any, int stack, int grid, hipStream_t stream, const DScene &sc, const double *rays,
                                long stride, const int *seg_counts, long seg, int n_val, int *out_prim, double *out_t,
                                double *out_b1, double *out_b2, const ShadowResolve &rs, unsigned long long *counters) {
    static bool attr_set[PT_MAX_DEVICES][2];
    const size_t lds = pt_trace_meta_lds(stack, sc.n_nodes);
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= PT_MAX_DEVICES) return hipErrorInvalidDevice;
    if (!attr_set[dev][any ? 1 : 0]) {
        hipError_t e = any ? hipFuncSetAttribute(reinterpret_cast<const void *>(&k_trace_meta<true>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)
                           : hipFuncSetAttribute(reinterpret_cast<const void *>(&k_trace_meta<false>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr_set[dev][any ? 1 : 0] = true;
    }
    if (seg_counts) {   // whole sets of PT_SEG waves
        const int per = PT_SEG / (PT_TOP_BLOCK / 64);
        grid = ((grid + per - 1) / per) * per;
    }
    if (any)
        hipLaunchKernelGGL((k_trace_meta<true>), dim3(grid), dim3(PT_TOP_BLOCK), lds, stream, sc, rays, stride, seg_counts,
                           seg, n_val, out_prim, out_t, out_b1, out_b2, rs, counters);
    else
        hipLaunchKernelGGL((k_trace_meta<false>), dim3(grid), dim3(PT_TOP_BLOCK), lds, stream, sc, rays, stride, seg_counts,
                           seg, n_val, out_prim, out_t, out_b1, out_b2, rs, counters);
    return hipGetLastError();
}

// LDS bytes of k_trace_top for a tree that needs `stack` todo levels and stages n_top nodes
size_t pt_trace_top_lds(int stack, int n_top) {
    return (size_t)n_top * sizeof(TopNode) + (size_t)stack * PT_TOP_BLOCK * sizeof(int);
}

hipError_t pt_launch_trace_top(bool any, int stack, int grid, hipStream_t stream, const DScene &sc, const double *rays,
                               long stride, const int *seg_counts, long seg, int n_val, int *out_prim, double *out_t,
                               double *out_b1, double *out_b2, const ShadowResolve &rs, unsigned long long *counters) {
    // the attribute belongs to the function ON THE CURRENT DEVICE: remember it per device (one process may
    // drive a context on each of the node's GPUs)
    static bool attr_set[PT_MAX_DEVICES][2];
    const size_t lds = pt_trace_top_lds(stack, sc.n_top);
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= PT_MAX_DEVICES) return hipErrorInvalidDevice;
    if (!attr_set[dev][any ? 1 : 0]) {
        hipError_t e = any ? hipFuncSetAttribute(reinterpret_cast<const void *>(&k_trace_top<true>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)
                           : hipFuncSetAttribute(reinterpret_cast<const void *>(&k_trace_top<false>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr_set[dev][any ? 1 : 0] = true;
    }
    if (seg_counts) {   // whole sets of PT_SEG waves
        const int per = PT_SEG / (PT_TOP_BLOCK / 64);
        grid = ((grid + per - 1) / per) * per;
    }
    if (any)
        hipLaunchKernelGGL((k_trace_top<true>), dim3(grid), dim3(PT_TOP_BLOCK), lds, stream, sc, rays, stride, seg_counts,
                           seg, n_val, out_prim, out_t, out_b1, out_b2, rs, counters);
    else
        hipLaunchKernelGGL((k_trace_top<false>), dim3(grid), dim3(PT_TOP_BLOCK), lds, stream, sc, rays, stride, seg_counts,
                           seg, n_val, out_prim, out_t, out_b1, out_b2, rs, counters);
    return hipGetLastError();
}

template <bool ANY>
static hipError_t launch_trace_t(int stack, int grid, hipStream_t stream, const DScene &sc, const double *rays,
                                 long stride, const int *seg_counts, long seg, int n_val, int *out_prim, double *out_t,
                                 double *out_b1, double *out_b2, const ShadowResolve &rs,
                                 unsigned long long *counters) {
#define PT_LAUNCH(S)                                                                                       \
    hipLaunchKernelGGL((k_trace<S, ANY>), dim3(grid), dim3(PT_BLOCK), 0, stream, sc, rays, stride, seg_counts, \
                       seg, n_val, out_prim, out_t, out_b1, out_b2, rs, counters)
    if (stack <= 16) PT_LAUNCH(16);
    else if (stack <= 24) PT_LAUNCH(24);
    else if (stack <= 32) PT_LAUNCH(32);
    else if (stack <= 48) PT_LAUNCH(48);
    else PT_LAUNCH(64);
#undef PT_LAUNCH
    return hipGetLastError();
}

// Resident workgroups per CU of the trace kernel that pt_launch_trace would pick (advisory
// occupancy API; used only to size the grid to one full residency round, never for correctness).
template <bool ANY>
static int trace_occupancy_t(int stack) {
    int nb = 0;
#define PT_OCC(S) (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_trace<S, ANY>, PT_BLOCK, 0)
    if (stack <= 16) PT_OCC(16);
    else if (stack <= 24) PT_OCC(24);
    else if (stack <= 32) PT_OCC(32);
    else if (stack <= 48) PT_OCC(48);
    else PT_OCC(64);
#undef PT_OCC
    return nb > 0 ? nb : 4;
}
int pt_trace_occupancy(bool any, int stack) {
    return any ? trace_occupancy_t<true>(stack) : trace_occupancy_t<false>(stack);
}

hipError_t pt_launch_trace(bool any, int stack, int grid, hipStream_t stream, const DScene &sc, const double *rays,
                           long stride, const int *seg_counts, long seg, int n_val, int *out_prim, double *out_t,
                           double *out_b1, double *out_b2, const ShadowResolve &rs, unsigned long long *counters) {
    if (seg_counts) grid = ((grid + PT_SEG_BLOCKS - 1) / PT_SEG_BLOCKS) * PT_SEG_BLOCKS;   // whole sets of PT_SEG waves
    if (any) return launch_trace_t<true>(stack, grid, stream, sc, rays, stride, seg_counts, seg, n_val, out_prim, out_t, out_b1, out_b2, rs, counters);
    return launch_trace_t<false>(stack, grid, stream, sc, rays, stride, seg_counts, seg, n_val, out_prim, out_t, out_b1, out_b2, rs, counters);
}

// ---- coherent chunks: one walk per wave ---------------------------------------------------------------------------
// k_trace_packet: the 64 rays of a chunk of camera rays (the 64 samples of a pixel) or of their shadow rays visit
// nearly the same nodes.  Here the WAVE walks the tree once: a stack of (node, lane mask) entries in LDS, the node and
// triangle records fetched with wave-uniform addresses (scalar loads: no vector-memory request at all), every lane of an
// entry's mask testing the box with its own ray.  Each lane still makes exactly the visits of its own walk, in its own
// order (bvh.py:322-410): lanes that disagree on dir_is_neg[axis] are split into two masks, the far child of either
// group is stacked under the near children, and a far entry is box-tested when it is popped, with the lane's maxt of
// that moment.  So hit / prim / t / b1 / b2 and the census are those of the per-lane kernels, bit for bit (tested).
// An entry is dropped when none of its lanes is still walking (any hit: lanes leave at their first hit).
// Stack depth: an interior node pops one entry and pushes at most four, the top one is popped next: <= 3 entries per
// level of the current path.
struct PacketEnt { int node; unsigned lo, hi; int pad; };
// Loads through the constant address space: with a wave-uniform address they are scalar loads whatever the compiler can
// prove about aliasing stores (the any-hit form's atomics on L made it fall back to vector loads for the triangles).
// The scene arrays are never written while a kernel runs.
typedef double pk_d2 __attribute__((ext_vector_type(2)));
typedef int pk_i2 __attribute__((ext_vector_type(2)));
typedef const __attribute__((address_space(4))) pk_d2 *pk_cd2;
typedef const __attribute__((address_space(4))) pk_i2 *pk_ci2;
__device__ __forceinline__ double2 pk_load(const void *base, int i) {
    const pk_d2 v = ((pk_cd2)(unsigned long long)base)[i];
    return make_double2(v.x, v.y);
}
// waves per SIMD (head mesh, level-0 launches of the 1024^2 x 64 spp render, ms): closest hit 4 / 5 / 6 / 8 -> 5.48 / 5.69 /
// 5.15 (79 VGPRs, 12 B of scratch) / 11.9 (64 VGPRs, 72 B); any hit 5 / 6 / 8 -> 6.94 / 6.95 / 6.39 (60 VGPRs, 36 B).  One
// walk per wave is a serial chain (scalar load -> test -> ballot -> stack): more waves hide it until the spills bite.
#ifndef PT_PACKET_WAVES
#define PT_PACKET_WAVES(ANY) ((ANY) ? 8 : 6)
#endif
template <bool ANY>
__global__ __launch_bounds__(PT_BLOCK, PT_PACKET_WAVES(ANY)) void k_trace_packet(DScene sc, const double *__restrict__ rays, long stride,
                                                           const int *__restrict__ seg_counts, long seg,
                                                           int *__restrict__ out_prim, double *__restrict__ out_t,
                                                           double *__restrict__ out_b1, double *__restrict__ out_b2,
                                                           ShadowResolve rs, unsigned long long *counters, int cap) {
    extern __shared__ __attribute__((aligned(16))) unsigned char s_mem[];
    PacketEnt *stk = reinterpret_cast<PacketEnt *>(s_mem) + (size_t)(threadIdx.x >> 6) * cap;
    unsigned nvis = 0, ntri = 0, nray = 0;
    const TraceOut out = {out_prim, out_t, out_b1, out_b2, rs};
    const int lane = threadIdx.x & 63;
    const int wave = (int)((blockIdx.x * PT_BLOCK + threadIdx.x) >> 6);
    const int nsub = (int)((gridDim.x * (PT_BLOCK / 64)) / PT_SEG);
    const int region = wave % PT_SEG;
    if (wave / PT_SEG < nsub) {
        const int n = seg_counts[region * PT_CNT_STRIDE];
        const long first = (long)region * seg;
        int *cursor = const_cast<int *>(seg_counts) + region * PT_CNT_STRIDE + PT_CNT_CURSOR;
        for (;;) {
            const int base = __builtin_amdgcn_readfirstlane(lane == 0 ? atomicAdd(cursor, 64) : 0);
            if (base >= n) break;
            const int k = base + lane;
            const bool valid = k < n;
            Ray r;
            r.o = r.d = mk3(0, 0, 0); r.mint = 0.; r.maxt = 0.;
            if (valid) r = load_ray(rays, stride, first + k);
            const SlabRay sv = slab_ray(sc, r);
            double b1 = 0., b2 = 0.;
            int res = ANY ? 0 : -1;
            unsigned long long alive = __ballot(valid);
            // the entry being visited lives in registers (cur_node, cur_m); the others wait on the LDS stack
            int sp = 0;
            int cur_node = 0;
            unsigned long long cur_m = sc.n_nodes > 0 ? alive : 0ull;
            for (;;) {
                if (cur_m == 0ull) {
                    if (sp == 0) break;
                    sp--;
                    const PacketEnt e = stk[sp];
                    cur_node = __builtin_amdgcn_readfirstlane(e.node);
                    cur_m = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)e.hi) << 32) |
                            (unsigned)__builtin_amdgcn_readfirstlane((int)e.lo);
                }
                const int node = cur_node;
                const unsigned long long m = cur_m & alive;
                cur_m = 0ull;
                if (m == 0ull) continue;
                const GNode *np = sc.nodes + node;                                       // wave-uniform: scalar loads
                const double2 n0 = pk_load(np, 0), n1 = pk_load(np, 1), n2 = pk_load(np, 2);
                const pk_i2 mv = ((pk_ci2)(unsigned long long)np)[6];
                const int2 meta = make_int2(mv.x, mv.y);
                bool hb = false;
                if ((m >> lane) & 1ull) {
                    nvis++;
                    hb = sv.fast ? slab_hit<true>(n0, n1, n2, r, sv) : slab_hit<false>(n0, n1, n2, r, sv);
                }
                const unsigned long long hm = __ballot(hb);
                if (hm == 0ull) continue;
                const int nprim = meta.y & 0xFFFF;
                if (nprim > 0) {
                    for (int i = 0; i < nprim; i++) {
                        const GTri *tp = sc.tris + meta.x + i;
                        const double2 a0 = pk_load(tp, 0), a1 = pk_load(tp, 1), a2 = pk_load(tp, 2), a3 = pk_load(tp, 3),
                                      a4 = pk_load(tp, 4);
                        // the test runs for every lane (the others' outcome is dropped): under `if (hb)` the compiler
                        // sinks the record's loads into the divergent block and makes them vector loads
                        double t, u1, u2;
                        const bool th = tri_math(a0, a1, a2, a3, a4, r, t, u1, u2);
                        if (hb) {
                            ntri++;
                            if (th) {
                                if (ANY) { res = 1; hb = false; }      // intersect_p returns at its first hit
                                else { res = meta.x + i; b1 = u1; b2 = u2; r.maxt = t; }
                            }
                        }
                    }
                    if (ANY) alive = __ballot(valid && res == 0);
                } else {
                    const bool neg = (sv.negmask >> (meta.y >> 16)) & 1;         // dir_is_neg[node.axis], bvh.py:350
                    const unsigned long long negm = __ballot(hb && neg), posm = hm & ~negm;
                    const int c1 = node + 1, c2 = meta.x;
                    PacketEnt p;
                    p.pad = 0;
#ifdef PT_DEBUG_ASSERT
                    assert(sp + 3 <= cap);      // <= 3 entries per interior level of the current path (see `cap` below)
#endif
                    // far children under near children; the two groups share no lane, so their order is free.  The
                    // topmost entry stays in registers.
                    if (posm) { p.node = c2; p.lo = (unsigned)posm; p.hi = (unsigned)(posm >> 32); stk[sp++] = p; }
                    if (negm) { p.node = c1; p.lo = (unsigned)negm; p.hi = (unsigned)(negm >> 32); stk[sp++] = p; }
                    if (posm) {
                        if (negm) { p.node = c2; p.lo = (unsigned)negm; p.hi = (unsigned)(negm >> 32); stk[sp++] = p; }
                        cur_node = c1; cur_m = posm;
                    } else {
                        cur_node = c2; cur_m = negm;
                    }
                }
            }
            if (valid) {
                store_result<ANY>(out, first + k, res, r, b1, b2);
                nray++;
            }
        }
    }
    if (counters) trace_count<ANY>(counters, nvis, ntri, nray);
}

size_t pt_trace_packet_lds(int stack) { return (size_t)(PT_BLOCK / 64) * (size_t)(3 * stack + 8) * sizeof(PacketEnt); }
int pt_trace_packet_occupancy(bool any, int stack) {
    int nb = 0;
    if (any) (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_trace_packet<true>, PT_BLOCK, pt_trace_packet_lds(stack));
    else (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_trace_packet<false>, PT_BLOCK, pt_trace_packet_lds(stack));
    return nb > 0 ? nb : 4;
}
// segmented input with chunk cursors only (a render's level-0 launches); `stack` = the tree's depth bound
hipError_t pt_launch_trace_packet(bool any, int stack, int grid, hipStream_t stream, const DScene &sc, const double *rays,
                                  long stride, const int *seg_counts, long seg, int *out_prim, double *out_t, double *out_b1,
                                  double *out_b2, const ShadowResolve &rs, unsigned long long *counters) {
    if (!seg_counts) return hipErrorInvalidValue;
    grid = ((grid + PT_SEG_BLOCKS - 1) / PT_SEG_BLOCKS) * PT_SEG_BLOCKS;
    // `stack` = pt_ctx::stack_need = the deepest leaf's depth as check_tree measured it on the uploaded tree (<= 64,
    // pt_scene_upload refuses deeper ones): a path holds at most that many interior nodes, each leaves at most three
    // entries on the stack (header comment), so 3 x stack + 8 entries cannot overflow.  -DPT_DEBUG_ASSERT checks it.
    if (stack < 1 || stack > 64) return hipErrorInvalidValue;
    const int cap = 3 * stack + 8;
    if (any)
        hipLaunchKernelGGL((k_trace_packet<true>), dim3(grid), dim3(PT_BLOCK), pt_trace_packet_lds(stack), stream, sc, rays,
                           stride, seg_counts, seg, out_prim, out_t, out_b1, out_b2, rs, counters, cap);
    else
        hipLaunchKernelGGL((k_trace_packet<false>), dim3(grid), dim3(PT_BLOCK), pt_trace_packet_lds(stack), stream, sc, rays,
                           stride, seg_counts, seg, out_prim, out_t, out_b1, out_b2, rs, counters, cap);
    return hipGetLastError();
}

// The shading frames of all triangles, once per scene upload (pt_device.h tri_frame).
__global__ __launch_bounds__(PT_BLOCK) void k_tri_frames(DScene sc, double *__restrict__ frames) {
    const int i = blockIdx.x * PT_BLOCK + threadIdx.x;
    if (i >= sc.n_tris) return;
    d3 nn, sn, tn;
    tri_frame<true>(sc, i, nn, sn, tn);
    double *f = frames + 9 * (size_t)i;
    f[0] = nn.x; f[1] = nn.y; f[2] = nn.z; f[3] = sn.x; f[4] = sn.y; f[5] = sn.z; f[6] = tn.x; f[7] = tn.y; f[8] = tn.z;
}
hipError_t pt_launch_tri_frames(hipStream_t stream, const DScene &sc, double *frames) {
    if (sc.n_tris <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_tri_frames, dim3((sc.n_tris + PT_BLOCK - 1) / PT_BLOCK), dim3(PT_BLOCK), 0, stream, sc, frames);
    return hipGetLastError();
}

// =========================================================================================
// sample streams
// =========================================================================================

// Fisher–Yates on a nibble-packed permutation of n <= 16 (DESIGN.md "sample streams")
__device__ inline unsigned long long lhs_perm(int n, uint32_t sid, uint32_t key) {
    unsigned long long p = 0xFEDCBA9876543210ull;
    for (int i = n - 1; i > 0; i--) {
        int j = (int)floor(hash_uniform(sid, key + (uint32_t)i) * (i + 1));
        if (j > i) j = i;
        unsigned long long a = (p >> (4 * i)) & 15ull, b = (p >> (4 * j)) & 15ull;
        p &= ~((15ull << (4 * i)) | (15ull << (4 * j)));
        p |= (b << (4 * i)) | (a << (4 * j));
    }
    return p;
}

struct SampleSrc {
    const double *row;   // parity: table row; production: nullptr
    uint32_t msid;       // seed-mixed sample ordinal (production) / raw ordinal (parity rng)
};

// value of flattened sample dimension k >= 5 for the PATH layout (all arrays have n = 1)
__device__ __forceinline__ double path_dim(const SampleSrc &s, int k) {
    return s.row ? s.row[k] : hash_uniform(s.msid, 0x10000u + (uint32_t)k);
}

// LightSample.u_pos of sample j of a Direct-lighting 2D array of n entries starting at `off`
// (latin_hypercube_2d, sampler/utility.py:41-44: x stratum = j, y stratum = shuffled)
__device__ inline void direct_light_uv(const SampleSrc &s, int off, int n, int j, unsigned long long perm,
                                       double &u0, double &u1) {
    if (s.row) { u0 = s.row[off + 2 * j]; u1 = s.row[off + 2 * j + 1]; return; }
    u0 = (j + hash_uniform(s.msid, 0x10000u + (uint32_t)(off + 2 * j))) / n;
    u1 = ((int)((perm >> (4 * j)) & 15ull) + hash_uniform(s.msid, 0x10000u + (uint32_t)(off + 2 * j + 1))) / n;
}

__device__ __forceinline__ void tile_decode(const RenderConst &rc, long out_i, int &px, int &py, int &si, uint32_t &sid) {
    // 32-bit unsigned divisions: pt_render refuses windows of more than 2^32 samples, and a 64-bit
    // divide costs ~4x the instructions of a 32-bit one on this VALU
    const uint32_t ns = (uint32_t)(rc.tile.s1 - rc.tile.s0), tw = (uint32_t)(rc.tile.x1 - rc.tile.x0);
    const uint32_t o = (uint32_t)out_i;
    const uint32_t pix = o / ns;
    si = rc.tile.s0 + (int)(o - pix * ns);
    const uint32_t row = pix / tw;
    py = pt_tile_row_y(rc.tile, (int)row);
    px = rc.tile.x0 + (int)(pix - row * tw);
    const int spp = rc.sp.xs * rc.sp.ys, W = rc.sp.x_end - rc.sp.x_start;
    sid = (uint32_t)((((long)(py - rc.sp.y_start) * W + (px - rc.sp.x_start)) * spp) + si);
}

__device__ __forceinline__ SampleSrc sample_src(const RenderConst &rc, uint32_t sid) {
    SampleSrc s;
    s.row = rc.table ? rc.table + (size_t)sid * rc.lay.dims : nullptr;
    s.msid = rc.table ? sid : mix_seed(sid, rc.sp.seed);
    return s;
}

// =========================================================================================
// K1 raygen: StratifiedSampler.generate (sampler/sampler/stratified.py:45-89) +
// PerspectiveCamera.generate_ray_differential (camera/projective.py:178-232; o and d only —
// ray differentials feed textures and every supported texture is constant).
// =========================================================================================
__device__ __forceinline__ bool ne_unity(double x) { return x < 1. - PT_EPS || x > 1. + PT_EPS; }
// Transform.__call__ on a Point with the perspective divide, transform/__init__.py:55-60
__device__ inline d3 xf4_point(const double *m, d3 p) {
    double r0 = (m[0] * p.x + m[1] * p.y + m[2] * p.z) + m[3];
    double r1 = (m[4] * p.x + m[5] * p.y + m[6] * p.z) + m[7];
    double r2 = (m[8] * p.x + m[9] * p.y + m[10] * p.z) + m[11];
    double r3 = (m[12] * p.x + m[13] * p.y + m[14] * p.z) + m[15];
    d3 r = mk3(r0, r1, r2);
    if (ne_unity(r3)) r = r / r3;
    return r;
}
// concentric_sample_disk, montecarlo/utility.py:122-168
__device__ inline void concentric_sample_disk(double u1, double u2, double &ox, double &oy) {
    double sx = 2. * u1 - 1., sy = 2. * u2 - 1., r, theta;
    if (sx == 0. && sy == 0.) { ox = 0.; oy = 0.; return; }
    if (sx > -sy) {
        if (sx > sy) { r = sx; theta = sy / sx; if (sy <= 0.) theta += 8.; }
        else { r = sy; theta = 2. - sx / sy; }
    } else {
        if (sx > sy) { r = -sy; theta = 6. + sx / sy; }
        else { r = -sx; theta = 4. - sy / sx; }
    }
    theta *= PT_PI / 4.;
    ox = r * cos(theta); oy = r * sin(theta);
}

// Region r's line of a counter block back to "empty": fill count, chunk cursor, back count (pt_kernels.h).
__device__ __forceinline__ void pt_clear_region_line(int *__restrict__ block, int r) {
    block[r * PT_CNT_STRIDE] = 0;
    block[r * PT_CNT_STRIDE + PT_CNT_CURSOR] = 0;
    block[r * PT_CNT_STRIDE + PT_CNT_BACK] = 0;
    block[r * PT_CNT_STRIDE + PT_CNT_CURSOR2] = 0;
}

// Where camera sample i of a batch goes in the segmented queue (pt_kernels.h): samples travel in
// groups of G = 2^gshift 64-sample chunks (consecutive pixels: the waves that serve a region work
// on neighbouring pixels at any one time: +6 % on the whole render over a chunk-interleaved mapping);
// group g of round j = g / PT_SEG goes to region (g + 17 j) % PT_SEG — a rotation per round, so
// every region receives one group per round (the capacity bound) and image columns are spread
// over all regions (the load balance).
__device__ __forceinline__ long seg_entry(int i, int gshift, long seg) {
    const int ch = i >> 6, g = ch >> gshift, j = g / PT_SEG;
    const int r = (g + 17 * j) % PT_SEG;
    return (long)r * seg + ((((long)j << gshift) + (ch & ((1 << gshift) - 1))) << 6) + (i & 63);
}

__global__ __launch_bounds__(PT_BLOCK) void k_raygen(const RenderConst *__restrict__ rcp, long base, int nb, int gshift,
                                                     Queue q, PathState st, unsigned long long *counters,
                                                     int *__restrict__ clr_hits) {
    const RenderConst &rc = *rcp;
    const int i = blockIdx.x * PT_BLOCK + threadIdx.x;
    if (blockIdx.x == 0 && threadIdx.x < PT_SEG) {
        // fill level of every region, in closed form: whole groups, except the batch's last one
        const int r = threadIdx.x, gsz = 64 << gshift;
        const int ngroups = (nb + gsz - 1) / gsz, J = ngroups / PT_SEG, R = ngroups % PT_SEG;
        const int b = ((r - 17 * J) % PT_SEG + PT_SEG) % PT_SEG;      // position of region r in the partial round
        int cnt = (J + (b < R ? 1 : 0)) * gsz;
        const int gl = ngroups - 1, jl = gl / PT_SEG;
        if (r == (gl + 17 * jl) % PT_SEG) cnt -= ngroups * gsz - nb;
        q.count[r * PT_CNT_STRIDE] = cnt;
        q.count[r * PT_CNT_STRIDE + PT_CNT_CURSOR] = 0;      // the chunk cursor of a dynamic trace launch (pt_kernels.h)
        // the hit list of level 0 (count, cursor, back count): nothing reads that block between the last k_shade
        // that used it and this batch's first k_list_hits, so no memset node is needed for it (pt_api.hip)
        if (clr_hits) pt_clear_region_line(clr_hits, r);
    }
    if (i >= nb) return;
    int px, py, si;
    uint32_t sid;
    tile_decode(rc, base + i, px, py, si, sid);
    SampleSrc ss = sample_src(rc, sid);
    double ix, iy, lu, lv;
    if (ss.row) { ix = ss.row[0]; iy = ss.row[1]; lu = ss.row[2]; lv = ss.row[3]; }
    else {
        const int sx = si % rc.sp.xs, sy = si / rc.sp.xs;
        double jx = rc.sp.jitter ? hash_uniform(ss.msid, 0x10000u + 0) : .5;
        double jy = rc.sp.jitter ? hash_uniform(ss.msid, 0x10000u + 1) : .5;
        ix = (sx + jx) / rc.sp.xs + px;
        iy = (sy + jy) / rc.sp.ys + py;
        lu = hash_uniform(ss.msid, 0x10000u + 2);
        lv = hash_uniform(ss.msid, 0x10000u + 3);
    }
    d3 p_cam = xf4_point(rc.cam.r2c, mk3(ix, iy, 0.));
    d3 o = mk3(0, 0, 0);
    d3 d = normalize3(p_cam);
    if (rc.cam.kind == PT_CAM_ORTHO) {   // OrthoCamera.generate_ray, camera/projective.py:80-108
        o = p_cam;
        d = mk3(0., 0., 1.);
    }
    if (rc.cam.lens_rad > 0.) {
        double a, b;
        concentric_sample_disk(lu, lv, a, b);
        a *= rc.cam.lens_rad; b *= rc.cam.lens_rad;
        double ft = rc.cam.focal_dist / d.z;
        d3 Pfoc = o + d * ft;
        o = mk3(a, b, 0.);
        d = normalize3(Pfoc - o);
    }
    o = xf4_point(rc.cam.c2w, o);
    d = xf_vec(rc.cam.c2w, d);
    if (rc.cam.kind == PT_CAM_ORTHO) {
        // OrthoCamera.generate_ray_differential (projective.py:110-125) transforms the ray that
        // generate_ray already took to world space by c2w a SECOND time (Q14)
        o = xf4_point(rc.cam.c2w, o);
        d = xf_vec(rc.cam.c2w, d);
    }
    const long S = q.stride;
    const long e = seg_entry(i, gshift, q.seg);
    q.rays[e] = o.x; q.rays[S + e] = o.y; q.rays[2 * S + e] = o.z;
    q.rays[3 * S + e] = d.x; q.rays[4 * S + e] = d.y; q.rays[5 * S + e] = d.z;
    q.rays[6 * S + e] = 0.; q.rays[7 * S + e] = __builtin_inf();
    q.slot[e] = i;
    const long P = st.stride;
    // st.T is not initialised: level 0 shades with T = 1 and writes it with the first continuation ray
    st.L[i] = 0.; st.L[P + i] = 0.; st.L[2 * P + i] = 0.;
    st.esc[i] = 0;
    st.img[i] = ix; st.img[P + i] = iy;
    if (counters && threadIdx.x == 0) {
        int cnt = nb - blockIdx.x * PT_BLOCK;
        atomicAdd(counters + (size_t)(blockIdx.x % PT_CNT_SHARDS) * CNT_COUNT + CNT_CAMERA_SAMPLES,
                  (unsigned long long)(cnt > PT_BLOCK ? PT_BLOCK : cnt));
    }
}

// =========================================================================================
// K4 shade (+K5 light sampling).  One lane per entry of the input queue (a traced ray and
// its hit record).  Emits shadow rays with their pre-weighted contribution, and at most one
// continuation ray; both queues are built with wave ballot + prefix sum (one atomic/wave).
// =========================================================================================

// estimate_direct AS IT EXECUTES (integrator/utility.py:203-257; only the light-sampling half
// contributes, Q3).  Returns true if a shadow ray must be traced; c = contribution if unoccluded.
template <bool EXT, typename R, int LOBES>
__device__ inline bool estimate_direct(const DScene &sc, const pt_light &lt, const VertexT<R> &vx, v3<R> wo, R u0,
                                       R u1, Ray &shadow, d3 &c) {
    const unsigned flags = BDF_ALL & ~BDF_SPECULAR;
    LightSampleT<R> ls;
    if (lt.kind == PT_LIGHT_DIFFUSE_SPHERE) sphere_light_sample_l<R>(lt, vx.p, vx.rEps, u0, u1, ls);
    else delta_light_sample_l<R>(lt, vx.p, vx.rEps, ls);
    if (!(ls.pdf > R(0) && !util_is_black(ls.Li))) return false;
    R bp = R(0);
    v3<R> f = (lt.kind == PT_LIGHT_DIFFUSE_SPHERE) ? bsdf_f_pdf<EXT, R, LOBES>(vx, wo, ls.wi, flags, bp) : bsdf_f<EXT, R, LOBES>(vx, wo, ls.wi, flags);
    if (util_is_black(f)) return false;
    R adot = r_abs(dot3(ls.wi, vx.nn));
    if (lt.kind != PT_LIGHT_DIFFUSE_SPHERE) {
        c = cvt3<double>((f * ls.Li) * (adot / ls.pdf));
    } else {
        R ff = 1 * ls.pdf, g = 1 * bp;
        R weight = ff * ff / (ff * ff + g * g);   // power_heuristic, montecarlo/utility.py:33-42
        c = cvt3<double>((f * ls.Li) * (adot * weight / ls.pdf));
    }
    shadow = ls.shadow;
    return true;
}

// One light of WhittedIntegrator.li (integrator/surface/whitted.py:44-53): no MIS, bsdf.f with the
// default flags, Spectrum.is_black tests.  Returns true if a shadow ray must be traced.
template <bool EXT, typename R, int LOBES>
__device__ inline bool estimate_whitted(const DScene &sc, const pt_light &lt, const VertexT<R> &vx, v3<R> wo, R u0,
                                        R u1, Ray &shadow, d3 &c) {
    LightSampleT<R> ls;
    if (lt.kind == PT_LIGHT_DIFFUSE_SPHERE) sphere_light_sample_l<R>(lt, vx.p, vx.rEps, u0, u1, ls);
    else delta_light_sample_l<R>(lt, vx.p, vx.rEps, ls);
    if (spec_is_black(ls.Li) || ls.pdf == R(0)) return false;
    v3<R> f = bsdf_f<EXT, R, LOBES>(vx, wo, ls.wi, BDF_ALL);
    if (spec_is_black(f)) return false;
    c = cvt3<double>((((f * ls.Li) * r_abs(dot3(ls.wi, vx.nn))) * R(1)) / ls.pdf);   // x transmittance (= 1) / pdf
    shadow = ls.shadow;
    return true;
}

__device__ __forceinline__ void push_shadow(const ShadowQueue &sh, int region, bool pred, const Ray &r, d3 c,
                                            int slot) {
    long pos = wave_append(pred, sh.count + region * PT_CNT_STRIDE);
    if (pos >= 0) {
        pos += (long)region * sh.seg;
        const long S = sh.stride;
        sh.rays[pos] = r.o.x; sh.rays[S + pos] = r.o.y; sh.rays[2 * S + pos] = r.o.z;
        sh.rays[3 * S + pos] = r.d.x; sh.rays[4 * S + pos] = r.d.y; sh.rays[5 * S + pos] = r.d.z;
        sh.rays[6 * S + pos] = r.mint; sh.rays[7 * S + pos] = r.maxt;
        sh.c[pos] = c.x; sh.c[S + pos] = c.y; sh.c[2 * S + pos] = c.z;
        sh.slot[pos] = slot;
    }
}

// KIND 0: DirectLightingIntegrator.li (integrator/surface/direct.py:70-105) as a loop over the
//   specular-reflection chain (specular_transmit contributes exactly 0, Q5, and is not traced;
//   specular_reflect, integrator/utility.py:120-155, recurses through renderer.li, which adds
//   Σ light.le on a miss, renderer/renderer.py:58-75).  `level` = ray.depth.
// KIND 1: PathIntegrator.li AS IT EXECUTES (integrator/surface/path.py:44-144; `specular_bounce`
//   always True after the first sample_f, Q4).  `level` = bounce_cnt.
#ifndef PT_SHADE_WAVES
#define PT_SHADE_WAVES 2
#endif
#ifndef PT_SHADE_WAVES_F32
#define PT_SHADE_WAVES_F32 3      // measured: 2 / 3 / 4 waves per SIMD -> shade 24.9 / 23.9 / 26.2 ms (overlapped timing)
#endif
#define PT_SHADE_LDS_MATS 64      // 34 KB
#define PT_SHADE_LDS_LIGHTS 16    // 4.5 KB (pt_scene_upload's limit)
static_assert(sizeof(pt_material) % 8 == 0 && sizeof(pt_light) % 8 == 0, "copied to LDS in 8-byte words");
// EXT (compile-time, so that plain scenes run the kernel unchanged): 0 = plain; 1 = the scene has textured lobes or
// mesh uvs; 2 = it has measured BRDF lobes as well or instead
// R: binary64 (parity; the reference's arithmetic) or binary32 (production only: BSDF / light arithmetic, pt_device.h)
// LOBES: the lobe types compiled in (pt_device.h lobe_type).  PART: which of the region's two hit lists this launch walks —
// 0 both (one class, or scenes listed in one list), 1 the front list, 2 the back list (its own cursor: PT_CNT_CURSOR2).
#ifndef PT_SHADE_WAVES_DIFFUSE
#define PT_SHADE_WAVES_DIFFUSE 2      // 205-226 VGPRs, no scratch; at 3 waves per SIMD (168 VGPRs) it spills 140-220 bytes: shade 14.0 -> 15.1 ms
#endif
template <int KIND, int EXT, typename R, int LOBES>
__global__ __launch_bounds__(PT_BLOCK, (LOBES != PT_LOBES_ALL ? PT_SHADE_WAVES_DIFFUSE : sizeof(R) == 4 ? PT_SHADE_WAVES_F32 : PT_SHADE_WAVES)) void k_shade(DScene sc, const RenderConst *__restrict__ rcp, long base,
                                                    int level, Queue qin, HitList hl, Queue qout, ShadowQueue sh,
                                                    PathState st, int part) {
    const RenderConst &rc = *rcp;
    // measured BRDF tables: the grid walk of measured_f is a chain of dependent loads (cell start -> samples), and at
    // 2 waves/SIMD nothing hides their latency: from LDS the 128 spp C5 render shades 2.5x faster than from L2
    // materials and lights are read field by field all through the vertex code; at 2 waves/SIMD nothing hides those
    // loads, so scenes with few materials keep them in LDS (shade 19.8 -> 17.4 ms per step on the head mesh).  Not in
    // the measured-BRDF variant, whose table already takes the LDS two blocks per CU can have.
    __shared__ pt_material s_mats[PT_SHADE_LDS_MATS];
    __shared__ pt_light s_lights[PT_SHADE_LDS_LIGHTS];
    const pt_material *mats_p = sc.mats;
    const pt_light *lights_p = sc.lights;
    if (EXT != 2 && sc.n_mats <= PT_SHADE_LDS_MATS && sc.n_lights <= PT_SHADE_LDS_LIGHTS) {
        const int nw = sc.n_mats * (int)(sizeof(pt_material) / 8), nl = sc.n_lights * (int)(sizeof(pt_light) / 8);
        for (int k = threadIdx.x; k < nw; k += PT_BLOCK)
            reinterpret_cast<double *>(s_mats)[k] = reinterpret_cast<const double *>(sc.mats)[k];
        for (int k = threadIdx.x; k < nl; k += PT_BLOCK)
            reinterpret_cast<double *>(s_lights)[k] = reinterpret_cast<const double *>(sc.lights)[k];
        __syncthreads();
        mats_p = s_mats;
        lights_p = s_lights;
    }
    const char *brdf = sc.brdf;
    int *hits = nullptr;
    if (EXT == 2) {
        extern __shared__ __attribute__((aligned(16))) double s_shade[];    // [hit lists][blob copy]
        hits = reinterpret_cast<int *>(s_shade) + threadIdx.x;
        if (sc.brdf_lds_bytes > 0) {
            double *dst = s_shade + PT_MEAS_HITS * PT_BLOCK / 2;
            const double *src = reinterpret_cast<const double *>(sc.brdf);
            for (int k = threadIdx.x; k < sc.brdf_lds_bytes / 8; k += PT_BLOCK) dst[k] = src[k];
            __syncthreads();
            brdf = reinterpret_cast<const char *>(dst);
        }
    }
    // wave w serves region w % PT_SEG of the input queue and appends to the SAME region of both
    // output queues (pt_kernels.h "segmented queues")
    const int wave = (int)((blockIdx.x * PT_BLOCK + threadIdx.x) >> 6);
    const int nsub = (int)((gridDim.x * (PT_BLOCK / 64)) / PT_SEG);
    const int region = wave % PT_SEG, sub = wave / PT_SEG;
    if (sub >= nsub) return;
    // the region's hits (k_list_hits): n_front listed from the region's start, n_back from its end (scenes with measured
    // lobes: front = hits on such materials).  The loop runs over a virtual index: the front list padded to whole chunks,
    // then the back list, so that no chunk mixes the two.
    const int n_front = hl.count[region * PT_CNT_STRIDE], n_back = hl.count[region * PT_CNT_STRIDE + PT_CNT_BACK];
    const int n_fpad = (n_front + 63) & ~63;
    const int nloop = n_fpad + ((n_back + 63) & ~63);   // whole waves stay convergent for the ballots
    const int lo = part == 2 ? n_fpad : 0, hi = part == 1 ? n_fpad : nloop;
    const long first = (long)region * qin.seg;
    // chunk kb of the region's hit list: the static stride, or (hl.dynamic) the next claim on the region's cursor — a
    // claim when needed, held in a scalar register: this kernel has no VGPR to carry a claim across the vertex code
    const int lane = (int)(threadIdx.x & 63);
    int *cursor = hl.count + region * PT_CNT_STRIDE + (part == 2 ? PT_CNT_CURSOR2 : PT_CNT_CURSOR);
    auto next_chunk = [&](int kb) {
        return hl.dynamic ? lo + __builtin_amdgcn_readfirstlane(lane == 0 ? atomicAdd(cursor, 64) : 0) : kb + nsub * 64;
    };
    for (int kb = hl.dynamic ? next_chunk(0) : lo + sub * 64; kb < hi; kb = next_chunk(kb)) {
        const int k = kb + lane;
        const bool is_back = k >= n_fpad;
        const bool live = is_back ? (k - n_fpad < n_back) : (k < n_front);
        const long i = first + (live ? hl.idx[is_back ? first + qin.seg - 1 - (k - n_fpad) : first + k] : 0);
        int slot = 0, prim = -1;
        if (live) { slot = qin.slot[i]; prim = qin.prim[i]; }
        const long P = st.stride, S = qin.stride;
        d3 T = mk3(0, 0, 0);
        if (live) T = level == 0 ? mk3(1., 1., 1.) : mk3(st.T[slot], st.T[P + slot], st.T[2 * P + slot]);
        const bool hit = live && prim >= 0;
        Ray ray;
        VertexT<R> vx;
        v3<R> wo = mkv<R>(R(0), R(0), R(0));
        SampleSrc ss;
        ss.row = nullptr; ss.msid = 0;
        if (hit) {
            ray.o = mk3(qin.rays[i], qin.rays[S + i], qin.rays[2 * S + i]);
            ray.d = mk3(qin.rays[3 * S + i], qin.rays[4 * S + i], qin.rays[5 * S + i]);
            Vertex vd;                                   // the vertex and its frame are built in binary64
            make_vertex<EXT != 0>(sc, ray, prim, qin.t[i], vd);
            vd.mat = mats_p + (vd.mat - sc.mats);
            vd.brdf = brdf;
            vd.brdf_in_lds = EXT == 2 && sc.brdf_lds_bytes > 0;
            vd.hits = hits;
            if (EXT != 0 && sc.textures)
                apply_textures(sc, vd, prim, qin.b1[i], qin.b2[i], sc.tex_scratch + ((size_t)blockIdx.x * PT_BLOCK + threadIdx.x));
            vx = vertex_as<R>(vd);
            wo = cvt3<R>(-ray.d);
            int px, py, si;
            uint32_t sid;
            tile_decode(rc, base + slot, px, py, si, sid);
            ss = sample_src(rc, sid);
        }
        bool cont = false;
        Ray nray;
        if (KIND == PT_INTEG_WHITTED) {
            // every light once with LightSample.from_rand(rng): 3 draws each, then 3 for specular_reflect
            // (integrator/utility.py:128); along the reflect chain depth d starts at draw d (3 nL + 3)
            uint32_t k = (uint32_t)level * (3u * (uint32_t)sc.n_lights + 3u);
            for (int li = 0; li < sc.n_lights; li++, k += 3) {
                bool need = false;
                Ray sr;
                d3 c = mk3(0, 0, 0);
                if (hit) {
                    need = estimate_whitted<EXT == 2, R, LOBES>(sc, lights_p[li], vx, wo, (R)hash_uniform(ss.msid, k),
                                                         (R)hash_uniform(ss.msid, k + 1), sr, c);
                    c = T * c;
                }
                push_shadow(sh, region, need, sr, c, slot);
            }
        } else if (KIND == PT_INTEG_DIRECT) {
            // uniform_sample_all_lights, integrator/utility.py:45-77
            for (int li = 0; li < sc.n_lights; li++) {
                const int ns = rc.lay.d_ns[li], off = rc.lay.d_lpos[li];
                unsigned long long perm = 0;
                if (hit && !ss.row) perm = lhs_perm(ns, ss.msid, 0x20000u + (uint32_t)off * 64u);
                for (int j = 0; j < ns; j++) {
                    bool need = false;
                    Ray sr;
                    d3 c = mk3(0, 0, 0);
                    if (hit) {
                        double u0, u1;
                        direct_light_uv(ss, off, ns, j, perm, u0, u1);
                        need = estimate_direct<EXT == 2, R, LOBES>(sc, lights_p[li], vx, wo, (R)u0, (R)u1, sr, c);
                        c = T * (c / (double)ns);
                    }
                    push_shadow(sh, region, need, sr, c, slot);
                }
            }
        }
        if (KIND != PT_INTEG_PATH) {
            // specular_reflect, integrator/utility.py:120-155 (shared by Direct and Whitted)
            if (hit && (level + 1 < rc.ig.max_depth)) {
                v3<R> wi, f;
                R pdf = bsdf_sample_f<EXT == 2, R, LOBES>(vx, wo, R(0), R(0), 0., BDF_REFLECTION | BDF_SPECULAR, wi, f);
                R adot = r_abs(dot3(wi, vx.nn));
                if (pdf > R(0) && !util_is_black(f) && adot != R(0)) {
                    T = T * cvt3<double>((f * adot) / pdf);
                    cont = true;
                    nray.o = vx.p; nray.d = cvt3<double>(wi); nray.mint = vx.rEps; nray.maxt = __builtin_inf();
                }
            }
        } else {
            bool need = false;
            Ray sr;
            d3 c = mk3(0, 0, 0);
            // rng() draw counter (path.py:105,119; integrator/utility.py:101,113-114): per bounce >= 3,
            // 7 draws in uniform_sample_one_light (if any light), 3 for the path BSDFSample, 1 for the
            // roulette once bounce_cnt > 3
            const uint32_t nl = sc.n_lights > 0 ? 7u : 0u;
            uint32_t k = 0;
            if (level >= 4) k = (nl + 3u) + (nl + 4u) * (uint32_t)(level - 4);
            if (hit && sc.n_lights > 0) {
                // uniform_sample_one_light, integrator/utility.py:80-117
                double ulnum, u0, u1;
                if (level < 3) {
                    ulnum = path_dim(ss, rc.lay.p_lnum[level]);
                    u0 = path_dim(ss, rc.lay.p_lpos[level]);
                    u1 = path_dim(ss, rc.lay.p_lpos[level] + 1);
                } else {
                    ulnum = hash_uniform(ss.msid, k);
                    u0 = hash_uniform(ss.msid, k + 1);
                    u1 = hash_uniform(ss.msid, k + 2);
                }
                int ln = (int)floor(ulnum * sc.n_lights);
                if (ln > sc.n_lights - 1) ln = sc.n_lights - 1;
                need = estimate_direct<EXT == 2, R, LOBES>(sc, lights_p[ln], vx, wo, (R)u0, (R)u1, sr, c);
                c = T * (c * (double)sc.n_lights);
            }
            push_shadow(sh, region, need, sr, c, slot);
            if (hit) {
                double u0, u1, uc;
                if (level < 3) {
                    u0 = path_dim(ss, rc.lay.p_pdir[level]);
                    u1 = path_dim(ss, rc.lay.p_pdir[level] + 1);
                    uc = path_dim(ss, rc.lay.p_pcom[level]);
                } else {
                    const uint32_t kk = k + nl;
                    u0 = hash_uniform(ss.msid, kk);
                    u1 = hash_uniform(ss.msid, kk + 1);
                    uc = hash_uniform(ss.msid, kk + 2);
                }
                v3<R> wi, f;
                R pdf = bsdf_sample_f<EXT == 2, R, LOBES>(vx, wo, (R)u0, (R)u1, uc, BDF_ALL, wi, f);
                if (!(spec_is_black(f) || pdf == R(0))) {
                    T = T * cvt3<double>((f * r_abs(dot3(wi, vx.nn))) / pdf);
                    cont = true;
                    if (level > 3) {
                        double ty = spec_y(T);
                        double cont_prob = ty < .5 ? ty : .5;
                        const uint32_t kr = k + nl + 3u;
                        if (hash_uniform(ss.msid, kr) > cont_prob) cont = false;
                        T = T / cont_prob;
                    }
                    if (level == rc.ig.max_depth) cont = false;
                    nray.o = vx.p; nray.d = cvt3<double>(wi); nray.mint = vx.rEps; nray.maxt = __builtin_inf();
                }
            }
        }
        long pos = wave_append(cont, qout.count + region * PT_CNT_STRIDE);
        if (pos >= 0) {
            pos += (long)region * qout.seg;
            const long SO = qout.stride;
            qout.rays[pos] = nray.o.x; qout.rays[SO + pos] = nray.o.y; qout.rays[2 * SO + pos] = nray.o.z;
            qout.rays[3 * SO + pos] = nray.d.x; qout.rays[4 * SO + pos] = nray.d.y; qout.rays[5 * SO + pos] = nray.d.z;
            qout.rays[6 * SO + pos] = nray.mint; qout.rays[7 * SO + pos] = nray.maxt;
            qout.slot[pos] = slot;
            st.T[slot] = T.x; st.T[P + slot] = T.y; st.T[2 * P + slot] = T.z;
        }
    }
}

hipError_t pt_launch_raygen(int grid, hipStream_t stream, const RenderConst *rc, long base, int nb, int gshift,
                            const Queue &q, const PathState &st, unsigned long long *counters, int *clr_hits) {
    hipLaunchKernelGGL(k_raygen, dim3(grid), dim3(PT_BLOCK), 0, stream, rc, base, nb, gshift, q, st, counters, clr_hits);
    return hipGetLastError();
}

// K3b, after a closest-hit trace of a render queue: an escaped continuation ray adds T * Σ light.le
// (renderer.py:66-68 / path.py:130-135), and the entries that hit something are listed per region
// (wave ballot + one atomic per wave) so that k_shade walks hits only.  From the second bounce on
// ~70 % of the head-mesh rays escape; shading them in place left most lanes of every wave idle
// (shade 25 -> 18 ms per step).  A pass of its own, ~0.1 ms per million entries: inside k_trace the
// extra live state spilled at the 80-VGPR budget and cost more than it saved.
// A wave takes PT_LIST_UNROLL consecutive 64-entry chunks per step: their `prim` loads are in flight together and ONE
// atomic (with return: the dependent step of the loop) reserves the list positions of all of them — at one chunk per
// step the waves sat parked on that round trip 91 % of their cycles (profiles/r02_head_sq.csv: 0.58 ms per launch).
#define PT_LIST_UNROLL 4
// It also resets the three counter blocks the launches AFTER it fill (clr0..2: this level's continuation queue and
// shadow queue, the next level's hit list): nothing reads or writes them while this kernel runs (pt_api.hip orders the
// streams so), and the three 8 KB memset nodes per level they replace were a tenth of the stream operations of a render.
__global__ __launch_bounds__(PT_BLOCK) void k_list_hits(Queue q, HitList hl, int *__restrict__ clr0, int *__restrict__ clr1,
                                                        int *__restrict__ clr2) {
    if (blockIdx.x == 0 && threadIdx.x < PT_SEG) {
        if (clr0) pt_clear_region_line(clr0, threadIdx.x);
        if (clr1) pt_clear_region_line(clr1, threadIdx.x);
        if (clr2) pt_clear_region_line(clr2, threadIdx.x);
    }
    const int wave = (int)((blockIdx.x * PT_BLOCK + threadIdx.x) >> 6);
    const int nsub = (int)((gridDim.x * (PT_BLOCK / 64)) / PT_SEG);
    const int region = wave % PT_SEG, sub = wave / PT_SEG;
    if (sub >= nsub) return;
    const int lane = (int)(threadIdx.x & 63);
    const int n = q.count[region * PT_CNT_STRIDE];
    const long first = (long)region * q.seg;
    for (int kb = sub * (64 * PT_LIST_UNROLL); kb < n; kb += nsub * (64 * PT_LIST_UNROLL)) {
        int prim[PT_LIST_UNROLL];
#pragma unroll
        for (int u = 0; u < PT_LIST_UNROLL; u++) {
            const int k = kb + u * 64 + lane;
            prim[u] = k < n ? q.prim[first + k] : 0;          // 0: past the end counts neither as a hit nor as an escape
        }
        unsigned long long mask[PT_LIST_UNROLL], back[PT_LIST_UNROLL];
        int total = 0, total_b = 0;
#pragma unroll
        for (int u = 0; u < PT_LIST_UNROLL; u++) {
            const int k = kb + u * 64 + lane;
            const bool live = k < n;
            if (live && prim[u] < 0 && hl.add_escape) hl.esc[q.slot[first + k]] = 1;
            const bool hit = live && prim[u] >= 0;
            const bool front = hit && (!hl.tri_cls || hl.tri_cls[prim[u]] != 0);
            mask[u] = __ballot(front);
            back[u] = __ballot(hit && !front);
            total += __popcll(mask[u]);
            total_b += __popcll(back[u]);
        }
        if (total + total_b == 0) continue;                   // wave-uniform
        int base = 0, base_b = 0;
        if (lane == 0) {
            if (total) base = atomicAdd(hl.count + region * PT_CNT_STRIDE, total);
            if (total_b) base_b = atomicAdd(hl.count + region * PT_CNT_STRIDE + PT_CNT_BACK, total_b);
        }
        base = __shfl(base, 0);
        base_b = __shfl(base_b, 0);
        const long last = first + q.seg - 1;                  // the back list grows downwards from the region's end
#pragma unroll
        for (int u = 0; u < PT_LIST_UNROLL; u++) {
            const int k = kb + u * 64 + lane;
            if (k < n && prim[u] >= 0) {
                if ((mask[u] >> lane) & 1ull) hl.idx[first + base + (int)lane_prefix(mask[u])] = k;
                else hl.idx[last - (base_b + (int)lane_prefix(back[u]))] = k;
            }
            base += __popcll(mask[u]);
            base_b += __popcll(back[u]);
        }
    }
}

hipError_t pt_launch_list_hits(int grid, hipStream_t stream, const Queue &q, const HitList &hl, int *clr0, int *clr1,
                               int *clr2) {
    grid = ((grid + PT_SEG_BLOCKS - 1) / PT_SEG_BLOCKS) * PT_SEG_BLOCKS;   // whole sets of PT_SEG waves
    hipLaunchKernelGGL(k_list_hits, dim3(grid), dim3(PT_BLOCK), 0, stream, q, hl, clr0, clr1, clr2);
    return hipGetLastError();
}

// `diffuse_back`: the scene's hits are listed in two classes by lobe set (HitList::tri_cls: front = materials with Blinn /
// Fresnel / specular lobes, back = the others) and the back list is shaded by the kernel compiled for PT_LOBES_DIFFUSE.
// `all_diffuse`: no material of the scene has any other lobe: one list, the diffuse kernel alone.
hipError_t pt_launch_shade(int kind, bool f32, int grid, hipStream_t stream, const DScene &sc, const RenderConst *rc,
                           long base, int level, const Queue &qin, const HitList &hl, const Queue &qout,
                           const ShadowQueue &sh, const PathState &st, bool diffuse_back, bool all_diffuse) {
    grid = ((grid + PT_SEG_BLOCKS - 1) / PT_SEG_BLOCKS) * PT_SEG_BLOCKS;   // whole sets of PT_SEG waves
    // the measured variants take up to PT_MEAS_LDS_MAX of dynamic LDS (hit lists + table blob): beyond the 64 KB a
    // kernel gets by default, so the limit is raised once per device and kernel
    static bool attr_set[PT_MAX_DEVICES][3][2];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= PT_MAX_DEVICES) return hipErrorInvalidDevice;
    if (sc.brdf != nullptr && (kind == PT_INTEG_DIRECT || kind == PT_INTEG_PATH || kind == PT_INTEG_WHITTED) &&
        !attr_set[dev][kind][f32 ? 1 : 0]) {
        const void *fn;
        if (f32)
            fn = kind == PT_INTEG_DIRECT ? reinterpret_cast<const void *>(&k_shade<PT_INTEG_DIRECT, 2, float, PT_LOBES_ALL>)
               : kind == PT_INTEG_WHITTED ? reinterpret_cast<const void *>(&k_shade<PT_INTEG_WHITTED, 2, float, PT_LOBES_ALL>)
                                          : reinterpret_cast<const void *>(&k_shade<PT_INTEG_PATH, 2, float, PT_LOBES_ALL>);
        else
            fn = kind == PT_INTEG_DIRECT ? reinterpret_cast<const void *>(&k_shade<PT_INTEG_DIRECT, 2, double, PT_LOBES_ALL>)
               : kind == PT_INTEG_WHITTED ? reinterpret_cast<const void *>(&k_shade<PT_INTEG_WHITTED, 2, double, PT_LOBES_ALL>)
                                          : reinterpret_cast<const void *>(&k_shade<PT_INTEG_PATH, 2, double, PT_LOBES_ALL>);
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, PT_MEAS_LDS_MAX);
        if (e != hipSuccess) return e;
        attr_set[dev][kind][f32 ? 1 : 0] = true;
    }
#define PT_SHADE(K, T, R, LB, PART) hipLaunchKernelGGL((k_shade<K, T, R, LB>), dim3(grid), dim3(PT_BLOCK), (T) == 2 ? (size_t)sc.brdf_lds_bytes + PT_MEAS_HITS * PT_BLOCK * sizeof(int) : 0, stream, sc, rc, base, level, qin, hl, qout, sh, st, PART)
    const int ext = sc.brdf != nullptr ? 2 : ((sc.textures != nullptr || sc.tri_uv != nullptr) ? 1 : 0);
    // measured scenes keep their own two classes (measured lobes in front) and one kernel for both
#define PT_SHADE_R(K, R) do { \
        if (ext == 2) PT_SHADE(K, 2, R, PT_LOBES_ALL, 0); \
        else if (all_diffuse) { if (ext == 1) PT_SHADE(K, 1, R, PT_LOBES_DIFFUSE, 0); else PT_SHADE(K, 0, R, PT_LOBES_DIFFUSE, 0); } \
        else if (diffuse_back) { \
            if (ext == 1) { PT_SHADE(K, 1, R, PT_LOBES_ALL, 1); PT_SHADE(K, 1, R, PT_LOBES_DIFFUSE, 2); } \
            else { PT_SHADE(K, 0, R, PT_LOBES_ALL, 1); PT_SHADE(K, 0, R, PT_LOBES_DIFFUSE, 2); } \
        } else { if (ext == 1) PT_SHADE(K, 1, R, PT_LOBES_ALL, 0); else PT_SHADE(K, 0, R, PT_LOBES_ALL, 0); } } while (0)
#define PT_SHADE_EXT(K) do { if (f32) PT_SHADE_R(K, float); else PT_SHADE_R(K, double); } while (0)
    if (kind == PT_INTEG_DIRECT) PT_SHADE_EXT(PT_INTEG_DIRECT);
    else if (kind == PT_INTEG_WHITTED) PT_SHADE_EXT(PT_INTEG_WHITTED);
    else PT_SHADE_EXT(PT_INTEG_PATH);
#undef PT_SHADE_EXT
#undef PT_SHADE_R
#undef PT_SHADE
    return hipGetLastError();
}

// =========================================================================================
// K7 film splat: the radiance guards of SamplerRendererTask.__call__ (renderer/renderer.py:
// 185-193) + ImageFilm.add_sample (film/film.py:137-198).  film_accum[y][x][4] f64.
//
// One workgroup owns a strip of SPLAT_PX consecutive pixels of one image row and all samples
// of those pixels that lie in the current batch.  Footprints (9x9 pixels for the 4-pixel
// Lanczos of the examples) are accumulated with LDS f64 atomics into a component-major LDS
// tile of (SPLAT_PX + 2*hx) x (2*hy + 1) pixels — lanes map to DIFFERENT pixels of the strip so
// that a wave's 64 simultaneous adds go to distinct LDS addresses — and the tile is flushed
// once with global f64 atomics: ~0.7 global atomics per sample instead of 324.
// =========================================================================================
#define SPLAT_PX 32
#define SPLAT_GROUP 8

__device__ __forceinline__ d3 guarded_radiance(d3 L, unsigned long long *counters) {
    if (isnan(L.x) || isnan(L.y) || isnan(L.z)) {
        if (counters) atomicAdd(counters + (size_t)(blockIdx.x % PT_CNT_SHARDS) * CNT_COUNT + CNT_GUARD_NAN, 1ull);
        return mk3(0, 0, 0);
    }
    const double y = spec_y(L);
    if (y < -PT_EPS) {
        if (counters) atomicAdd(counters + (size_t)(blockIdx.x % PT_CNT_SHARDS) * CNT_COUNT + CNT_GUARD_NEG, 1ull);
        return mk3(0, 0, 0);
    }
    if (y == __builtin_inf()) {
        if (counters) atomicAdd(counters + (size_t)(blockIdx.x % PT_CNT_SHARDS) * CNT_COUNT + CNT_GUARD_INF, 1ull);
        return mk3(0, 0, 0);
    }
    return L;
}

__global__ __launch_bounds__(PT_BLOCK) void k_splat(const RenderConst *__restrict__ rcp, long base, int nb,
                                                    int row_first, int hx, int hy, int use_lds, PathState st,
                                                    double *__restrict__ film, double *__restrict__ sample_L,
                                                    unsigned long long *counters) {
    extern __shared__ double s_tile[];   // [4][th][tw_l]
    const RenderConst &rc = *rcp;
    const pt_film_desc &fm = rc.film;
    const int tw = rc.tile.x1 - rc.tile.x0, ns = rc.tile.s1 - rc.tile.s0;
    const int nchunks = (tw + SPLAT_PX - 1) / SPLAT_PX;
    const int row = row_first + (int)(blockIdx.x / nchunks);          // row inside the tile
    const int cx0 = (int)(blockIdx.x % nchunks) * SPLAT_PX;           // first strip pixel inside the tile
    const int py = pt_tile_row_y(rc.tile, row);
    const int tw_l = SPLAT_PX + 2 * hx, th_l = 2 * hy + 1;
    const int lx0 = rc.tile.x0 + cx0 - hx, ly0 = py - hy;             // film coords of LDS tile origin
    const int plane = tw_l * th_l;
    static_assert(PT_BLOCK == 256, "one thread copies one entry of the 16x16 filter table");
    __shared__ double s_table[256];          // the film's filter table (film.py:129-135): one LDS read per tap
    s_table[threadIdx.x] = fm.table[threadIdx.x];
    if (use_lds)
        for (int k = threadIdx.x; k < 4 * plane; k += PT_BLOCK) s_tile[k] = 0.;
    __syncthreads();
    const int q = threadIdx.x % SPLAT_PX, sub = threadIdx.x / SPLAT_PX;
    const int px_t = cx0 + q;                                         // pixel column inside the tile
    const long P = st.stride;
    const double xwInv = 1. / fm.xw, ywInv = 1. / fm.yw;
    const int fx1 = fm.x_start + fm.x_cnt - 1, fy1 = fm.y_start + fm.y_cnt - 1;
    // taps of ONE sample, each added on its own (the general path)
    auto splat_taps = [&](double dX, double dY, int x0, int x1, int y0, int y1, double X, double Y, double Z) {
        const bool in_tile = use_lds && x0 >= lx0 && x1 < lx0 + tw_l && y0 >= ly0 && y1 < ly0 + th_l;
        for (int y = y0; y <= y1; y++) {
            int ify = (int)floor(fabs((y - dY) * 16 * ywInv));
            ify = min(ify, 15);
            for (int x = x0; x <= x1; x++) {
                int ifx = (int)floor(fabs((x - dX) * 16 * xwInv));
                ifx = min(ifx, 15);
                const double wt = s_table[ify * 16 + ifx];
                double *t = in_tile ? s_tile + (y - ly0) * tw_l + (x - lx0)
                                    : film + 4 * ((size_t)(y - fm.y_start) * fm.x_cnt + (x - fm.x_start));
                const int cs = in_tile ? plane : 1;
                unsafeAtomicAdd(t, wt * X);
                unsafeAtomicAdd(t + cs, wt * Y);
                unsafeAtomicAdd(t + 2 * cs, wt * Z);
                unsafeAtomicAdd(t + 3 * cs, wt);
            }
        }
    };
    if (px_t < tw) {
        const long pix_i = ((long)row * tw + px_t) * ns;
        constexpr int STEP = PT_BLOCK / SPLAT_PX;
        // Every sample of this pixel has imageX in [px, px+1), so all their footprints fall in one
        // window of <= 2w+2 pixels per axis.  A thread gathers SPLAT_GROUP of its samples and adds their
        // SUM to each window pixel: 4 LDS atomics per window pixel instead of 4 per sample tap (8x fewer
        // at 64 spp; the LDS atomic rate, not the arithmetic, bounded this kernel).  Per sample and axis
        // the filter-table index of every window column/row is packed 5 bits apiece (31 = not covered).
        const int px = rc.tile.x0 + px_t;
        const int wx0 = (int)ceil(px - .5 - fm.xw), wx1 = (int)floor(px + .5 + fm.xw);
        const int wy0 = (int)ceil(py - .5 - fm.yw), wy1 = (int)floor(py + .5 + fm.yw);
        const int W = wx1 - wx0 + 1, H = wy1 - wy0 + 1;
        const bool grouped = use_lds && film && W <= 12 && H <= 12 && wx0 >= lx0 && wx1 < lx0 + tw_l && wy0 >= ly0 &&
                             wy1 < ly0 + th_l;
        for (int s0 = sub; s0 < ns; s0 += STEP * SPLAT_GROUP) {
            double X[SPLAT_GROUP], Y[SPLAT_GROUP], Z[SPLAT_GROUP];
            unsigned long long cx[SPLAT_GROUP], cy[SPLAT_GROUP];
            bool any = false;
#pragma unroll
            for (int j = 0; j < SPLAT_GROUP; j++) {
                X[j] = Y[j] = Z[j] = 0.;
                cx[j] = cy[j] = ~0ull;
                const int sidx = s0 + j * STEP;
                const long out_i = pix_i + sidx;
                if (sidx >= ns || out_i < base || out_i >= base + nb) continue;
                const int slot = (int)(out_i - base);
                d3 Lraw = mk3(st.L[slot], st.L[P + slot], st.L[2 * P + slot]);
                if (st.esc[slot])        // the path's last ray escaped: L += T * Σ light.le, after all its other terms
                    Lraw = Lraw + mk3(st.T[slot] * st.le[0], st.T[P + slot] * st.le[1], st.T[2 * P + slot] * st.le[2]);
                d3 L = guarded_radiance(Lraw, counters);
                if (sample_L) {
                    sample_L[3 * out_i] = L.x; sample_L[3 * out_i + 1] = L.y; sample_L[3 * out_i + 2] = L.z;
                }
                if (!film) continue;
                const double dX = st.img[slot] - .5, dY = st.img[P + slot] - .5;
                int x0 = (int)ceil(dX - fm.xw), x1 = (int)floor(dX + fm.xw);
                int y0 = (int)ceil(dY - fm.yw), y1 = (int)floor(dY + fm.yw);
                x0 = max(x0, fm.x_start); x1 = min(x1, fx1);
                y0 = max(y0, fm.y_start); y1 = min(y1, fy1);
                if (x1 - x0 < 0 || y1 - y0 < 0) continue;
                // rgb2xyz, spectral/spectrum.py:31-34
                const double Xs = (0.412453 * L.x + 0.357580 * L.y) + 0.180423 * L.z;
                const double Ys = (0.212671 * L.x + 0.715160 * L.y) + 0.072169 * L.z;
                const double Zs = (0.019334 * L.x + 0.119193 * L.y) + 0.950227 * L.z;
                if (!grouped || x0 < wx0 || x1 > wx1 || y0 < wy0 || y1 > wy1) {   // e.g. a host sampler's stray sample
                    splat_taps(dX, dY, x0, x1, y0, y1, Xs, Ys, Zs);
                    continue;
                }
                X[j] = Xs; Y[j] = Ys; Z[j] = Zs;
                unsigned long long bx = 0, by = 0;
                for (int w = 0; w < W; w++) {
                    const int x = wx0 + w;
                    int c = 31;
                    if (x >= x0 && x <= x1) c = min((int)floor(fabs((x - dX) * 16 * xwInv)), 15);
                    bx |= (unsigned long long)c << (5 * w);
                }
                for (int h = 0; h < H; h++) {
                    const int y = wy0 + h;
                    int c = 31;
                    if (y >= y0 && y <= y1) c = min((int)floor(fabs((y - dY) * 16 * ywInv)), 15);
                    by |= (unsigned long long)c << (5 * h);
                }
                cx[j] = bx; cy[j] = by;
                any = true;
            }
            if (!any) continue;
            // the group's sum for every window pixel.  (Unrolling the 10 x 10 window of the 4-pixel filters so that the
            // 5-bit fields come out with constant shifts doubled the kernel's time: 800 tap bodies.)
            for (int h = 0; h < H; h++) {
                double *trow = s_tile + (wy0 + h - ly0) * tw_l + (wx0 - lx0);
                int rowb[SPLAT_GROUP];
#pragma unroll
                for (int j = 0; j < SPLAT_GROUP; j++) rowb[j] = ((int)(cy[j] >> (5 * h)) & 31) * 16;
                for (int w = 0; w < W; w++) {
                    double a0 = 0., a1 = 0., a2 = 0., a3 = 0.;
                    bool hitp = false;
#pragma unroll
                    for (int j = 0; j < SPLAT_GROUP; j++) {
                        const int ifx = (int)(cx[j] >> (5 * w)) & 31;
                        if ((rowb[j] | (ifx << 4)) & 256) continue;      // 31 in either field: not covered
                        const double wt = s_table[rowb[j] + ifx];
                        a0 += wt * X[j]; a1 += wt * Y[j]; a2 += wt * Z[j]; a3 += wt;
                        hitp = true;
                    }
                    if (!hitp) continue;
                    unsafeAtomicAdd(trow + w, a0);
                    unsafeAtomicAdd(trow + w + plane, a1);
                    unsafeAtomicAdd(trow + w + 2 * plane, a2);
                    unsafeAtomicAdd(trow + w + 3 * plane, a3);
                }
            }
        }
    }
    if (!use_lds || !film) return;
    __syncthreads();
    for (int k = threadIdx.x; k < plane; k += PT_BLOCK) {
        const int y = ly0 + k / tw_l, x = lx0 + k % tw_l;
        if (x < fm.x_start || x > fx1 || y < fm.y_start || y > fy1) continue;
        const double w = s_tile[3 * plane + k];
        const double a = s_tile[k], b = s_tile[plane + k], c = s_tile[2 * plane + k];
        if (w == 0. && a == 0. && b == 0. && c == 0.) continue;
        double *g = film + 4 * ((size_t)(y - fm.y_start) * fm.x_cnt + (x - fm.x_start));
        unsafeAtomicAdd(g + 0, a);
        unsafeAtomicAdd(g + 1, b);
        unsafeAtomicAdd(g + 2, c);
        unsafeAtomicAdd(g + 3, w);
    }
}

hipError_t pt_launch_splat(hipStream_t stream, const RenderConst *rc, const pt_tile &tile, const pt_film_desc &fm,
                           long base, int nb, const PathState &st, double *film, double *sample_L,
                           unsigned long long *counters) {
    const int tw = tile.x1 - tile.x0, ns = tile.s1 - tile.s0;
    const long per_row = (long)tw * ns;
    const int row_first = (int)(base / per_row), row_last = (int)((base + nb - 1) / per_row);
    const int nchunks = (tw + SPLAT_PX - 1) / SPLAT_PX;
    // samples of pixel p lie in [p, p+1): dX in [p-.5, p+.5) so the footprint reaches at most
    // ceil(w + .5) pixels to either side (anything beyond falls back to global atomics)
    int hx = (int)ceil(fm.xw + .5), hy = (int)ceil(fm.yw + .5);
    size_t lds = (size_t)4 * (SPLAT_PX + 2 * hx) * (2 * hy + 1) * sizeof(double);
    int use_lds = lds <= 48 * 1024;
    if (!use_lds) { hx = hy = 0; lds = 0; }
    const int grid = (row_last - row_first + 1) * nchunks;
    hipLaunchKernelGGL(k_splat, dim3(grid), dim3(PT_BLOCK), lds, stream, rc, base, nb, row_first, hx, hy, use_lds, st,
                       film, sample_L, counters);
    return hipGetLastError();
}

// K8 film resolve: ImageFilm.write_image pixel loop, film/film.py:243-256 (xyz2rgb spectrum.py:25-28)
__global__ __launch_bounds__(PT_BLOCK) void k_film_resolve(const double *__restrict__ acc, long npix,
                                                           double *__restrict__ rgb) {
    const long i = (long)blockIdx.x * PT_BLOCK + threadIdx.x;
    if (i >= npix) return;
    const double a0 = acc[4 * i], a1 = acc[4 * i + 1], a2 = acc[4 * i + 2], ws = acc[4 * i + 3];
    double r = (3.240479 * a0 - 1.537150 * a1) - 0.498535 * a2;
    double g = (-0.969256 * a0 + 1.875991 * a1) + 0.041556 * a2;
    double b = (0.055648 * a0 - 0.204043 * a1) + 1.057311 * a2;
    if (!(ws == 0.)) {
        r = r / ws; g = g / ws; b = b / ws;
        r = r > 0. ? r : 0.; g = g > 0. ? g : 0.; b = b > 0. ? b : 0.;
    }
    rgb[3 * i] = r; rgb[3 * i + 1] = g; rgb[3 * i + 2] = b;
}

hipError_t pt_launch_film_resolve(hipStream_t stream, const double *acc, long npix, double *rgb) {
    const int grid = (int)((npix + PT_BLOCK - 1) / PT_BLOCK);
    if (grid > 0) hipLaunchKernelGGL(k_film_resolve, dim3(grid), dim3(PT_BLOCK), 0, stream, acc, npix, rgb);
    return hipGetLastError();
}

// K9 8-bit image: the array conversion of imageio.write_image (utility/imageio.py:60-65).  Pass 1: per-channel
// maximum.  binary64 values are compared through an order-preserving 64-bit key so that one integer atomic max
// per wave serves negative values too; a NaN (positive quiet) maps above +inf and so propagates like np.max.
__device__ inline unsigned long long f64_key(double v) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ inline double f64_unkey(unsigned long long k) {
    const unsigned long long b = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
    return __longlong_as_double((long long)b);
}

__global__ __launch_bounds__(PT_BLOCK) void k_chan_max(const double *__restrict__ rgb, long npix,
                                                       unsigned long long *__restrict__ keys) {
    unsigned long long m0 = 0, m1 = 0, m2 = 0;      // key 0 is below every finite value and -inf
    for (long i = (long)blockIdx.x * PT_BLOCK + threadIdx.x; i < npix; i += (long)gridDim.x * PT_BLOCK) {
        const unsigned long long a = f64_key(rgb[3 * i]), b = f64_key(rgb[3 * i + 1]), c = f64_key(rgb[3 * i + 2]);
        m0 = a > m0 ? a : m0; m1 = b > m1 ? b : m1; m2 = c > m2 ? c : m2;
    }
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long a = __shfl_xor(m0, o), b = __shfl_xor(m1, o), c = __shfl_xor(m2, o);
        m0 = a > m0 ? a : m0; m1 = b > m1 ? b : m1; m2 = c > m2 ? c : m2;
    }
    if ((threadIdx.x & 63) == 0) {
        atomicMax(keys, m0); atomicMax(keys + 1, m1); atomicMax(keys + 2, m2);
    }
}

// Pass 2: `if coef != 0: v = (255 * v) / coef`, then ndarray.astype('uint8'): the C conversion numpy compiles to on
// x86-64 (truncate to a 32-bit integer, keep the low byte; out-of-range and NaN give the "integer indefinite"
// 0x80000000, i.e. byte 0).
__global__ __launch_bounds__(PT_BLOCK) void k_to_u8(const double *__restrict__ rgb, long n,
                                                    const unsigned long long *__restrict__ keys,
                                                    unsigned char *__restrict__ out) {
    const long i = (long)blockIdx.x * PT_BLOCK + threadIdx.x;
    if (i >= n) return;
    const double coef = f64_unkey(keys[i % 3]);
    double v = rgb[i];
    if (!(coef == 0.)) v = (255. * v) / coef;
    int q = 0;
    if (v > -2147483649. && v < 2147483648.) q = (int)v;
    out[i] = (unsigned char)(q & 255);
}

hipError_t pt_launch_film_to_u8(hipStream_t stream, const double *rgb, long npix, unsigned long long *keys,
                                unsigned char *out) {
    if (npix <= 0) return hipSuccess;
    hipError_t e = hipMemsetAsync(keys, 0, 3 * sizeof(unsigned long long), stream);
    if (e != hipSuccess) return e;
    long g = (npix + PT_BLOCK - 1) / PT_BLOCK;
    if (g > 2048) g = 2048;
    hipLaunchKernelGGL(k_chan_max, dim3((int)g), dim3(PT_BLOCK), 0, stream, rgb, npix, keys);
    const long n = 3 * npix;
    hipLaunchKernelGGL(k_to_u8, dim3((int)((n + PT_BLOCK - 1) / PT_BLOCK)), dim3(PT_BLOCK), 0, stream, rgb, n, keys, out);
    return hipGetLastError();
}
