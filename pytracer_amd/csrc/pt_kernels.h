// pt_kernels.h — shared declarations of the kernels (pt_kernels.hip) and the C-ABI host
// layer (pt_api.hip).
#pragma once

#include "pt_device.h"

#define PT_BLOCK 256   // 4 waves of 64 lanes
static_assert(PT_BLOCK == PT_MEAS_STRIDE, "hit lists of measured_f are strided by the shade block size");
#define PT_MAX_DEVICES 64   // devices one process can hold contexts on (per-device launch attributes)
#define PT_CNT_SHARDS 256

enum {
    CNT_NODE_VISITS = 0,
    CNT_TRI_TESTS,
    CNT_RAYS_CLOSEST,
    CNT_RAYS_ANY,
    CNT_CAMERA_SAMPLES,
    CNT_GUARD_NAN,
    CNT_GUARD_NEG,
    CNT_GUARD_INF,
    CNT_NODE_VISITS_ANY,
    CNT_TRI_TESTS_ANY,
    CNT_COUNT
};

// rows of a tile (include/pytracer_hip.h pt_tile: y_step / y_phase interleave)
__host__ __device__ inline int pt_tile_rows(const pt_tile &t) {
    const int st = t.y_step > 1 ? t.y_step : 1, ph = t.y_step > 1 ? t.y_phase : 0;
    const int span = t.y1 - t.y0 - ph;
    return span > 0 ? (span + st - 1) / st : 0;
}
__host__ __device__ inline int pt_tile_row_y(const pt_tile &t, int row) {
    return t.y_step > 1 ? t.y0 + t.y_phase + row * t.y_step : t.y0 + row;
}

// Segmented queues.  A render queue of capacity `stride` is PT_SEG regions of `seg` entries
// (stride = PT_SEG * seg), each with its own fill counter; entry k of region r is element
// r*seg + k of every array.  Region r is a private lane through the pipeline: the waves with
// (global wave id) % PT_SEG == r read region r of the input queue and append to region r of the
// output queues, so a region never receives more entries than its input region held (x the
// shadow rays per entry) and cannot overflow, and one counter sees 1/PT_SEG of the appends.
// With ONE counter per queue the ~1 M wave-level appends of a 64 M-sample level serialised on
// one L2 word (~88 atomics/us, MI355X_MICROARCH.md) and cost more than the shading itself.
// Grids of kernels that walk a segmented queue are multiples of PT_SEG_BLOCKS workgroups.
#ifndef PT_SEG
#define PT_SEG 64
#endif
// Region r's fill counter is element r * PT_CNT_STRIDE of a counter block: one 128-byte line each.
// Packed, the 64 counters shared two lines and every append in the chip serialised on them in L2
// (k_list_hits: 4.9 ms for 67 M entries packed).
#define PT_CNT_STRIDE 32
#define PT_CNT_BLOCK (PT_SEG * PT_CNT_STRIDE)
// Element PT_CNT_CURSOR of a region's line is its chunk cursor: a trace launch with PT_TRACE_DYNAMIC set in `n_val`
// (segmented input only) lets the waves of a region draw 64-ray chunks from it instead of striding over the region,
// so waves that drew short rays take more chunks.  It pays when a wave gets few chunks (a shard of a render, a deep
// bounce): with ~20 chunks per wave the slowest wave of a static launch ends ~15 % after the mean.  The cursor is
// cleared with the fill counter (the memset of the block / k_raygen), so a queue can be traced dynamically once per
// fill; results and census do not depend on which wave traced a ray.
#define PT_CNT_CURSOR 1
#define PT_TRACE_DYNAMIC 1
#define PT_TRACE_PACKET 4    // with PT_TRACE_DYNAMIC: a launch over coherent chunks (level 0 of a render) walks per wave (k_trace_packet)
#define PT_TRACE_HALF 8      // with PT_TRACE_REFILL: the pair records are read at half size (GPairH: binary32 outer boxes, exact fallback)
#define PT_TRACE_REFILL 2    // with PT_TRACE_DYNAMIC, k_trace_pair launches: lanes that finish take new rays (k_trace_pair_refill)
#define PT_SEG_BLOCKS (PT_SEG / (PT_BLOCK / 64))

// ray queue: traced rays and (after K3) their hit records, plus the path slot they belong to
struct Queue {
    double *rays;   // [8][stride]: ox,oy,oz,dx,dy,dz,mint,maxt
    long stride;
    long seg;       // region size, a multiple of 64
    int *slot;      // [stride]
    int *prim;      // [stride] closest-hit primitive or -1
    double *t;      // [stride] ray.maxt after traversal
    double *b1, *b2;
    int *count;     // [PT_SEG] device counters, PT_CNT_STRIDE apart
};

// shadow-ray queue (segmented the same way) with the contribution each ray carries if unoccluded
struct ShadowQueue {
    double *rays;   // [8][stride]
    double *c;      // [3][stride]
    long stride;
    long seg;       // = shadow rays per entry x Queue::seg
    int *slot;
    int *count;     // [PT_SEG]
};

// hits of a closest-hit trace over a render queue, listed per region (pt_kernels.hip k_list_hits)
struct HitList {
    int *idx;          // [stride] region-local entry numbers, dense from region * seg; nullptr = off
    int *count;        // [PT_SEG]
    int *esc;          // PathState::esc
    int add_escape;    // level > 0: a miss marks its path escaped (esc[slot] = 1); k_splat adds T * Σ light.le for it
    int dynamic;       // k_shade draws its 64-entry chunks from the region's cursor (PT_CNT_CURSOR) instead of striding
    // Hit classes: hits on triangles of class 1 are listed from the FRONT of the region, the others from its BACK (second
    // counter: element PT_CNT_BACK of the region's line), so that a shading wave holds one class only.  nullptr = one list.
    // Scenes with measured BRDF lobes: class 1 = the material has a measured lobe (its table walk costs 5x a plain vertex
    // and a wave pays its most expensive lane).  Other scenes: class 1 = the material has a Blinn / Fresnel / specular
    // lobe, class 0 = diffuse lobes only, and the two lists are shaded by two kernels (pt_launch_shade).
    const unsigned char *tri_cls;   // [n_tris]
};
#define PT_CNT_BACK 2
#define PT_CNT_CURSOR2 3     // chunk cursor of the back list when the two lists are shaded by two launches (k_shade PART 2)

struct ShadowResolve {
    const double *c;
    long cstride;
    const int *slot;
    double *L;
    long lstride;
};

// per-path state, indexed by slot (= sample ordinal inside the batch)
struct PathState {
    double *T;     // [3][stride] path throughput
    double *L;     // [3][stride] accumulated radiance
    double *img;   // [2][stride] imageX, imageY
    int *esc;      // [stride] 1: the path's continuation ray left the scene (k_list_hits).  A path escapes at most once
                   // and its throughput is not touched afterwards, so the radiance it picks up there, T * Σ light.le
                   // (renderer.py:66-68 / path.py:130-135), is added when the sample is splatted: a plain 4-byte store per
                   // escaped ray instead of three scattered binary64 atomics + three loads (k_list_hits 1.30 -> see DESIGN)
    double le[3];  // Σ light.le
    long stride;
};

// offsets into a flattened Sample row (sampler/sample.py:36-76): [imageX,imageY,lens_u,lens_v,time,
// oneD arrays..., twoD arrays...] with the requests of direct.py:46-62 / path.py:34-42
struct Layout {
    int32_t dims;
    int32_t d_lpos[16], d_ns[16];
    int32_t p_lnum[3], p_lpos[3], p_pcom[3], p_pdir[3];
};

struct RenderConst {
    pt_camera cam;
    pt_film_desc film;
    pt_sampler_desc sp;
    pt_integrator_desc ig;
    pt_tile tile;
    Layout lay;
    const double *table;
};

// seg_counts != nullptr: segmented input (regions of `seg` entries); else dense rays [0, n_val)
hipError_t pt_launch_trace(bool any, int stack, int grid, hipStream_t stream, const DScene &sc, const double *rays,
                           long stride, const int *seg_counts, long seg, int n_val, int *out_prim, double *out_t,
                           double *out_b1, double *out_b2, const ShadowResolve &rs, unsigned long long *counters);
int pt_trace_occupancy(bool any, int stack);
// fills frames[n_tris][9] with the shading frame of every triangle (sc.tri_frames must still be nullptr)
hipError_t pt_launch_tri_frames(hipStream_t stream, const DScene &sc, double *frames);
size_t pt_trace_packet_lds(int stack);
int pt_trace_packet_occupancy(bool any, int stack);
hipError_t pt_launch_trace_packet(bool any, int stack, int grid, hipStream_t stream, const DScene &sc, const double *rays,
                                  long stride, const int *seg_counts, long seg, int *out_prim, double *out_t, double *out_b1,
                                  double *out_b2, const ShadowResolve &rs, unsigned long long *counters);
// the same trace with the first sc.n_top breadth-first nodes staged in LDS (1024-thread workgroups)
// k_trace_meta: child/primitive words of all nodes in LDS (small trees); bytes of the table and of the whole block
__host__ __device__ inline size_t pt_trace_meta_table_bytes(int n_nodes) { return (((size_t)n_nodes * 8) + 15) & ~(size_t)15; }
size_t pt_trace_meta_lds(int stack, int n_nodes);
hipError_t pt_launch_trace_meta(bool any, int stack, int grid, hipStream_t stream, const DScene &sc, const double *rays,
                                long stride, const int *seg_counts, long seg, int n_val, int *out_prim, double *out_t,
                                double *out_b1, double *out_b2, const ShadowResolve &rs, unsigned long long *counters);
// k_trace_meta_s: 16-bit todo entries; two 512-thread workgroups per CU, or (packed: the node word in 32 bits) four of 256
size_t pt_trace_meta_s_lds(int stack, int n_nodes, bool packed);
hipError_t pt_launch_trace_meta_s(bool any, bool packed, int stack, int grid, hipStream_t stream, const DScene &sc, const double *rays,
                                  long stride, const int *seg_counts, long seg, int n_val, int *out_prim, double *out_t,
                                  double *out_b1, double *out_b2, const ShadowResolve &rs, unsigned long long *counters);
hipError_t pt_launch_trace_top(bool any, int stack, int grid, hipStream_t stream, const DScene &sc, const double *rays,
                               long stride, const int *seg_counts, long seg, int n_val, int *out_prim, double *out_t,
                               double *out_b1, double *out_b2, const ShadowResolve &rs, unsigned long long *counters);
size_t pt_trace_top_lds(int stack, int n_top);
// k_trace_pair: the walk fetched by parent (GPair records); `stack` todo levels of 8-byte entries in LDS
size_t pt_trace_pair_lds(int lds_levels);
size_t pt_trace_pair_spill_bytes(int levels, long grid_threads);
int pt_trace_pair_occupancy(bool any, int lds_levels, bool half);   // half: of the half-record refill kernel
// grid must already be a whole number of PT_SEG-wave sets for segmented input (the caller sized `spill` for it)
hipError_t pt_launch_trace_pair(bool any, int lds_levels, void *spill, int grid, hipStream_t stream, const DScene &sc,
                                const double *rays, long stride, const int *seg_counts, long seg, int n_val, int *out_prim,
                                double *out_t, double *out_b1, double *out_b2, const ShadowResolve &rs,
                                unsigned long long *counters, int refill_min, const int *order);
// order[region start + k] = the region-local entry to trace k-th: windows of `window` entries (a power of two <= 16384) sorted
// by origin cell / direction octant (mode: low 4 bits = key arrangement, high bits = grid bits per axis, 0 = 4)
hipError_t pt_launch_sort_rays(hipStream_t stream, const DScene &sc, const double *rays, long stride, const int *seg_counts,
                               long seg, int window, int mode, int *order);
hipError_t pt_launch_raygen(int grid, hipStream_t stream, const RenderConst *rc, long base, int nb, int gshift,
                            const Queue &q, const PathState &st, unsigned long long *counters, int *clr_hits);
// clr0..2 (nullable): counter blocks (PT_CNT_BLOCK ints each) the kernel resets for the launches after it
hipError_t pt_launch_list_hits(int grid, hipStream_t stream, const Queue &q, const HitList &hl, int *clr0, int *clr1,
                               int *clr2);
hipError_t pt_launch_shade(int kind, bool f32, int grid, hipStream_t stream, const DScene &sc, const RenderConst *rc,
                           long base, int level, const Queue &qin, const HitList &hl, const Queue &qout,
                           const ShadowQueue &sh, const PathState &st, bool diffuse_back, bool all_diffuse);
hipError_t pt_launch_splat(hipStream_t stream, const RenderConst *rc, const pt_tile &tile, const pt_film_desc &fm,
                           long base, int nb, const PathState &st, double *film, double *sample_L,
                           unsigned long long *counters);
hipError_t pt_launch_film_resolve(hipStream_t stream, const double *acc, long npix, double *rgb);
hipError_t pt_launch_film_to_u8(hipStream_t stream, const double *rgb, long npix, unsigned long long *keys,
                                unsigned char *out);
