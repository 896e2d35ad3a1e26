// pt_api.hip — the C ABI of include/pytracer_hip.h: context, scene upload, batched
// intersection, the wavefront render loop and film resolve.  Host code only launches the
// kernels of pt_kernels.hip; there is no CPU fallback anywhere in this library.
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <math.h>
#include <algorithm>
#include <cmath>
#include <new>
#include <vector>

#include "pt_kernels.h"

// One launch of a profiled call (pt_launch_log): which kernel, on which level of which wavefront batch, the census
// block it counted into and the event pair around it.
struct LogEnt {
    int stage, kernel, level, batch;
    int slice;                       // census block of a trace launch, -1: none
    long ev;                         // index of the first event of its pair, -1: not timed
    pt_launch_record rec;            // filled by read_counters
};
// Census blocks: with profiling on every trace launch of a call counts into a block of its own, so that each launch
// can be priced (rays, node visits, triangle tests, ms: bench.py roofline.per_launch); launches beyond the last
// block share it.  256 blocks x 20 KB = 5 MB per context.
#define PT_LOG_SLICES 256
#define PT_SLICE_WORDS (PT_CNT_SHARDS * CNT_COUNT)

struct pt_ctx {
    int device;
    char err[512];
    // scene (device)
    bool have_scene;
    DScene sc;
    void *d_nodes, *d_tris, *d_verts, *d_mat, *d_flip, *d_mats, *d_lights, *d_top, *d_tex, *d_tex_scratch, *d_uv, *d_brdf, *d_tri_cls, *d_frames;
    void *d_box;             // the boxes of all nodes at 48-byte stride (DScene::node_box): trees that k_trace_meta(_s) can walk
    void *d_meta;            // packed (offset, n_prim | axis) words of all nodes (DScene::node_meta)
    void *d_pairs_h;         // GPairH records (the same at half size)
    int pair_half;           // PT_PAIR_HALF: k_trace_pair_refill reads the half-size records; unset (-1) = yes
    void *d_pairs;           // GPair records of the interior nodes (k_trace_pair)
    int n_pairs;
    int use_pair;            // PT_TRACE_PAIR: 0 never, 1 always, unset (-1) = by tree size (launch_trace)
    int shade_dynamic;       // PT_SHADE_DYNAMIC: the same for k_shade's walk over the hit list
    int use_packet;          // PT_TRACE_PACKET: bit 0 / bit 1 = the level-0 closest-hit / any-hit launches of a render walk per wave; unset (-1) = pt_render decides
    int packet_occ[2];
    int pair_refill;         // PT_PAIR_REFILL: k_trace_pair launches of a render refill their lanes (needs the cursors); unset (-1) = by tree size
    int refill_min;          // PT_REFILL_MIN: lanes that must be waiting before k_trace_pair_refill claims new rays (0: the kernel's default)
    int trace_dynamic;       // PT_TRACE_DYNAMIC: 0 = static chunk assignment in the trace launches of a render, else (default) chunk cursors
    size_t brdf_bytes;
    int top_stack;           // todo levels of k_trace_top's LDS stacks (0: top-of-tree staging off)
    int meta_short;          // PT_TRACE_META_SHORT: 0 = never k_trace_meta_s (16-bit todo entries, small workgroups), 1 = its 512 x 2 form, 2 (default) = 256 x 4 with packed node words where they fit
    bool meta_packable;      // every node's (offset, n_prim) fits the 32-bit word of k_trace_meta_s<.., true> (offset < 2^20, n_prim < 2^10)
    int meta_stack;          // todo levels of k_trace_meta's LDS stacks (0: the tree's child words do not fit in LDS)
    int use_meta;            // k_trace_meta(_s) when the tree fits: PT_TRACE_META=0 never, 1 closest-hit launches only,
                             // 2 (default) both: with 1024-thread workgroups any-hit lost 5 %, with k_trace_meta_s'
                             // four of 256 threads it gains 4 % (15.9 -> 15.2 ms, the step 64.35 -> 64.0)
    int stack_need;          // deepest leaf = max simultaneous todo entries (bvh.py:332 allows 64)
    bool has_spec_reflect;   // any material with a SpecularReflection lobe (Direct recursion)
    bool diffuse_back;       // hits listed in two classes by lobe set, the back list shaded by the diffuse-only kernel
    int sort_rays;           // PT_SORT_RAYS: bit 0 / bit 1 = the continuation / shadow rays of a render are binned by origin cell and
                             // octant before k_trace_pair(_refill) traces them (k_sort_rays); unset (-1) = by tree size
    int sort_window, sort_mode;   // PT_SORT_WINDOW (entries per sorted window), PT_SORT_MODE (key arrangement | grid bits << 4)
    int *order_q, *order_sh[2];   // [cap] / [sh_cap]: trace order of the continuation queue / the two shadow queues
    int lobe_sets;           // PT_SHADE_LOBESETS: the diffuse class is shaded by the kernel compiled for it: 1 (default) on the levels
                             // large enough to pay for a second launch, 2 on every level, 0 never (the classes only)
                             // order the hit lists and one kernel shades both
    bool all_diffuse;        // no material has a Blinn / Fresnel / specular / measured lobe: the diffuse-only kernel alone
    int light_ns[16];
    int shadow_per_vertex_direct;
    // workspace (device), sized for `cap` paths per batch
    long cap, sh_cap;
    void *ws;
    size_t ws_bytes;
    long ws_paths; int ws_spv;   // the batch size and shadow rays per vertex the workspace was sized for
    Queue q[2];
    ShadowQueue sh[2];       // double-buffered: the any-hit trace of level k runs on `aux` beside level k+1's closest-hit
    PathState st;            // trace and shading, which fills the other buffer
    hipStream_t aux;         // second stream of a render (created with the context)
    hipEvent_t ev_shade[2], ev_any[2];   // shade(k) done -> aux may trace sh[k&1]; that trace done -> sh[k&1] may be refilled
    int overlap;             // PT_OVERLAP=0: everything on the caller's stream
    bool any_pending[2];     // an any-hit trace recorded ev_any[k] on `aux` and nothing has waited for it yet
    int use_top;             // stage the top of the tree in LDS (k_trace_top): PT_TRACE_TOP=0 never, 1 always,
                             // unset = for trees beyond the L2 (>= 2^18 nodes = 16 MB).  Measured:
                             // head mesh 5.5 k nodes -5 %, head_loop 352 k nodes +5 %, 10^6-triangle soup +19 %
    int *hit_idx;            // [cap] hit list of the current closest-hit trace (HitList::idx)
    int *d_counts;           // [6][PT_CNT_BLOCK]: region counter blocks of q0, q1, shadow queue 0, the hit list of even
                             // levels, shadow queue 1, the hit list of odd levels
    int trace_occ[2];        // resident blocks/CU of the closest / any trace kernels (0 = not queried)
    int pair_occ[2][2];      // the same for k_trace_pair / the half-record k_trace_pair_refill, [half][any]
    int pair_levels;         // todo levels the tree needs; the first pair_lds_levels of them live in LDS (PT_PAIR_LDS_LEVELS)
    int pair_lds_levels;
    long pair_max_blocks;    // grid limit of k_trace_pair = what d_spill was sized for (a multiple of PT_SEG_BLOCKS)
    void *d_spill;           // global part of k_trace_pair's todo stacks
    RenderConst *d_rc;
    RenderConst *h_rc;       // what d_rc holds (valid when rc_valid): an identical descriptor set is not copied again
    bool rc_valid;
    unsigned long long *d_counters;   // [PT_LOG_SLICES] census blocks (one per trace launch when profiling, else block 0) + 4 words
    int n_cu;
    // measurement (pt_set_profiling): event pairs around launches; every launch of a profiled call is logged
    int profiling;                   // 0 off; 1 every launch; 2 trace launches only
    std::vector<hipEvent_t> *ev;     // pool, used pairwise
    std::vector<LogEnt> *log;        // launches of the last profiled pt_render / pt_intersect (pt_launch_log)
    size_t ev_used;
    int slices_used;                 // census blocks handed out in this call
};

enum { STAGE_RAYGEN = PT_STAGE_RAYGEN, STAGE_TRACE_CLOSEST = PT_STAGE_TRACE_CLOSEST, STAGE_TRACE_ANY = PT_STAGE_TRACE_ANY,
       STAGE_SHADE = PT_STAGE_SHADE, STAGE_SPLAT = PT_STAGE_SPLAT, STAGE_COUNT };

static int fail(pt_ctx *c, int code, const char *fmt, ...) {
    if (c) {
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(c->err, sizeof(c->err), fmt, ap);
        va_end(ap);
    }
    return code;
}

#define HIPCHK(c, call)                                                                             \
    do {                                                                                            \
        hipError_t e_ = (call);                                                                     \
        if (e_ != hipSuccess)                                                                       \
            return fail(c, PT_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

extern "C" int pt_version(void) { return PT_ABI_VERSION; }

// Every entry point runs with the context's device current and puts the caller's device back on
// every exit path: under torch a changed current device silently redirects the caller's next
// allocations and current_stream() look-ups.
struct DeviceGuard {
    int prev;
    hipError_t err;
    explicit DeviceGuard(int dev) : prev(-1), err(hipSuccess) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) err = hipSetDevice(dev);
        else prev = -1;                      // nothing to restore
    }
    ~DeviceGuard() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
    DeviceGuard(const DeviceGuard &) = delete;
    DeviceGuard &operator=(const DeviceGuard &) = delete;
};

// Scratch of pt_intersect_static (include/pytracer_hip.h): the sharded census counters of one trace launch
// ([PT_CNT_SHARDS][CNT_COUNT] u64 = 20 KB), in a file-scope device array instead of memory from hipMalloc.
// Every GPU the library runs on holds its own copy at its own address until the process ends.  In host code
// the NAME of the array is only a stand-in for hipGetSymbolAddress: the address a kernel may use is looked
// up once per device in pt_ctx_create, with that device current, and kept in g_static_scratch[device].
#define PT_STATIC_SCRATCH_WORDS (PT_CNT_SHARDS * CNT_COUNT)
static __device__ __align__(256) unsigned long long scratch[PT_STATIC_SCRATCH_WORDS];
static unsigned long long *g_static_scratch[PT_MAX_DEVICES];

static char g_create_err[512] = "null context";
extern "C" int pt_ctx_destroy(pt_ctx *c);

extern "C" const char *pt_last_error(const pt_ctx *ctx) { return ctx ? ctx->err : g_create_err; }

#define CREATE_CHK(call)                                                                               \
    do {                                                                                               \
        hipError_t e_ = (call);                                                                        \
        if (e_ != hipSuccess) {                                                                        \
            snprintf(g_create_err, sizeof(g_create_err), "%s failed: %s", #call, hipGetErrorString(e_)); \
            if (c) (void)pt_ctx_destroy(c);      /* frees whatever was allocated so far */             \
            return PT_ERR_HIP;                                                                         \
        }                                                                                              \
    } while (0)

extern "C" int pt_ctx_create(int device, pt_ctx **out) {
    if (!out) return PT_ERR_ARG;
    *out = nullptr;
    pt_ctx *c = nullptr;
    int ndev = 0;
    CREATE_CHK(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) {
        snprintf(g_create_err, sizeof(g_create_err), "device %d out of range (%d visible)", device, ndev);
        return PT_ERR_ARG;
    }
    if (device >= PT_MAX_DEVICES) {
        snprintf(g_create_err, sizeof(g_create_err), "device %d: at most %d devices per process", device, PT_MAX_DEVICES);
        return PT_ERR_CAPACITY;
    }
    c = (pt_ctx *)calloc(1, sizeof(pt_ctx));
    if (!c) return PT_ERR_ARG;
    c->device = device;
    DeviceGuard dg(device);
    CREATE_CHK(dg.err);
    if (!g_static_scratch[device]) {     // once per device, with that device current (see `scratch` above)
        unsigned long long *p = nullptr;
        CREATE_CHK(hipGetSymbolAddress((void **)&p, HIP_SYMBOL(scratch)));
        g_static_scratch[device] = p;
    }
    hipDeviceProp_t prop;
    CREATE_CHK(hipGetDeviceProperties(&prop, device));
    c->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    // region fill counters of q0, q1, shadow queue 0, the hit list, shadow queue 1
    CREATE_CHK(hipMalloc((void **)&c->d_counts, 6 * PT_CNT_BLOCK * sizeof(int)));
    c->use_top = -1;
    if (const char *e = getenv("PT_TRACE_TOP")) c->use_top = atoi(e);
    c->use_meta = 2;
    if (const char *e = getenv("PT_TRACE_META")) c->use_meta = atoi(e);
    c->use_pair = -1;
    if (const char *e = getenv("PT_TRACE_PAIR")) c->use_pair = atoi(e);
    CREATE_CHK(hipMalloc((void **)&c->d_rc, sizeof(RenderConst)));
    c->h_rc = (RenderConst *)calloc(1, sizeof(RenderConst));
    if (!c->h_rc) { (void)pt_ctx_destroy(c); return PT_ERR_CAPACITY; }
    // + 4 words behind the census blocks: the channel maxima of pt_film_to_u8
    CREATE_CHK(hipMalloc((void **)&c->d_counters, ((size_t)PT_LOG_SLICES * PT_SLICE_WORDS + 4) * sizeof(unsigned long long)));
    c->overlap = 1;
    if (const char *e = getenv("PT_OVERLAP")) c->overlap = atoi(e);
    c->trace_dynamic = -1;
    c->meta_short = 2;
    if (const char *e = getenv("PT_TRACE_META_SHORT")) c->meta_short = atoi(e);
    c->pair_refill = -1;
    c->use_packet = -1;
    if (const char *e = getenv("PT_TRACE_PACKET")) c->use_packet = atoi(e);
    if (const char *e = getenv("PT_PAIR_REFILL")) c->pair_refill = atoi(e);
    if (const char *e = getenv("PT_REFILL_MIN")) c->refill_min = atoi(e);
    c->pair_half = -1;
    if (const char *e = getenv("PT_PAIR_HALF")) c->pair_half = atoi(e);
    c->sort_rays = -1;
    if (const char *e = getenv("PT_SORT_RAYS")) c->sort_rays = atoi(e);
    c->sort_window = 4096;
    if (const char *e = getenv("PT_SORT_WINDOW")) c->sort_window = atoi(e);
    c->sort_mode = 0;
    if (const char *e = getenv("PT_SORT_MODE")) c->sort_mode = atoi(e);
    c->lobe_sets = 1;
    if (const char *e = getenv("PT_SHADE_LOBESETS")) c->lobe_sets = atoi(e);
    c->shade_dynamic = -1;
    if (const char *e = getenv("PT_SHADE_DYNAMIC")) c->shade_dynamic = atoi(e);
    if (const char *e = getenv("PT_TRACE_DYNAMIC")) c->trace_dynamic = atoi(e);
    {
        // the second stream carries the any-hit traces, which nothing but the final splat waits for: PT_AUX_PRIORITY=1
        // creates it with the lowest priority so that the critical path on the caller's stream is served first
        int lo = 0, hi = 0, prio = 0;
        if (const char *e = getenv("PT_AUX_PRIORITY")) prio = atoi(e);
        if (prio != 0 && hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess && lo != hi)
            CREATE_CHK(hipStreamCreateWithPriority(&c->aux, hipStreamNonBlocking, prio > 0 ? lo : hi));
        else
            CREATE_CHK(hipStreamCreateWithFlags(&c->aux, hipStreamNonBlocking));
    }
    for (int k = 0; k < 2; k++) {
        CREATE_CHK(hipEventCreateWithFlags(&c->ev_shade[k], hipEventDisableTiming));
        CREATE_CHK(hipEventCreateWithFlags(&c->ev_any[k], hipEventDisableTiming));
    }
    c->ev = new std::vector<hipEvent_t>();
    c->log = new std::vector<LogEnt>();
    *out = c;
    return PT_OK;
}

extern "C" int pt_set_profiling(pt_ctx *c, int enable) {
    if (!c) return PT_ERR_ARG;
    c->profiling = enable < 0 ? 0 : (enable > 2 ? 1 : enable);
    return PT_OK;
}

extern "C" const char *pt_kernel_name(int32_t id) {
    switch (id) {
    case PT_KERNEL_TRACE: return "k_trace";
    case PT_KERNEL_TRACE_META: return "k_trace_meta";
    case PT_KERNEL_TRACE_TOP: return "k_trace_top";
    case PT_KERNEL_TRACE_PAIR: return "k_trace_pair";
    case PT_KERNEL_TRACE_PAIR_REFILL: return "k_trace_pair_refill";
    case PT_KERNEL_TRACE_PACKET: return "k_trace_packet";
    default: return "";
    }
}

// One logged launch of a profiled call: an entry in c->log, a pair of events on `stream` around it when this
// profiling level times its stage, and (trace launches) a census block of its own.  No-op unless profiling.
struct StageTimer {
    pt_ctx *c;
    hipStream_t stream;
    bool timed;
    long idx;                        // entry in c->log, -1: nothing logged
    StageTimer(pt_ctx *c_, hipStream_t s, int stage, int level = -1, int batch = 0) : c(c_), stream(s), timed(false), idx(-1) {
        if (!c->profiling) return;
        const bool trace = stage == STAGE_TRACE_CLOSEST || stage == STAGE_TRACE_ANY;
        LogEnt e;
        memset(&e, 0, sizeof(e));
        e.stage = stage; e.level = level; e.batch = batch; e.slice = -1; e.ev = -1;
        if (trace) {
            e.slice = c->slices_used < PT_LOG_SLICES ? c->slices_used : PT_LOG_SLICES - 1;
            c->slices_used++;
        }
        c->log->push_back(e);
        idx = (long)c->log->size() - 1;
        if (c->profiling == 2 && !trace) return;
        if (c->ev_used + 2 > c->ev->size()) {
            hipEvent_t a, b;
            if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return;
            c->ev->push_back(a);
            c->ev->push_back(b);
        }
        (*c->log)[(size_t)idx].ev = (long)c->ev_used;
        timed = true;
        (void)hipEventRecord((*c->ev)[c->ev_used], stream);
    }
    // the census block this launch counts into (`all` = the call's block 0 when nothing is logged)
    unsigned long long *counters(unsigned long long *all) const {
        const int slice = idx >= 0 ? (*c->log)[(size_t)idx].slice : -1;
        return (all && slice >= 0) ? c->d_counters + (size_t)slice * PT_SLICE_WORDS : all;
    }
    void set_kernel(int id) { if (idx >= 0) (*c->log)[(size_t)idx].kernel = id; }
    ~StageTimer() {
        if (!timed) return;
        (void)hipEventRecord((*c->ev)[c->ev_used + 1], stream);
        c->ev_used += 2;
    }
};

// a profiled call starts an empty log
static void log_begin(pt_ctx *c) {
    c->ev_used = 0;
    c->slices_used = 0;
    c->log->clear();
}

static void collect_timing(pt_ctx *c, pt_stats *stats) {
    // caller has synchronised the stream(s)
    for (LogEnt &e : *c->log) {
        e.rec.stage = e.stage; e.rec.kernel = e.kernel; e.rec.level = e.level; e.rec.batch = e.batch;
        e.rec.ms = -1.;
        float ms = 0.f;
        if (e.ev < 0 || hipEventElapsedTime(&ms, (*c->ev)[(size_t)e.ev], (*c->ev)[(size_t)e.ev + 1]) != hipSuccess) continue;
        e.rec.ms = ms;
        switch (e.stage) {
        case STAGE_RAYGEN: stats->ms_raygen += ms; break;
        case STAGE_TRACE_CLOSEST: stats->ms_trace_closest += ms; stats->launches_trace_closest++; break;
        case STAGE_TRACE_ANY: stats->ms_trace_any += ms; stats->launches_trace_any++; break;
        case STAGE_SHADE: stats->ms_shade += ms; break;
        default: stats->ms_splat += ms; break;
        }
    }
    c->ev_used = 0;
}

extern "C" int pt_launch_log(pt_ctx *c, pt_launch_record *out, int32_t cap, int32_t *n_out) {
    if (!c || !n_out || cap < 0 || (cap > 0 && !out)) return PT_ERR_ARG;
    const size_t n = c->log->size();
    *n_out = (int32_t)n;
    for (size_t k = 0; k < n && k < (size_t)cap; k++) out[k] = (*c->log)[k].rec;
    return PT_OK;
}

static void free_scene(pt_ctx *c) {
    void **ps[] = {&c->d_nodes, &c->d_tris, &c->d_verts, &c->d_mat, &c->d_flip, &c->d_mats, &c->d_lights, &c->d_top,
                   &c->d_tex, &c->d_tex_scratch, &c->d_uv, &c->d_brdf, &c->d_pairs, &c->d_spill, &c->d_tri_cls, &c->d_frames, &c->d_meta, &c->d_pairs_h, &c->d_box};
    for (void **p : ps) {
        if (*p) (void)hipFree(*p);
        *p = nullptr;
    }
    c->have_scene = false;
}

extern "C" int pt_ctx_destroy(pt_ctx *c) {
    if (!c) return PT_OK;
    DeviceGuard dg(c->device);
    free_scene(c);
    if (c->ws) (void)hipFree(c->ws);
    if (c->d_counts) (void)hipFree(c->d_counts);
    if (c->d_rc) (void)hipFree(c->d_rc);
    free(c->h_rc);
    if (c->d_counters) (void)hipFree(c->d_counters);
    if (c->aux) { (void)hipStreamSynchronize(c->aux); (void)hipStreamDestroy(c->aux); }
    for (int k = 0; k < 2; k++) {
        if (c->ev_shade[k]) (void)hipEventDestroy(c->ev_shade[k]);
        if (c->ev_any[k]) (void)hipEventDestroy(c->ev_any[k]);
    }
    if (c->ev) { for (hipEvent_t e : *c->ev) (void)hipEventDestroy(e); delete c->ev; }
    delete c->log;
    free(c);
    return PT_OK;
}

// Validates every index the kernels dereference and measures the deepest leaf.
static int check_tree(pt_ctx *c, const pt_scene_desc *d, int *max_depth) {
    const int n = d->n_nodes, p = d->n_tris;
    *max_depth = 0;
    if (n == 0) return PT_OK;
    std::vector<int> stack_node, stack_depth;
    std::vector<char> seen((size_t)n, 0);
    stack_node.push_back(0);
    stack_depth.push_back(0);
    long visited = 0;
    while (!stack_node.empty()) {
        int i = stack_node.back(), dep = stack_depth.back();
        stack_node.pop_back();
        stack_depth.pop_back();
        if (i < 0 || i >= n) return fail(c, PT_ERR_ARG, "node index %d out of range", i);
        if (seen[i]) return fail(c, PT_ERR_ARG, "node %d reachable twice (not a tree)", i);
        seen[i] = 1;
        visited++;
        const int32_t *m = d->node_meta + 3 * (size_t)i;
        if (m[1] < 0 || m[1] > 0xFFFF) return fail(c, PT_ERR_ARG, "node %d: n_prim %d out of range", i, m[1]);
        if (m[1] > 0) {
            if (m[0] < 0 || (long)m[0] + m[1] > p) return fail(c, PT_ERR_ARG, "leaf %d: primitive range out of bounds", i);
            *max_depth = std::max(*max_depth, dep);
        } else {
            if (m[2] < 0 || m[2] > 2) return fail(c, PT_ERR_ARG, "node %d: axis %d", i, m[2]);
            if (i + 1 >= n || m[0] <= i || m[0] >= n) return fail(c, PT_ERR_ARG, "node %d: child out of range", i);
            stack_node.push_back(i + 1); stack_depth.push_back(dep + 1);
            stack_node.push_back(m[0]); stack_depth.push_back(dep + 1);
        }
    }
    (void)visited;
    return PT_OK;
}

template <typename T>
static hipError_t upload(void **dst, const T *src, size_t count) {
    *dst = nullptr;
    size_t bytes = std::max<size_t>(count, 1) * sizeof(T);
    hipError_t e = hipMalloc(dst, bytes);
    if (e != hipSuccess) return e;
    if (count) e = hipMemcpy(*dst, src, count * sizeof(T), hipMemcpyHostToDevice);
    return e;
}

// rounds added to the per-cell guess of the measured BRDF walk (PT_MEAS_GUESS_BIAS, tuning only: results do not depend on it)
static int pt_meas_guess_bias() {
    const char *e = getenv("PT_MEAS_GUESS_BIAS");
    return e ? atoi(e) : 0;      // measured on the C5 workload: shade 208 / 226 / 295 ms per step for 0 / +1 / +2
}

// Measured BRDF tables (PT_LOBE_MEASURED_IRISO): one grid per distinct (first sample, count) range, all of them in
// one device blob (BrdfGrid, pt_device.h).  `mats` is the copy of the materials that goes to the device: a measured
// lobe's `a` becomes the index of its grid.  Ranges were validated by the caller.
static int upload_brdf_tables(pt_ctx *c, const pt_scene_desc *d, std::vector<pt_material> &mats) {
    std::vector<std::pair<int64_t, int64_t>> tables;
    for (auto &m : mats)
        for (int k = 0; k < m.n_lobes; k++)
            if (m.lobes[k].type == PT_LOBE_MEASURED_IRISO) {
                const std::pair<int64_t, int64_t> key((int64_t)m.lobes[k].a, (int64_t)m.lobes[k].b);
                size_t t = 0;
                while (t < tables.size() && tables[t] != key) t++;
                if (t == tables.size()) tables.push_back(key);
                m.lobes[k].a = (double)t;
            }
    if (!tables.empty()) {
        // blob: [BrdfGrid x T][p, v doubles of all tables][cell starts of all tables][start rounds of all tables],
        // addressed by byte offsets from its first byte so that k_shade can run it from an LDS copy
        size_t n_all = 0, cells_all = 0;
        std::vector<BrdfGrid> grids(tables.size());
        for (size_t t = 0; t < tables.size(); t++) {
            const int64_t n = tables[t].second;
            const char *e_per = getenv("PT_MEAS_CELL_SAMPLES");      // tuning only: samples per grid cell
            int g = (int)ceil(cbrt((double)n / (e_per && atof(e_per) > 0. ? atof(e_per) : 2.)));
            g = std::min(std::max(g, 1), 32);
            grids[t].n = (int32_t)n;
            for (int k = 0; k < 3; k++) grids[t].g[k] = g;
            n_all += (size_t)n;
            cells_all += (size_t)g * g * g + 1;
        }
        const size_t off_pv = grids.size() * sizeof(BrdfGrid), off_cells = off_pv + n_all * 6 * sizeof(double);
        const size_t off_k = off_cells + cells_all * sizeof(int32_t);
        const size_t blob_bytes = (off_k + cells_all * sizeof(int32_t) + 7) & ~(size_t)7;
        if (blob_bytes > ((size_t)1 << 31) - 8) return fail(c, PT_ERR_CAPACITY, "measured BRDF tables exceed 2 GB");
        std::vector<char> blob(blob_bytes);
        HIPCHK(c, hipMalloc(&c->d_brdf, blob.size()));
        double *pv = (double *)(blob.data() + off_pv);
        int32_t *cells = (int32_t *)(blob.data() + off_cells);
        int32_t *kguess = (int32_t *)(blob.data() + off_k);
        size_t pos = 0, cpos = 0;
        for (size_t t = 0; t < tables.size(); t++) {
            BrdfGrid &G = grids[t];
            const int64_t n = G.n;
            const double *P = d->brdf_p + 3 * tables[t].first, *V = d->brdf_v + 3 * tables[t].first;
            double bmax[3];
            for (int k = 0; k < 3; k++) {
                G.bmin[k] = n ? P[k] : 0.;
                bmax[k] = G.bmin[k];
            }
            for (int64_t i = 0; i < n; i++)
                for (int k = 0; k < 3; k++) {
                    G.bmin[k] = std::min(G.bmin[k], P[3 * i + k]);
                    bmax[k] = std::max(bmax[k], P[3 * i + k]);
                }
            for (int k = 0; k < 3; k++) {
                const double ext = bmax[k] - G.bmin[k];
                const bool ok = ext > 0. && std::isfinite(ext) && std::isfinite(G.bmin[k]) && std::isfinite(G.g[k] / ext);
                G.inv[k] = ok ? G.g[k] / ext : 0.;
                G.cw[k] = ok ? ext / G.g[k] : 0.;
                if (!std::isfinite(G.bmin[k])) G.bmin[k] = 0.;
            }
            const size_t nc = (size_t)G.g[0] * G.g[1] * G.g[2];
            auto cell_of = [&](const double *q) {
                int ci[3];
                for (int k = 0; k < 3; k++) {           // the same expression as grid_cell (pt_device.h)
                    const double f = floor((q[k] - G.bmin[k]) * G.inv[k]);
                    ci[k] = f >= (double)(G.g[k] - 1) ? G.g[k] - 1 : (f > 0. ? (int)f : 0);
                }
                return ((size_t)ci[2] * G.g[1] + ci[1]) * G.g[0] + ci[0];
            };
            int32_t *cs = cells + cpos;
            std::fill(cs, cs + nc + 1, 0);
            for (int64_t i = 0; i < n; i++) cs[cell_of(P + 3 * i) + 1]++;
            for (size_t q = 0; q < nc; q++) cs[q + 1] += cs[q];
            std::vector<int32_t> fill(cs, cs + nc);
            double *p_out = pv + pos * 6, *v_out = p_out + (size_t)n * 3;
            for (int64_t i = 0; i < n; i++) {
                const int32_t at = fill[cell_of(P + 3 * i)]++;
                memcpy(p_out + 3 * (size_t)at, P + 3 * i, 3 * sizeof(double));
                memcpy(v_out + 3 * (size_t)at, V + 3 * i, 3 * sizeof(double));
            }
            G.off_p = (int32_t)(off_pv + pos * 6 * sizeof(double));
            G.off_v = G.off_p + (int32_t)((size_t)n * 3 * sizeof(double));
            G.off_cells = (int32_t)(off_cells + cpos * sizeof(int32_t));
            G.off_kguess = (int32_t)(off_k + cpos * sizeof(int32_t));
            // the radii of measured.py:53,66, by the same multiplications; radius[15] = 1.81 is the first > 1.5
            G.radius[0] = .01;
            for (int k = 1; k < 16; k++) G.radius[k] = G.radius[k - 1] * 1.414;
            // kguess[cell]: the round at which a query at the cell's centre would end = the first radius that
            // reaches the third-nearest sample; the samples within W cells are examined (beyond: W cell widths).
            // Only a starting point for measured_f's phase 1, which corrects it.
            {
                const int W = 2;
                double cw_min = INFINITY;
                for (int k = 0; k < 3; k++)
                    if (G.cw[k] > 0.) cw_min = std::min(cw_min, G.cw[k]);
                int32_t *ks = kguess + cpos;
                for (int cz = 0; cz < G.g[2]; cz++)
                    for (int cy = 0; cy < G.g[1]; cy++)
                        for (int cx = 0; cx < G.g[0]; cx++) {
                            const int ci[3] = {cx, cy, cz};
                            double ctr[3];
                            for (int k = 0; k < 3; k++) ctr[k] = G.bmin[k] + (ci[k] + .5) * G.cw[k];
                            double best[3] = {INFINITY, INFINITY, INFINITY};     // three smallest squared distances
                            for (int z = std::max(cz - W, 0); z <= std::min(cz + W, G.g[2] - 1); z++)
                                for (int y = std::max(cy - W, 0); y <= std::min(cy + W, G.g[1] - 1); y++) {
                                    const size_t row = ((size_t)z * G.g[1] + y) * G.g[0];
                                    for (int32_t i = cs[row + std::max(cx - W, 0)]; i < cs[row + std::min(cx + W, G.g[0] - 1) + 1]; i++) {
                                        double d2 = 0.;
                                        for (int k = 0; k < 3; k++) {
                                            const double dd = p_out[3 * (size_t)i + k] - ctr[k];
                                            d2 += dd * dd;
                                        }
                                        if (d2 < best[2]) {
                                            best[2] = d2;
                                            if (best[2] < best[1]) std::swap(best[1], best[2]);
                                            if (best[1] < best[0]) std::swap(best[0], best[1]);
                                        }
                                    }
                                }
                            const double d3 = std::min(sqrt(best[2]), (W + .5) * cw_min);
                            int k0 = 0;
                            while (k0 < 15 && !(G.radius[k0] >= d3)) k0++;
                            ks[((size_t)cz * G.g[1] + cy) * G.g[0] + cx] = std::max(std::min(k0 + pt_meas_guess_bias(), 15), 0);
                        }
            }
            pos += (size_t)n;
            cpos += nc + 1;
        }
        c->brdf_bytes = blob_bytes;
        memcpy(blob.data(), grids.data(), off_pv);
        HIPCHK(c, hipMemcpy(c->d_brdf, blob.data(), blob.size(), hipMemcpyHostToDevice));
    }
    return PT_OK;
}

static int scene_upload_impl(pt_ctx *c, const pt_scene_desc *d) {
    if (!c || !d) return PT_ERR_ARG;
    if (d->n_nodes < 0 || d->n_tris < 0 || d->n_materials < 0 || d->n_lights < 0) return fail(c, PT_ERR_ARG, "negative count");
    if ((d->n_nodes && (!d->node_bounds || !d->node_meta)) ||
        (d->n_tris && (!d->tri_verts || !d->tri_mat || !d->tri_flip)) || (d->n_materials && !d->materials) ||
        (d->n_lights && !d->lights) || d->n_textures < 0 || (d->n_textures && !d->textures))
        return fail(c, PT_ERR_ARG, "null array");
    for (int i = 0; i < d->n_textures; i++)
        if (d->textures[i].kind < 0 || d->textures[i].kind > PT_TEX_WRINKLED || d->textures[i].mapping < 0 ||
            d->textures[i].mapping > PT_MAP_IDENTITY3D ||
            ((d->textures[i].kind == PT_TEX_CHECKER3D || d->textures[i].kind == PT_TEX_WRINKLED) != (d->textures[i].mapping == PT_MAP_IDENTITY3D)) ||
            (d->textures[i].kind == PT_TEX_WRINKLED && !(d->textures[i].v[0][1] >= 0. && d->textures[i].v[0][1] <= 64.)))
            return fail(c, PT_ERR_ARG, "texture %d: kind / mapping / octaves", i);
    if (d->n_lights > 16) return fail(c, PT_ERR_CAPACITY, "at most 16 lights");
    DeviceGuard dg(c->device);
    HIPCHK(c, dg.err);
    int depth = 0;
    int rc = check_tree(c, d, &depth);
    if (rc != PT_OK) return rc;
    if (depth > 64) return fail(c, PT_ERR_DEPTH, "BVH leaf depth %d exceeds the 64-entry todo stack (bvh.py:332)", depth);
    for (int i = 0; i < d->n_tris; i++)
        if (d->tri_mat[i] < 0 || d->tri_mat[i] >= d->n_materials) return fail(c, PT_ERR_ARG, "triangle %d: material index", i);
    bool spec = false;
    for (int i = 0; i < d->n_materials; i++) {
        const pt_material &m = d->materials[i];
        if (m.n_lobes < 0 || m.n_lobes > PT_MAX_LOBES) return fail(c, PT_ERR_ARG, "material %d: n_lobes", i);
        for (int k = 0; k < m.n_lobes; k++) {
            if (m.lobes[k].type < 0 || m.lobes[k].type > PT_LOBE_MEASURED_IRISO) return fail(c, PT_ERR_ARG, "material %d: lobe type", i);
            if (m.lobes[k].type == PT_LOBE_MEASURED_IRISO) {
                const double a = m.lobes[k].a, b = m.lobes[k].b;
                if (!(a >= 0. && b >= 0. && a == floor(a) && b == floor(b) && a + b <= (double)d->n_brdf_samples) ||
                    (b > 0. && (!d->brdf_p || !d->brdf_v)) || b > 1e9)
                    return fail(c, PT_ERR_ARG, "material %d: measured BRDF sample range", i);
            }
            if (m.lobes[k].tex < 0 || (m.lobes[k].tex & 0xFFFF) > d->n_textures || PT_LOBE_PARAM_TEX(m.lobes[k].tex) > d->n_textures ||
                (m.lobes[k].tex & ~(0xFFFF | PT_LOBE_DROP_IF_BLACK | (0x7FF << PT_LOBE_PARAM_TEX_SHIFT))) ||
                (PT_LOBE_PARAM_TEX(m.lobes[k].tex) && m.lobes[k].type != PT_LOBE_OREN_NAYAR && m.lobes[k].type != PT_LOBE_MICROFACET_BLINN))
                return fail(c, PT_ERR_ARG, "material %d: texture index", i);
            spec |= m.lobes[k].type == PT_LOBE_SPEC_REFLECT;
        }
    }
    // Nothing of the context is touched before every check has passed: a rejected scene leaves the previous
    // one (its light_ns, its shadow-queue sizing) exactly as it was.
    int spv = 0, ns_new[16] = {0};
    double le[3] = {0, 0, 0};
    for (int i = 0; i < d->n_lights; i++) {
        const pt_light &l = d->lights[i];
        if (l.kind < 0 || l.kind > PT_LIGHT_POINT) return fail(c, PT_ERR_ARG, "light %d: kind", i);
        if (l.ns < 1 || l.ns > 16) return fail(c, PT_ERR_CAPACITY, "light %d: ns must be in 1..16", i);
        ns_new[i] = l.ns;
        spv += l.ns;
        if (l.kind == PT_LIGHT_DIFFUSE_SPHERE) for (int k = 0; k < 3; k++) le[k] += l.emit[k];
    }
    free_scene(c);          // the commit point: from here on the old scene is gone (have_scene = false)
    memcpy(c->light_ns, ns_new, sizeof(c->light_ns));
    // build the device records
    std::vector<GNode> nodes((size_t)d->n_nodes);
    for (int i = 0; i < d->n_nodes; i++) {
        GNode &g = nodes[i];
        memcpy(g.bmin, d->node_bounds + 6 * (size_t)i, 3 * sizeof(double));
        memcpy(g.bmax, d->node_bounds + 6 * (size_t)i + 3, 3 * sizeof(double));
        const int32_t *m = d->node_meta + 3 * (size_t)i;
        g.offset = m[0];
        g.nprim_axis = m[1] | (m[2] << 16);
        g._pad = 0;
    }
    std::vector<GTri> tris((size_t)d->n_tris);
    for (int i = 0; i < d->n_tris; i++) {
        const double *v = d->tri_verts + 9 * (size_t)i;
        GTri &g = tris[i];
        for (int k = 0; k < 3; k++) {
            g.p1[k] = v[k];
            g.e1[k] = v[3 + k] - v[k];   // e1 = p2 - p1, triangle.py:183
            g.e2[k] = v[6 + k] - v[k];   // e2 = p3 - p1, triangle.py:184
        }
        g.leaf_nprim = 0;
        g.leaf_node = 0;
    }
    bool exact_slab_scene = false;      // a box with min > max (or NaN): every ray takes the literal walk anyway
    for (size_t i = 0; i < (size_t)d->n_nodes && !exact_slab_scene; i++) {
        const double *b = d->node_bounds + 6 * i;
        if (!(b[0] <= b[3]) || !(b[1] <= b[4]) || !(b[2] <= b[5])) exact_slab_scene = true;
    }
    // children-pair records (k_trace_pair): interior nodes numbered in depth-first (= index) order, root = pair 0
    {
        std::vector<int32_t> pair_of((size_t)d->n_nodes, -1);
        int32_t n_pairs = 0;
        for (int i = 0; i < d->n_nodes; i++)
            if ((nodes[i].nprim_axis & 0xFFFF) == 0) pair_of[(size_t)i] = n_pairs++;
            else {
                GTri &t0 = tris[(size_t)nodes[i].offset];       // leaf ranges were validated by check_tree
                t0.leaf_nprim = nodes[i].nprim_axis & 0xFFFF;
                t0.leaf_node = i;
            }
        std::vector<GPair> pairs((size_t)n_pairs);
        for (int i = 0; i < d->n_nodes; i++) {
            if (pair_of[(size_t)i] < 0) continue;
            GPair &p = pairs[(size_t)pair_of[(size_t)i]];
            memset(&p, 0, sizeof(p));
            const int ch[2] = {i + 1, nodes[i].offset};
            for (int k = 0; k < 2; k++) {
                const GNode &g = nodes[(size_t)ch[k]];
                double *b = k ? p.b1 : p.b0;
                memcpy(b, g.bmin, 3 * sizeof(double));
                memcpy(b + 3, g.bmax, 3 * sizeof(double));
                const int32_t ref = (g.nprim_axis & 0xFFFF) ? ~g.offset : pair_of[(size_t)ch[k]];
                (k ? p.ref1 : p.ref0) = ref;
                (k ? p.node1 : p.node0) = ch[k];
            }
            p.axis = nodes[i].nprim_axis >> 16;
            p.self = i;
        }
        c->n_pairs = n_pairs;
        if (n_pairs > 0) {
            HIPCHK(c, upload(&c->d_pairs, pairs.data(), pairs.size()));
            {   // the same records with binary32 outer boxes
                std::vector<GPairH> ph((size_t)n_pairs);
                float w_max = 0.f;
                for (size_t k = 0; k < ph.size(); k++) {
                    const GPair &p = pairs[k];
                    GPairH &h = ph[k];
                    double w = 0.;
                    for (int ch = 0; ch < 2; ch++) {
                        const double *b = ch ? p.b1 : p.b0;
                        float *o = ch ? h.b1 : h.b0;
                        for (int a = 0; a < 6; a++) {
                            float f = (float)b[a];
                            if (a < 3) { if ((double)f > b[a]) f = nextafterf(f, -INFINITY); }     // minima towards -inf
                            else { if ((double)f < b[a]) f = nextafterf(f, INFINITY); }            // maxima towards +inf
                            o[a] = f;
                            const double dist = fabs((double)f - b[a]);
                            w = (dist > w || dist != dist) ? dist : w;
                        }
                    }
                    float wf = (float)w;
                    if ((double)wf < w) wf = nextafterf(wf, INFINITY);
                    if (!(w == w)) wf = INFINITY;                                                 // NaN box: always the exact record
                    h.w = wf;
                    w_max = wf > w_max ? wf : w_max;
                    h.ref0 = p.ref0; h.ref1 = p.ref1; h.axis = p.axis;
                }
                c->sc.pair_w_max = w_max;
                HIPCHK(c, upload(&c->d_pairs_h, ph.data(), ph.size()));
            }
            // todo stacks: <= one entry per level of the tree (+ margin); the first L levels in LDS, all of them again
            // in the spill area (the literal walk of rays with zero / NaN direction components uses only that)
            c->pair_levels = (std::max(depth, 1) + 2) & ~1;
            int L = 12;
            if (const char *e = getenv("PT_PAIR_LDS_LEVELS")) L = atoi(e);
            c->pair_lds_levels = std::max(2, std::min(L, c->pair_levels));
            c->pair_max_blocks = (((long)c->n_cu * 8 + PT_SEG_BLOCKS - 1) / PT_SEG_BLOCKS) * PT_SEG_BLOCKS;
            // two areas: an any-hit and a closest-hit launch may run at the same time (two streams of one render)
            HIPCHK(c, hipMalloc(&c->d_spill, 2 * pt_trace_pair_spill_bytes(c->pair_levels, c->pair_max_blocks * 256)));
        }
    }
    HIPCHK(c, upload(&c->d_nodes, nodes.data(), nodes.size()));
    {
        std::vector<int32_t> meta(2 * ((size_t)d->n_nodes + 1), 0);      // padded to a whole int4
        c->meta_packable = true;
        for (int i = 0; i < d->n_nodes; i++) {
            meta[2 * (size_t)i] = nodes[i].offset; meta[2 * (size_t)i + 1] = nodes[i].nprim_axis;
            if (nodes[i].offset < 0 || nodes[i].offset >= (1 << 20) || (nodes[i].nprim_axis & 0xFFFF) >= (1 << 10) ||
                (nodes[i].nprim_axis >> 16) < 0 || (nodes[i].nprim_axis >> 16) > 3) c->meta_packable = false;
        }
        HIPCHK(c, upload(&c->d_meta, meta.data(), meta.size()));
    }
    // top of the tree for k_trace_top: the first K nodes in breadth-first order, K from the LDS left
    // beside 1024 per-lane stacks (160 KB per CU on gfx950)
    int n_top = 0;
    c->top_stack = 0;
    c->meta_stack = 0;
    {
        const int levels = (std::max(depth, 1) + 3) & ~3;
        if (d->n_nodes > 0 && pt_trace_meta_lds(levels, d->n_nodes) <= (size_t)160 * 1024 - 1024) c->meta_stack = levels;
        if (c->meta_stack > 0) {        // the boxes alone, 48 bytes apart: what k_trace_meta(_s) fetches per visit
            std::vector<double> box(6 * (size_t)d->n_nodes);
            for (int i = 0; i < d->n_nodes; i++) {
                memcpy(&box[6 * (size_t)i], nodes[i].bmin, 3 * sizeof(double));
                memcpy(&box[6 * (size_t)i + 3], nodes[i].bmax, 3 * sizeof(double));
            }
            HIPCHK(c, upload(&c->d_box, box.data(), box.size()));
        }
        long k = ((long)160 * 1024 - 1024 - (long)levels * 4096) / (long)sizeof(TopNode);
        if (const char *e = getenv("PT_TOP_NODES")) k = std::min(k, atol(e));
        k = std::min<long>(k, d->n_nodes);
        if (k >= 64) {
            std::vector<int> order, slot((size_t)d->n_nodes, -1);
            order.reserve((size_t)k);
            order.push_back(0);
            for (size_t h = 0; h < order.size() && (long)order.size() < k; h++) {
                const int i = order[h];
                const int32_t *m = d->node_meta + 3 * (size_t)i;
                if (m[1] == 0) {
                    order.push_back(i + 1);
                    if ((long)order.size() < k) order.push_back(m[0]);
                }
            }
            for (size_t h = 0; h < order.size(); h++) slot[(size_t)order[h]] = (int)h;
            std::vector<TopNode> top(order.size());
            for (size_t h = 0; h < order.size(); h++) {
                const int i = order[h];
                const GNode &g = nodes[(size_t)i];
                TopNode &t = top[h];
                memcpy(t.bmin, g.bmin, sizeof(t.bmin));
                memcpy(t.bmax, g.bmax, sizeof(t.bmax));
                t.offset = g.offset;
                t.nprim_axis = g.nprim_axis;
                t.ref0 = t.ref1 = 0;
                if ((g.nprim_axis & 0xFFFF) == 0) {
                    t.ref0 = slot[(size_t)i + 1] >= 0 ? ~slot[(size_t)i + 1] : i + 1;
                    t.ref1 = slot[(size_t)g.offset] >= 0 ? ~slot[(size_t)g.offset] : g.offset;
                }
            }
            HIPCHK(c, upload(&c->d_top, top.data(), top.size()));
            n_top = (int)top.size();
            c->top_stack = levels;
        }
    }
    HIPCHK(c, upload(&c->d_tris, tris.data(), tris.size()));
    HIPCHK(c, upload(&c->d_verts, d->tri_verts, 9 * (size_t)d->n_tris));
    HIPCHK(c, upload(&c->d_mat, d->tri_mat, (size_t)d->n_tris));
    HIPCHK(c, upload(&c->d_flip, d->tri_flip, (size_t)d->n_tris));
    // measured BRDF tables: one grid per distinct (first sample, count); the device copy of a measured lobe
    // carries the index of its grid in `a`
    std::vector<pt_material> mats(d->materials, d->materials + d->n_materials);
    rc = upload_brdf_tables(c, d, mats);
    if (rc != PT_OK) return rc;
    HIPCHK(c, upload(&c->d_mats, mats.data(), mats.size()));
    c->diffuse_back = c->all_diffuse = false;
    if (c->d_brdf && d->n_tris > 0 && !getenv("PT_NO_HIT_CLASSES")) {
        // per triangle: does its material carry a measured lobe?  k_list_hits lists those hits apart (pt_kernels.h HitList)
        std::vector<unsigned char> cls((size_t)d->n_tris, 0);
        for (int i = 0; i < d->n_tris; i++) {
            const pt_material &m = mats[(size_t)d->tri_mat[i]];
            for (int l = 0; l < m.n_lobes; l++)
                if (m.lobes[l].type == PT_LOBE_MEASURED_IRISO) cls[(size_t)i] = 1;
        }
        HIPCHK(c, upload(&c->d_tri_cls, cls.data(), cls.size()));
    } else if (!c->d_brdf && d->n_tris > 0 && !getenv("PT_NO_HIT_CLASSES")) {
        // Shading by lobe set (material.py:122-141 Matte against :166-189, :329-364 ...): the class of a material is
        // the set of lobe types it can push; a triangle whose material has diffuse lobes only (Lambert, Oren-Nayar)
        // is class 0 and its hits are shaded by the kernel compiled without the other lobes (pt_kernels.hip k_shade).
        std::vector<unsigned char> mat_general((size_t)d->n_materials, 0);
        for (int i = 0; i < d->n_materials; i++)
            for (int l = 0; l < mats[(size_t)i].n_lobes; l++)
                if (!((PT_LOBES_DIFFUSE >> mats[(size_t)i].lobes[l].type) & 1)) mat_general[(size_t)i] = 1;
        std::vector<unsigned char> cls((size_t)d->n_tris, 0);
        long n_general = 0;
        for (int i = 0; i < d->n_tris; i++) n_general += (cls[(size_t)i] = mat_general[(size_t)d->tri_mat[i]]);
        if (n_general == 0) c->all_diffuse = true;
        else if (n_general < d->n_tris) {
            c->diffuse_back = true;
            HIPCHK(c, upload(&c->d_tri_cls, cls.data(), cls.size()));
        }
    }
    HIPCHK(c, upload(&c->d_lights, d->lights, (size_t)d->n_lights));
    if (d->n_textures > 0) {
        HIPCHK(c, upload(&c->d_tex, d->textures, (size_t)d->n_textures));
        // one pt_material per thread of the largest shade grid (pt_render: n_cu * 4 blocks, rounded up to a
        // whole set of PT_SEG waves)
        const size_t threads = ((size_t)c->n_cu * 4 + PT_SEG_BLOCKS) * PT_BLOCK;
        HIPCHK(c, hipMalloc(&c->d_tex_scratch, threads * sizeof(pt_material)));
    }
    if (d->tri_uv && d->n_tris > 0) HIPCHK(c, upload(&c->d_uv, d->tri_uv, 6 * (size_t)d->n_tris));
    DScene &s = c->sc;
    s.nodes = (const GNode *)c->d_nodes;
    s.node_meta = (const int2 *)c->d_meta;
    s.node_box = (const double2 *)c->d_box;
    s.pairs = (const GPair *)c->d_pairs;
    s.pairs_h = (const GPairH *)c->d_pairs_h;
    s.top = n_top ? (const TopNode *)c->d_top : nullptr;
    s.n_top = n_top;
    s.exact_slab = exact_slab_scene ? 1 : 0;
    s.tris = (const GTri *)c->d_tris;
    s.tri_verts = (const double *)c->d_verts;
    s.tri_mat = (const int32_t *)c->d_mat;
    s.tri_flip = (const int32_t *)c->d_flip;
    s.tri_frames = nullptr;
    s.mats = (const pt_material *)c->d_mats;
    s.lights = (const pt_light *)c->d_lights;
    s.textures = (const pt_texture *)c->d_tex;
    s.tex_scratch = (pt_material *)c->d_tex_scratch;
    s.tri_uv = (const double *)c->d_uv;
    s.brdf = (const char *)c->d_brdf;
    s.brdf_lds_bytes = (c->d_brdf && c->brdf_bytes + PT_MEAS_HITS * PT_BLOCK * sizeof(int) <= (size_t)PT_MEAS_LDS_MAX && !getenv("PT_BRDF_NO_LDS")) ? (int32_t)c->brdf_bytes : 0;
    s.n_nodes = d->n_nodes; s.n_tris = d->n_tris; s.n_mats = d->n_materials; s.n_lights = d->n_lights;
    memcpy(s.le_sum, le, sizeof(le));
    if (d->n_tris > 0) {
        // the shading frame of every triangle, by the device code a per-hit evaluation would run (pt_device.h tri_frame)
        HIPCHK(c, hipMalloc(&c->d_frames, 9 * (size_t)d->n_tris * sizeof(double)));
        HIPCHK(c, pt_launch_tri_frames(nullptr, s, (double *)c->d_frames));
        HIPCHK(c, hipStreamSynchronize(nullptr));
        s.tri_frames = (const double *)c->d_frames;
    }
    c->stack_need = std::max(depth, 1);
    c->trace_occ[0] = c->trace_occ[1] = 0;
    memset(c->pair_occ, 0, sizeof(c->pair_occ));
    c->has_spec_reflect = spec;
    c->shadow_per_vertex_direct = spv;
    c->have_scene = true;
    return PT_OK;
}

// No C++ exception crosses the C ABI (std::vector growth on a huge scene): PT_ERR_CAPACITY, scene invalidated.
extern "C" int pt_scene_upload(pt_ctx *c, const pt_scene_desc *d) {
    try {
        return scene_upload_impl(c, d);
    } catch (const std::bad_alloc &) {
        if (c) { DeviceGuard dg(c->device); free_scene(c); }
        return fail(c, PT_ERR_CAPACITY, "out of host memory preparing the scene for upload");
    } catch (...) {
        if (c) { DeviceGuard dg(c->device); free_scene(c); }
        return fail(c, PT_ERR_ARG, "unexpected C++ exception in pt_scene_upload");
    }
}

// Grid of a trace launch: one full residency round (CUs x resident workgroups of that kernel),
// grid-stride inside.  A grid that is not a multiple of the residency leaves a partial second
// round: 2048 blocks on a 1280-block residency cost 14-17 % (scripts/trace_micro.py).
static int trace_grid(pt_ctx *c, long n, bool any) {
    long blocks = (n + PT_BLOCK - 1) / PT_BLOCK;
    int &occ = c->trace_occ[any ? 1 : 0];
    if (occ == 0) occ = pt_trace_occupancy(any, c->stack_need);
    long per_cu = occ;
    if (const char *e = getenv("PT_TRACE_BLOCKS_PER_CU")) per_cu = std::max(1l, atol(e));
    long cap = (long)c->n_cu * per_cu;
    return (int)std::max<long>(1, std::min(blocks, cap));
}

// would a non-packet trace launch of this kind run k_trace_pair(_refill)?  (the kernels that can trace through `order`)
static bool trace_uses_pair(const pt_ctx *c, bool any) {
    int use_pair = c->use_pair;
    if (use_pair < 0) use_pair = c->meta_stack > 0 ? 0 : 3;
    return c->d_pairs && c->n_pairs > 0 && c->d_spill && (use_pair & (any ? 2 : 1));
}

// every trace launch goes through here: top-of-tree kernel when the tree qualified at upload
static hipError_t launch_trace(pt_ctx *c, bool any, long n_bound, hipStream_t stream, const double *rays, long stride,
                               const int *seg_counts, long seg, int n_val, int *out_prim, double *out_t, double *out_b1,
                               double *out_b2, const ShadowResolve &rs, unsigned long long *counters, int *kid = nullptr,
                               const int *order = nullptr) {
    int kid_dummy;
    if (!kid) kid = &kid_dummy;
    // children-pair walk (PT_TRACE_PAIR: bit 0 = closest-hit launches, bit 1 = any-hit launches; unset: by tree size)
    // Measured (profiles/r02_*): head mesh (5.5 k nodes, tree L1/L2-resident, k_trace_meta
    // applies) closest 40.7 -> 45.7 ms, any 20.5 -> 20.5 ms: off; head_loop (352 k nodes) closest 2.1 -> 1.4 ms, any
    // 3.3 -> 2.7 ms; 10^6-triangle soup (2 M nodes) closest 114 -> 88 ms, any 61 -> 42 ms: on for trees whose
    // child words do not fit in LDS.
    if (seg_counts && (n_val & PT_TRACE_PACKET)) {      // coherent chunks (pt_render, level 0): one walk per wave
        int &occ = c->packet_occ[any ? 1 : 0];
        if (occ == 0) occ = pt_trace_packet_occupancy(any, c->stack_need);
        const long blocks = (n_bound + PT_BLOCK - 1) / PT_BLOCK;
        const int grid = (int)std::max<long>(1, std::min<long>(blocks, (long)c->n_cu * occ));
        *kid = PT_KERNEL_TRACE_PACKET;
        return pt_launch_trace_packet(any, c->stack_need, grid, stream, c->sc, rays, stride, seg_counts, seg, out_prim, out_t,
                                      out_b1, out_b2, rs, counters);
    }
    int use_pair = c->use_pair;
    if (use_pair < 0) use_pair = c->meta_stack > 0 ? 0 : 3;
    if (c->d_pairs && c->n_pairs > 0 && c->d_spill && (use_pair & (any ? 2 : 1))) {
        // lane refill (k_trace_pair_refill) pays where walks are long and uneven: by tree size unless PT_PAIR_REFILL says;
        // it reads the half-size records unless PT_PAIR_HALF=0 (soup, 16 spp, one stream: closest 59.5 -> 55.5 ms, any 30.7 -> 27.3)
        const bool refill = seg_counts && (n_val & PT_TRACE_DYNAMIC) && (c->pair_refill < 0 ? c->sc.n_nodes >= (1 << 20) : c->pair_refill != 0);
        const bool half = refill && c->pair_half != 0;
        int &occ = c->pair_occ[half ? 1 : 0][any ? 1 : 0];
        if (occ == 0) {
            occ = pt_trace_pair_occupancy(any, c->pair_lds_levels, half);
            // the occupancy API counts LDS to the byte; a grid one block per CU too large leaves a partial second
            // residency round (measured: 32 KB stacks, API 5, 117 vs 88 ms): keep a margin
            const long by_lds = ((long)160 * 1024 - 4096) / (long)std::max<size_t>(pt_trace_pair_lds(c->pair_lds_levels), 1);
            occ = (int)std::max<long>(1, std::min<long>(occ, by_lds));
        }
        long per_cu = occ;
        if (const char *e = getenv("PT_TRACE_BLOCKS_PER_CU")) per_cu = std::max(1l, atol(e));
        const long blocks = (n_bound + 255) / 256;
        long grid = std::max<long>(1, std::min<long>(blocks, (long)c->n_cu * per_cu));
        if (seg_counts) grid = ((grid + PT_SEG_BLOCKS - 1) / PT_SEG_BLOCKS) * PT_SEG_BLOCKS;   // whole sets of PT_SEG waves
        grid = std::min<long>(grid, c->pair_max_blocks);                                        // what the spill area holds
        char *spill = (char *)c->d_spill + (any ? pt_trace_pair_spill_bytes(c->pair_levels, c->pair_max_blocks * 256) : 0);
        *kid = refill ? PT_KERNEL_TRACE_PAIR_REFILL : PT_KERNEL_TRACE_PAIR;
        return pt_launch_trace_pair(any, c->pair_lds_levels, spill, (int)grid, stream, c->sc, rays, stride, seg_counts,
                                    seg, refill ? (n_val | PT_TRACE_REFILL | (half ? PT_TRACE_HALF : 0)) : n_val, out_prim, out_t, out_b1, out_b2, rs, counters,
                                    c->refill_min ? c->refill_min : (half ? 24 : 0),   // half-size records: 24 waiting lanes (soup closest 55.3 -> 54.6 ms); else the kernel's 32
                                    order);
    }
    // ... for launches that can hold enough rays to pay for its 1024-thread workgroups and their LDS staging: a launch
    // bounded by fewer than 2^20 rays runs the plain kernel (C1, 16 k camera rays: closest-hit 0.170 -> 0.125 ms,
    // the whole render 0.470 -> 0.421 ms; PT_TRACE_META=3 / 4 = closest / both at every size)
    const bool meta_size_ok = n_bound >= (1l << 20) || c->use_meta >= 3;
    if (c->meta_stack > 0 && c->use_top <= 0 && meta_size_ok && (c->use_meta == 2 || c->use_meta == 4 || ((c->use_meta == 1 || c->use_meta == 3) && !any))) {
        *kid = PT_KERNEL_TRACE_META;
        // small workgroups with 16-bit todo entries where they fit (k_trace_meta_s): four of 256 threads with the node
        // words packed into 32 bits, else two of 512; PT_TRACE_META_SHORT=1 stops at the latter, 0 keeps the 1024-thread form
        const size_t lds_cu = (size_t)160 * 1024 - 2048;
        if (c->meta_short > 0 && c->sc.n_nodes <= 65535) {
            const bool packed = c->meta_short > 1 && c->meta_packable && 4 * pt_trace_meta_s_lds(c->meta_stack, c->sc.n_nodes, true) <= lds_cu;
            if (packed || 2 * pt_trace_meta_s_lds(c->meta_stack, c->sc.n_nodes, false) <= lds_cu) {
                const int bt = packed ? 256 : 512;
                const long blocks = (n_bound + bt - 1) / bt;
                const int grid = (int)std::max<long>(1, std::min<long>(blocks, (long)(1024 / bt) * c->n_cu));
                return pt_launch_trace_meta_s(any, packed, c->meta_stack, grid, stream, c->sc, rays, stride, seg_counts, seg, n_val,
                                              out_prim, out_t, out_b1, out_b2, rs, counters);
            }
        }
        const long blocks = (n_bound + 1023) / 1024;
        const int grid = (int)std::max<long>(1, std::min<long>(blocks, c->n_cu));
        return pt_launch_trace_meta(any, c->meta_stack, grid, stream, c->sc, rays, stride, seg_counts, seg, n_val, out_prim,
                                    out_t, out_b1, out_b2, rs, counters);
    }
    if (c->top_stack > 0 && (c->use_top > 0 || (c->use_top < 0 && c->sc.n_nodes >= (1 << 18)))) {
        const long blocks = (n_bound + 1023) / 1024;
        const int grid = (int)std::max<long>(1, std::min<long>(blocks, c->n_cu));
        *kid = PT_KERNEL_TRACE_TOP;
        return pt_launch_trace_top(any, c->top_stack, grid, stream, c->sc, rays, stride, seg_counts, seg, n_val, out_prim,
                                   out_t, out_b1, out_b2, rs, counters);
    }
    *kid = PT_KERNEL_TRACE;
    return pt_launch_trace(any, c->stack_need, trace_grid(c, n_bound, any), stream, c->sc, rays, stride, seg_counts, seg,
                           n_val, out_prim, out_t, out_b1, out_b2, rs, counters);
}

// Sums the census shards at `src` (device) into stats.  Host side on the stack: no allocation of any kind.
static void add_census(const unsigned long long *h, pt_stats *stats);
static int read_counters_from(pt_ctx *c, const unsigned long long *src, hipStream_t stream, pt_stats *stats) {
    static_assert(CNT_COUNT <= 16, "counter block");
    unsigned long long sh[PT_CNT_SHARDS * CNT_COUNT];
    HIPCHK(c, hipMemcpyAsync(sh, src, sizeof(sh), hipMemcpyDeviceToHost, stream));
    HIPCHK(c, hipStreamSynchronize(stream));
    unsigned long long h[CNT_COUNT] = {0};
    for (int k = 0; k < PT_CNT_SHARDS; k++)
        for (int j = 0; j < CNT_COUNT; j++) h[j] += sh[(size_t)k * CNT_COUNT + j];
    add_census(h, stats);
    return PT_OK;
}
static void add_census(const unsigned long long *h, pt_stats *stats) {
    stats->node_visits += h[CNT_NODE_VISITS];
    stats->tri_tests += h[CNT_TRI_TESTS];
    stats->rays_closest += h[CNT_RAYS_CLOSEST];
    stats->rays_any += h[CNT_RAYS_ANY];
    stats->camera_samples += h[CNT_CAMERA_SAMPLES];
    stats->guard_nan += h[CNT_GUARD_NAN];
    stats->guard_neg += h[CNT_GUARD_NEG];
    stats->guard_inf += h[CNT_GUARD_INF];
    stats->node_visits_any += h[CNT_NODE_VISITS_ANY];
    stats->tri_tests_any += h[CNT_TRI_TESTS_ANY];
}

// The census of a call: block 0 alone, or (profiling) every block a trace launch of this call counted into, each
// also written to that launch's log record.  The caller's stream is synchronised.
static int read_counters(pt_ctx *c, hipStream_t stream, pt_stats *stats) {
    if (!c->profiling) return read_counters_from(c, c->d_counters, stream, stats);
    const int ns = std::max(1, std::min(c->slices_used, PT_LOG_SLICES));
    std::vector<unsigned long long> host((size_t)ns * PT_SLICE_WORDS);
    HIPCHK(c, hipMemcpyAsync(host.data(), c->d_counters, host.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, stream));
    HIPCHK(c, hipStreamSynchronize(stream));
    std::vector<unsigned long long> sums((size_t)ns * CNT_COUNT, 0ull);
    for (int s_ = 0; s_ < ns; s_++) {
        unsigned long long *h = sums.data() + (size_t)s_ * CNT_COUNT;
        for (int k = 0; k < PT_CNT_SHARDS; k++)
            for (int j = 0; j < CNT_COUNT; j++) h[j] += host[((size_t)s_ * PT_CNT_SHARDS + k) * CNT_COUNT + j];
        add_census(h, stats);
    }
    collect_timing(c, stats);
    for (LogEnt &e : *c->log) {
        if (e.slice < 0 || e.slice >= ns) continue;
        const bool shared = e.slice == PT_LOG_SLICES - 1 && c->slices_used > PT_LOG_SLICES;   // more launches than blocks
        const unsigned long long *h = sums.data() + (size_t)e.slice * CNT_COUNT;
        e.rec.rays = shared ? 0 : h[CNT_RAYS_CLOSEST] + h[CNT_RAYS_ANY];
        e.rec.node_visits = shared ? 0 : h[CNT_NODE_VISITS];
        e.rec.tri_tests = shared ? 0 : h[CNT_TRI_TESTS];
    }
    return PT_OK;
}

extern "C" int pt_intersect(pt_ctx *c, const double *rays, int64_t n, int any_hit, int32_t *prim, double *t,
                            double *b1, double *b2, pt_stats *stats, void *stream_) {
    if (!c) return PT_ERR_ARG;
    if (!c->have_scene) return fail(c, PT_ERR_NO_SCENE, "pt_intersect before pt_scene_upload");
    if (n < 0 || n > 0x7FFFFFFF) return fail(c, PT_ERR_ARG, "ray count %lld out of range", (long long)n);
    if (n == 0) return PT_OK;
    if (!rays || !prim) return fail(c, PT_ERR_ARG, "null rays/prim");
    hipStream_t stream = (hipStream_t)stream_;
    DeviceGuard dg(c->device);
    HIPCHK(c, dg.err);
    unsigned long long *cnt = nullptr;
    if (stats) {
        HIPCHK(c, hipMemsetAsync(c->d_counters, 0, PT_SLICE_WORDS * sizeof(unsigned long long), stream));
        cnt = c->d_counters;
    }
    ShadowResolve rs = {nullptr, 0, nullptr, nullptr, 0};
    log_begin(c);
    {
        StageTimer tm(c, stream, any_hit ? STAGE_TRACE_ANY : STAGE_TRACE_CLOSEST);
        int kid = 0;
        HIPCHK(c, launch_trace(c, any_hit != 0, (long)n, stream, rays, (long)n, nullptr, 0, (int)n, prim, t, b1, b2, rs,
                               tm.counters(cnt), &kid));
        tm.set_kernel(kid);
    }
    if (stats) return read_counters(c, stream, stats);
    return PT_OK;
}

// pt_intersect for callers that cannot allocate (include/pytracer_hip.h): the census counters live in the
// file-scope device array `scratch` (through its per-device address, never its name), no event is created,
// nothing is allocated or freed.  With stats == NULL the call is two asynchronous operations on `stream`
// (a 20 KB memset of the scratch and the trace launch) and returns without synchronising.
extern "C" int pt_intersect_static(pt_ctx *c, const double *rays, int64_t n, int any_hit, int32_t *prim, double *t,
                                   double *b1, double *b2, pt_stats *stats, void *stream_) {
    if (!c) return PT_ERR_ARG;
    if (!c->have_scene) return fail(c, PT_ERR_NO_SCENE, "pt_intersect_static before pt_scene_upload");
    if (n < 0) return fail(c, PT_ERR_ARG, "ray count %lld is negative", (long long)n);
    if (n > PT_INTERSECT_STATIC_MAX_RAYS)
        return fail(c, PT_ERR_CAPACITY, "pt_intersect_static: %lld rays exceed the maximum of %lld per call",
                    (long long)n, (long long)PT_INTERSECT_STATIC_MAX_RAYS);
    if (n == 0) return PT_OK;
    if (!rays || !prim) return fail(c, PT_ERR_ARG, "null rays/prim");
    hipStream_t stream = (hipStream_t)stream_;
    DeviceGuard dg(c->device);
    HIPCHK(c, dg.err);
    unsigned long long *cnt = g_static_scratch[c->device];      // looked up in pt_ctx_create on this device
    if (!cnt) return fail(c, PT_ERR_HIP, "static scratch of device %d has no address", c->device);
    // the array keeps the previous call's sums: the census starts at zero
    HIPCHK(c, hipMemsetAsync(cnt, 0, PT_STATIC_SCRATCH_WORDS * sizeof(unsigned long long), stream));
    const ShadowResolve rs = {nullptr, 0, nullptr, nullptr, 0};
    HIPCHK(c, launch_trace(c, any_hit != 0, (long)n, stream, rays, (long)n, nullptr, 0, (int)n, prim, t, b1, b2, rs, cnt));
    if (stats) return read_counters_from(c, cnt, stream, stats);
    return PT_OK;
}

static size_t align_up(size_t x) { return (x + 255) & ~(size_t)255; }

// carve the wavefront workspace: two ray queues, the shadow queue (all segmented, pt_kernels.h)
// and the path state.  `paths` = paths in flight, spv = shadow rays one shaded entry can emit.
static int ensure_workspace(pt_ctx *c, long paths, int spv, int gshift) {
    const long gsz = 64l << gshift;                                         // whole groups per region (seg_entry)
    const long seg = gsz * ((((paths + gsz - 1) / gsz) + PT_SEG - 1) / PT_SEG);
    const long cap = seg * PT_SEG, sh_cap = cap * spv;
    if (c->ws && c->cap == cap && c->sh_cap == sh_cap) return PT_OK;
    if (c->ws) { (void)hipFree(c->ws); c->ws = nullptr; }
    c->cap = c->sh_cap = 0; c->ws_bytes = 0; c->ws_paths = 0;
    size_t bytes = 0;
    auto need = [&](size_t b) { size_t off = bytes; bytes += align_up(b); return off; };
    size_t off_q[2][6];
    for (int k = 0; k < 2; k++) {
        off_q[k][0] = need(8 * cap * sizeof(double));
        off_q[k][1] = need(cap * sizeof(int));
        off_q[k][2] = need(cap * sizeof(int));
        off_q[k][3] = need(cap * sizeof(double));
        off_q[k][4] = need(cap * sizeof(double));
        off_q[k][5] = need(cap * sizeof(double));
    }
    size_t off_sr[2], off_sc[2], off_ss[2];
    for (int k = 0; k < 2; k++) {
        off_sr[k] = need(8 * sh_cap * sizeof(double));
        off_sc[k] = need(3 * sh_cap * sizeof(double));
        off_ss[k] = need(sh_cap * sizeof(int));
    }
    size_t off_T = need(3 * cap * sizeof(double)), off_L = need(3 * cap * sizeof(double));
    size_t off_img = need(2 * cap * sizeof(double));
    size_t off_hit = need(cap * sizeof(int));
    size_t off_esc = need(cap * sizeof(int));
    size_t off_oq = need(cap * sizeof(int)), off_osh[2] = {need(sh_cap * sizeof(int)), need(sh_cap * sizeof(int))};
    {
        hipError_t e = hipMalloc(&c->ws, bytes);
        if (e == hipErrorOutOfMemory) { (void)hipGetLastError(); c->ws = nullptr; return PT_ERR_CAPACITY; }   // the caller tries a smaller batch
        HIPCHK(c, e);
    }
    char *b = (char *)c->ws;
    for (int k = 0; k < 2; k++) {
        Queue &q = c->q[k];
        q.rays = (double *)(b + off_q[k][0]); q.stride = cap; q.seg = seg;
        q.slot = (int *)(b + off_q[k][1]); q.prim = (int *)(b + off_q[k][2]);
        q.t = (double *)(b + off_q[k][3]); q.b1 = (double *)(b + off_q[k][4]); q.b2 = (double *)(b + off_q[k][5]);
        q.count = c->d_counts + k * PT_CNT_BLOCK;
    }
    for (int k = 0; k < 2; k++) {
        ShadowQueue &sh = c->sh[k];
        sh.rays = (double *)(b + off_sr[k]); sh.c = (double *)(b + off_sc[k]); sh.slot = (int *)(b + off_ss[k]);
        sh.stride = sh_cap; sh.seg = seg * spv; sh.count = c->d_counts + (k ? 4 : 2) * PT_CNT_BLOCK;
    }
    c->st.T = (double *)(b + off_T); c->st.L = (double *)(b + off_L); c->st.img = (double *)(b + off_img);
    c->st.esc = (int *)(b + off_esc);
    c->st.stride = cap;
    c->hit_idx = (int *)(b + off_hit);
    c->order_q = (int *)(b + off_oq);
    for (int k = 0; k < 2; k++) c->order_sh[k] = (int *)(b + off_osh[k]);
    c->cap = cap; c->sh_cap = sh_cap; c->ws_bytes = bytes; c->ws_paths = paths; c->ws_spv = spv;
    return PT_OK;
}

static int make_layout(const pt_ctx *c, const pt_integrator_desc *ig, Layout *L) {
    memset(L, 0, sizeof(*L));
    const int nl = c->sc.n_lights;
    if (ig->kind == PT_INTEG_DIRECT) {
        int n1 = 0;
        for (int i = 0; i < nl; i++) n1 += 2 * c->light_ns[i];
        int o2 = 5 + n1;
        for (int i = 0; i < nl; i++) {      // direct.py:54-60: LightSampleOffset then BSDFSampleOffset per light
            const int ns = c->light_ns[i];
            L->d_ns[i] = ns;
            L->d_lpos[i] = o2; o2 += 2 * ns;   // light u_pos
            o2 += 2 * ns;                      // bsdf u_dir (only read by the dead branch of estimate_direct)
        }
        L->dims = 5 + n1 + 2 * n1;
    } else if (ig->kind == PT_INTEG_WHITTED) {
        L->dims = 5;                        // WhittedIntegrator requests no samples
    } else {
        int o1 = 5, o2 = 5 + 12;
        for (int i = 0; i < 3; i++) {       // path.py:38-42
            o1++;                            // light u_com (one shape per light: unused)
            L->p_lpos[i] = o2; o2 += 2;
            L->p_lnum[i] = o1++;
            o1++; o2 += 2;                   // bsdf sample of estimate_direct (dead branch)
            L->p_pcom[i] = o1++;
            L->p_pdir[i] = o2; o2 += 2;
        }
        L->dims = 35;
    }
    return PT_OK;
}

extern "C" int pt_render(pt_ctx *c, const pt_camera *cam, const pt_film_desc *film, const pt_sampler_desc *sp,
                         const pt_integrator_desc *ig, const pt_tile *tile, const double *sample_table,
                         int32_t table_dims, double *film_accum, double *sample_L, pt_stats *stats, void *stream_) {
    if (!c) return PT_ERR_ARG;
    if (!c->have_scene) return fail(c, PT_ERR_NO_SCENE, "pt_render before pt_scene_upload");
    if (!cam || !film || !sp || !ig || !tile) return fail(c, PT_ERR_ARG, "null descriptor");
    if (ig->kind != PT_INTEG_DIRECT && ig->kind != PT_INTEG_PATH && ig->kind != PT_INTEG_WHITTED)
        return fail(c, PT_ERR_ARG, "integrator kind %d", ig->kind);
    if (cam->kind != PT_CAM_PERSPECTIVE && cam->kind != PT_CAM_ORTHO) return fail(c, PT_ERR_ARG, "camera kind %d", cam->kind);
    if (ig->max_depth < 0 || ig->max_depth > 64) return fail(c, PT_ERR_ARG, "max_depth %d", ig->max_depth);
    if (ig->shade_f32 != 0 && ig->shade_f32 != 1) return fail(c, PT_ERR_ARG, "shade_f32 %d", ig->shade_f32);
    if (ig->shade_f32 && sample_table)
        return fail(c, PT_ERR_ARG, "shade_f32 is production-only: a sample table (parity mode) runs the binary64 kernels");
    if (sp->xs < 1 || sp->ys < 1 || sp->x_end < sp->x_start || sp->y_end < sp->y_start) return fail(c, PT_ERR_ARG, "sampler window");
    const int spp = sp->xs * sp->ys;
    if (tile->x0 < sp->x_start || tile->x1 > sp->x_end || tile->y0 < sp->y_start || tile->y1 > sp->y_end ||
        tile->x1 < tile->x0 || tile->y1 < tile->y0 || tile->s0 < 0 || tile->s1 > spp || tile->s1 < tile->s0)
        return fail(c, PT_ERR_ARG, "tile outside the sampler window");
    if (tile->y_step < 0 || tile->y_phase < 0 || (tile->y_step > 1 ? tile->y_phase >= tile->y_step : tile->y_phase != 0))
        return fail(c, PT_ERR_ARG, "tile row interleave: need 0 <= y_phase < y_step (y_step %d, y_phase %d)", tile->y_step,
                    tile->y_phase);
    if (film->x_cnt < 1 || film->y_cnt < 1 || !(film->xw > 0.) || !(film->yw > 0.)) return fail(c, PT_ERR_ARG, "film");
    const long window = (long)(sp->x_end - sp->x_start) * (sp->y_end - sp->y_start) * spp;
    if (window > 0xFFFFFFFFl) return fail(c, PT_ERR_CAPACITY, "more than 2^32 samples in the sampler window");
    RenderConst h;
    memset(&h, 0, sizeof(h));
    h.cam = *cam; h.film = *film; h.sp = *sp; h.ig = *ig; h.tile = *tile;
    make_layout(c, ig, &h.lay);
    h.table = sample_table;
    if (sample_table && table_dims != h.lay.dims)
        return fail(c, PT_ERR_ARG, "sample table has %d dims, this scene/integrator needs %d", table_dims, h.lay.dims);
    const long total = (long)(tile->x1 - tile->x0) * pt_tile_rows(*tile) * (tile->s1 - tile->s0);
    if (total == 0) return PT_OK;
    hipStream_t stream = (hipStream_t)stream_;
    DeviceGuard dg(c->device);
    HIPCHK(c, dg.err);

    // paths in flight per wavefront pass.  288 GB of HBM makes a whole 1024^2 x 64 spp render
    // (67 M paths, ~430 B of state each ≈ 29 GB) one batch, which amortises the latency-bound
    // tails of the sparsely populated deep-bounce launches over 16x more work.  Up to 2^28 paths (115 GB, within
    // the half of the free memory taken below): the 10^6-triangle soup at 256 spp, 530 M paths, whose deep launches
    // run ~1 ms each however few rays they hold, 1 377 ms in 8 batches of 2^26, 1 203 in 4, 1 179 in 2.
    const int spv = ig->kind == PT_INTEG_DIRECT ? std::max(c->shadow_per_vertex_direct, 1)
                    : (ig->kind == PT_INTEG_WHITTED ? std::max(c->sc.n_lights, 1) : 1);
    long batch = 1l << 28;
    if (const char *e = getenv("PT_BATCH")) { long v = atol(e); if (v >= 64) batch = v; }
    {
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
            const size_t per_path = 8 * 8 * 2 + 4 * 2 * 2 + 8 * 3 * 2 + 2 * (8 * 11 + 4) * (size_t)std::max(1, ig->kind == PT_INTEG_DIRECT ? c->shadow_per_vertex_direct : (ig->kind == PT_INTEG_WHITTED ? c->sc.n_lights : 1)) + 8 * 8 + 4;
            const size_t budget = (free_b + c->ws_bytes) / 2;     // leave half of what is free to the caller
            const long fit = (long)(budget / per_path);
            if (fit < batch) {
                // bounded by memory.  What is free moves by the allocation granule with every call, so the same render
                // would get a slightly different batch each time and free and allocate ~100 GB per call (and fail now
                // and then: the soup bench's second context did): a workspace the context already holds for this kind
                // of render is kept as long as it is at least half of what would fit now
                if (c->ws && c->ws_spv == spv && c->ws_paths <= batch && c->ws_paths >= fit / 2) batch = c->ws_paths;
                else batch = std::max(fit, 1l << 16);
            }
        }
    }
    batch = std::min(batch, total);
    batch = (batch + 63) & ~63l;
    // chunk-group size of the segmented queues: up to 64 chunks (the 4096-entry sort window), fewer
    // for small batches so that every region still gets several groups
    int gshift = 0;
    int rc;
    for (;;) {
        gshift = 0;
        while (gshift < 6 && ((batch / 64) >> (gshift + 1)) >= 4 * PT_SEG) gshift++;
        rc = ensure_workspace(c, batch, spv, gshift);
        if (rc != PT_ERR_CAPACITY || batch <= (1l << 20)) break;
        batch = ((batch / 2) + 63) & ~63l;          // hipMemGetInfo promised more than hipMalloc gives: half the batch
    }
    if (rc == PT_ERR_CAPACITY) return fail(c, PT_ERR_HIP, "no memory for a wavefront batch of %ld paths", batch);
    if (rc != PT_OK) return rc;
    memcpy(c->st.le, c->sc.le_sum, sizeof(c->st.le));

    // a previous pt_render that failed half-way may have left any-hit traces running on the second stream: nothing of
    // this call may touch the queues before they are done
    for (int k = 0; k < 2; k++)
        if (c->any_pending[k]) {
            HIPCHK(c, hipStreamWaitEvent(stream, c->ev_any[k], 0));
            c->any_pending[k] = false;
        }
    // the descriptor block stays resident: a call with the descriptors of the previous one (every step of a
    // progressive or benchmark loop) copies nothing
    if (!c->rc_valid || memcmp(c->h_rc, &h, sizeof(h)) != 0) {
        c->rc_valid = false;
        HIPCHK(c, hipMemcpyAsync(c->d_rc, &h, sizeof(h), hipMemcpyHostToDevice, stream));
        memcpy(c->h_rc, &h, sizeof(h));
        c->rc_valid = true;
    }
    unsigned long long *cnt = nullptr;
    if (stats) {
        // profiling: one census block per trace launch (read_counters), all cleared here
        const size_t blocks = c->profiling ? PT_LOG_SLICES : 1;
        HIPCHK(c, hipMemsetAsync(c->d_counters, 0, blocks * PT_SLICE_WORDS * sizeof(unsigned long long), stream));
        cnt = c->d_counters;
    }
    // number of shade levels: Path shades bounce 0..max_depth; Direct recurses only through
    // specular reflection (depth+1 < max_depth)
    int levels;
    if (ig->kind == PT_INTEG_PATH) levels = ig->max_depth + 1;
    else levels = c->has_spec_reflect ? std::max(ig->max_depth, 1) : 1;

    const ShadowResolve no_rs = {nullptr, 0, nullptr, nullptr, 0};
    // chunk cursors instead of the static stride (pt_kernels.h PT_TRACE_DYNAMIC) for the launches over a freshly
    // filled queue; the pretest launch shares q[0] with the closest-hit launch after it and stays static
    const int dyn = c->trace_dynamic != 0 ? PT_TRACE_DYNAMIC : 0;
    const int shade_dyn = c->shade_dynamic != 0 ? 1 : 0;
    // one walk per wave (k_trace_packet) for the launches whose chunks are coherent: camera rays always (the 64 samples
    // of a pixel); their shadow rays when there is one per vertex and the tree is small (head mesh 9.2 -> 6.9 ms; with
    // two light samples per vertex, head_loop Direct, 2.1 -> 2.3 ms; from the random triangles of the 10^6-triangle
    // soup 30.6 -> 81.9 ms)
    int use_packet = c->use_packet;
    // ... and only for launches that give every resident wave several chunks: a chunk's per-wave walk is one serial chain
    // (~25 us against ~10 us for 64 per-lane walks side by side), which a 16 k-ray launch cannot hide (C1: 0.49 vs 0.54 ms)
    if (use_packet < 0)
        use_packet = std::min(batch, total) >= (1l << 20) ? (1 | ((ig->kind == PT_INTEG_PATH && c->meta_stack > 0) ? 2 : 0)) : 0;
    // Counter blocks of the hit lists alternate with the level: k_list_hits(k) fills block k & 1 and resets the other
    // one for level k + 1 (nothing has touched it since k_shade(k - 1) ended); k_raygen resets block 0.
    auto hit_count = [&](int level) { return c->d_counts + ((level & 1) ? 5 : 3) * PT_CNT_BLOCK; };
    // ray binning (k_sort_rays) for the launches k_trace_pair(_refill) traces.  Default: off (measured).
    int sort_rays = c->sort_rays < 0 ? 0 : c->sort_rays;
    if (c->sc.n_nodes == 0 || !dyn) sort_rays = 0;
    auto sorted = [&](bool any, hipStream_t s_, const double *rays, long stride, const int *counts, long seg, int *order) -> const int * {
        if (!(sort_rays & (any ? 2 : 1)) || !trace_uses_pair(c, any)) return nullptr;
        if (pt_launch_sort_rays(s_, c->sc, rays, stride, counts, seg, c->sort_window, c->sort_mode, order) != hipSuccess) return nullptr;
        return order;
    };
    auto hit_list = [&](int level) {
        HitList hl{};
        hl.idx = c->hit_idx; hl.count = hit_count(level);
        hl.esc = c->st.esc;
        hl.add_escape = level > 0;
        hl.dynamic = shade_dyn;
        hl.tri_cls = (const unsigned char *)c->d_tri_cls;
        return hl;
    };
    log_begin(c);
    int batch_no = -1;
    for (long base = 0; base < total; base += batch) {
        batch_no++;
        const int nb = (int)std::min(batch, total - base);
        const int g = (nb + PT_BLOCK - 1) / PT_BLOCK;
        const int gs = (int)std::max<long>(1, std::min<long>((nb + PT_BLOCK - 1) / PT_BLOCK, (long)c->n_cu * 4));
        {
            StageTimer tm(c, stream, STAGE_RAYGEN, -1, batch_no);
            HIPCHK(c, pt_launch_raygen(g, stream, c->d_rc, base, nb, gshift, c->q[0], c->st, cnt, hit_count(0)));
        }
        int kid = 0;
        if (ig->pretest) {   // renderer.py:168: same hit flag as the closest-hit pass below
            StageTimer tm(c, stream, STAGE_TRACE_ANY, -1, batch_no);
            HIPCHK(c, launch_trace(c, true, nb, stream, c->q[0].rays, c->q[0].stride, c->q[0].count, c->q[0].seg, 0,
                                   nullptr, nullptr, nullptr, nullptr, no_rs, tm.counters(cnt), &kid));
            tm.set_kernel(kid);
        }
        {
            StageTimer tm(c, stream, STAGE_TRACE_CLOSEST, 0, batch_no);
            HIPCHK(c, launch_trace(c, false, nb, stream, c->q[0].rays, c->q[0].stride, c->q[0].count, c->q[0].seg,
                                   dyn | ((dyn && (use_packet & 1)) ? PT_TRACE_PACKET : 0), c->q[0].prim, c->q[0].t, c->q[0].b1, c->q[0].b2, no_rs, tm.counters(cnt), &kid));
            tm.set_kernel(kid);
        }
        int cur = 0;
        // The any-hit (shadow) trace of level k depends only on shade(k); level k+1's closest-hit trace, hit list and
        // shading do not depend on it (its only output is atomic adds into L).  With PT_OVERLAP (default) it runs on
        // the context's second stream beside them: the sparse deep-bounce launches no longer leave the machine idle
        // in their tails.  Shadow queues alternate, so shade(k+2) waits for trace_any(k).
        // Not with measured BRDF lobes (unless PT_OVERLAP=2): that shade kernel needs its two 78 KB-LDS workgroups per
        // CU, and an any-hit kernel resident beside it costs more than the filled tails give (C5: 371 vs 395 ms).
        const bool ovl = c->overlap != 0 && (c->overlap >= 2 || c->sc.brdf == nullptr);
        bool *any_pending = c->any_pending;
        for (int level = 0; level < levels; level++) {
            Queue &qin = c->q[cur], &qout = c->q[cur ^ 1];
            const int sb = ovl ? (level & 1) : 0;
            ShadowQueue &sh = c->sh[sb];
            if (ovl && any_pending[sb]) {
                HIPCHK(c, hipStreamWaitEvent(stream, c->ev_any[sb], 0));
                any_pending[sb] = false;
            }
            {
                StageTimer tm(c, stream, STAGE_SHADE, level, batch_no);
                // k_list_hits also resets the counter blocks that k_shade and the next level fill: the continuation
                // queue, this level's shadow queue (its last reader, the any-hit trace of level - 2 or level - 1, is
                // done: the wait above / stream order) and the next level's hit list
                HIPCHK(c, pt_launch_list_hits(gs * 2, stream, qin, hit_list(level), qout.count, sh.count, hit_count(level + 1)));
                // two kernels (all lobes / diffuse only) for the two hit lists only where the level can hold enough hits to pay for
                // the second launch: a sparse level costs each launch its latency floor (shard 0 of 8, 8.4 M paths: 9.39 -> 9.20 ms
                // with one kernel from level 1 on; the full 67 M-path render keeps two on levels 0-2: 64.0 against 64.1 ms)
                const bool two_kernels = c->lobe_sets > 1 || (c->lobe_sets == 1 && ((long)nb >> (2 * level)) >= (1l << 24));
                HIPCHK(c, pt_launch_shade(ig->kind, ig->shade_f32 != 0, gs, stream, c->sc, c->d_rc, base, level, qin,
                                          hit_list(level), qout, sh, c->st, c->diffuse_back && two_kernels, c->all_diffuse && c->lobe_sets));
            }
            ShadowResolve rs = {sh.c, sh.stride, sh.slot, c->st.L, c->st.stride};
            hipStream_t any_stream = stream;
            if (ovl) {
                HIPCHK(c, hipEventRecord(c->ev_shade[sb], stream));
                HIPCHK(c, hipStreamWaitEvent(c->aux, c->ev_shade[sb], 0));
                any_stream = c->aux;
            }
            {
                StageTimer tm(c, any_stream, STAGE_TRACE_ANY, level, batch_no);
                const bool pk = dyn && level == 0 && (use_packet & 2);
                const int *ord = pk ? nullptr : sorted(true, any_stream, sh.rays, sh.stride, sh.count, sh.seg, c->order_sh[sb]);
                HIPCHK(c, launch_trace(c, true, (long)nb * spv, any_stream, sh.rays, sh.stride, sh.count, sh.seg,
                                       dyn | (pk ? PT_TRACE_PACKET : 0), nullptr, nullptr,
                                       nullptr, nullptr, rs, tm.counters(cnt), &kid, ord));
                tm.set_kernel(kid);
            }
            if (ovl) {
                HIPCHK(c, hipEventRecord(c->ev_any[sb], c->aux));
                any_pending[sb] = true;
            }
            if (getenv("PT_DEBUG_LEVELS") && base == 0) {   // debugging aid: queue sizes per level (syncs)
                std::vector<int> h(6 * PT_CNT_BLOCK);
                long tot[6] = {0, 0, 0, 0, 0, 0};
                (void)hipStreamSynchronize(c->aux);
                (void)hipMemcpyAsync(h.data(), c->d_counts, h.size() * sizeof(int), hipMemcpyDeviceToHost, stream);
                (void)hipStreamSynchronize(stream);
                for (int k = 0; k < 6 * PT_CNT_BLOCK; k += PT_CNT_STRIDE) tot[k / PT_CNT_BLOCK] += h[k];
                fprintf(stderr, "[pt] level %d: in %ld (hits %ld), continuation %ld, shadow %ld\n", level, tot[cur], tot[(level & 1) ? 5 : 3],
                        tot[cur ^ 1], tot[sb ? 4 : 2]);
            }
            if (level + 1 < levels) {
                StageTimer tm(c, stream, STAGE_TRACE_CLOSEST, level + 1, batch_no);
                const int *ord = sorted(false, stream, qout.rays, qout.stride, qout.count, qout.seg, c->order_q);
                HIPCHK(c, launch_trace(c, false, nb, stream, qout.rays, qout.stride, qout.count, qout.seg, dyn, qout.prim,
                                       qout.t, qout.b1, qout.b2, no_rs, tm.counters(cnt), &kid, ord));
                tm.set_kernel(kid);
                cur ^= 1;
            }
        }
        for (int k = 0; k < 2; k++)      // the film needs every shadow contribution
            if (any_pending[k]) {
                HIPCHK(c, hipStreamWaitEvent(stream, c->ev_any[k], 0));
                any_pending[k] = false;
            }
        {
            StageTimer tm(c, stream, STAGE_SPLAT, -1, batch_no);
            HIPCHK(c, pt_launch_splat(stream, c->d_rc, *tile, *film, base, nb, c->st, film_accum, sample_L, cnt));
        }
    }
    if (stats) return read_counters(c, stream, stats);
    return PT_OK;
}

extern "C" int pt_film_resolve(pt_ctx *c, const double *film_accum, int32_t x_cnt, int32_t y_cnt, double *rgb,
                               void *stream_) {
    if (!c) return PT_ERR_ARG;
    if (!film_accum || !rgb || x_cnt < 0 || y_cnt < 0) return fail(c, PT_ERR_ARG, "film_resolve arguments");
    DeviceGuard dg(c->device);
    HIPCHK(c, dg.err);
    HIPCHK(c, pt_launch_film_resolve((hipStream_t)stream_, film_accum, (long)x_cnt * y_cnt, rgb));
    return PT_OK;
}

extern "C" int pt_film_to_u8(pt_ctx *c, const double *rgb, int32_t x_cnt, int32_t y_cnt, uint8_t *out, void *stream_) {
    if (!c) return PT_ERR_ARG;
    if (x_cnt < 0 || y_cnt < 0) return fail(c, PT_ERR_ARG, "film_to_u8 arguments");
    if ((long)x_cnt * y_cnt == 0) return PT_OK;      // np.max of an empty channel raises in the reference; nothing to write
    if (!rgb || !out) return fail(c, PT_ERR_ARG, "film_to_u8 arguments");
    DeviceGuard dg(c->device);
    HIPCHK(c, dg.err);
    HIPCHK(c, pt_launch_film_to_u8((hipStream_t)stream_, rgb, (long)x_cnt * y_cnt,
                                   c->d_counters + (size_t)PT_LOG_SLICES * PT_SLICE_WORDS, out));
    return PT_OK;
}
