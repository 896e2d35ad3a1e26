// pt_build.cpp — host-native BVH construction with the reference's exact behaviour.
//
// Restates BVH.__init__ / _recursive_build / _partition / _insertion_sort / _bucket_partition /
// _flatten_tree (aggregate/accelerator/bvh.py:68-107,110-150,152-275,277-293): the tree, the
// primitive order and every bound come out identical to the Python build, so primitive ids and
// traversal order (observable through Q12) are the reference's.  259 s of Python for the
// 177 k-triangle head_loop scene become a fraction of a second here.
//
// The build is inherently sequential in the way that matters for parity (the in-place partition
// loops below decide the primitive order), so it runs on the host; the kernels only consume the
// flattened result (pt_scene_upload).
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <limits>
#include <new>
#include <utility>
#include <vector>

#include "../../include/pytracer_hip.h"

namespace {

thread_local char g_build_err[256] = "";

int build_fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_build_err, sizeof(g_build_err), fmt, ap);
    va_end(ap);
    return code;
}

// BBox with the min/max of geometry/__init__.py:466-492 (Python min(a, b): b if b < a else a)
struct Box {
    double lo[3], hi[3];
    Box() {
        for (int k = 0; k < 3; k++) {
            lo[k] = std::numeric_limits<double>::infinity();
            hi[k] = -std::numeric_limits<double>::infinity();
        }
    }
    void add(const Box &o) {
        for (int k = 0; k < 3; k++) {
            if (o.lo[k] < lo[k]) lo[k] = o.lo[k];
            if (o.hi[k] > hi[k]) hi[k] = o.hi[k];
        }
    }
    void add_point(const double *p) {
        for (int k = 0; k < 3; k++) {
            if (p[k] < lo[k]) lo[k] = p[k];
            if (p[k] > hi[k]) hi[k] = p[k];
        }
    }
    // geometry/__init__.py:523-528
    double surface_area() const {
        const double dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        return 2. * (dx * dy + dx * dz + dy * dz);
    }
    // geometry/__init__.py:537-547
    int maximum_extent() const {
        const double dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        if (dx > dy && dx > dz) return 0;
        if (dy > dz) return 1;
        return 2;
    }
};

// BVH._BVHPrimitive, bvh.py:31-35
struct BuildPrim {
    int32_t idx;
    Box b;
    double c[3];
};

struct Task {
    int64_t start, end;
    int64_t patch;   // interior node whose `offset` (second child) this node becomes, or -1
    int depth;
};

// bvh.py:110-120.  The loop stops at i == k, so element k is never classified.
int64_t partition_middle(std::vector<BuildPrim> &d, int64_t start, int64_t end, double mid, int dim) {
    int64_t i = start, k = end - 1;
    while (i < k) {
        if (d[i].c[dim] < mid) i++;
        else { std::swap(d[i], d[k]); k--; }
    }
    return k;
}

// bvh.py:123-130
void insertion_sort(std::vector<BuildPrim> &d, int64_t start, int64_t end, int dim) {
    for (int64_t i = start; i < end; i++) {
        int64_t k = i;
        BuildPrim dk = d[k];
        while (k > start && dk.c[dim] < d[k - 1].c[dim]) { d[k] = d[k - 1]; k--; }
        d[k] = dk;
    }
}

inline int bucket_of(double c, double cmin, double diff, int n_buckets) {
    int b = (int)(n_buckets * ((c - cmin) / diff));   // INT() truncates, bvh.py:139,221
    if (b == n_buckets) b -= 1;
    return b;
}

// bvh.py:133-150
int64_t partition_bucket(std::vector<BuildPrim> &d, int64_t start, int64_t end, int min_cost_idx, int n_buckets,
                         int dim, const Box &cb) {
    int64_t i = start, k = end - 1;
    const double diff = cb.hi[dim] - cb.lo[dim];
    while (i < k) {
        const int b = bucket_of(d[i].c[dim], cb.lo[dim], diff, n_buckets);
        if (b <= min_cost_idx) i++;
        else { std::swap(d[i], d[k]); k--; }
    }
    return k;
}

}   // namespace

extern "C" const char *pt_build_last_error(void) { return g_build_err; }

static int bvh_build_impl(const double *prim_bounds, int64_t n, int32_t max_prim_per_node, int32_t method,
                          double *node_bounds, int32_t *node_meta, int32_t *prim_order, int64_t *n_nodes,
                          int32_t *max_depth) {
    g_build_err[0] = 0;
    if (n < 0 || n > 0x3FFFFFFF) return build_fail(PT_ERR_ARG, "primitive count %lld out of range", (long long)n);
    if (!n_nodes) return build_fail(PT_ERR_ARG, "null n_nodes");
    *n_nodes = 0;
    if (max_depth) *max_depth = 0;
    if (n == 0) return PT_OK;   // bvh.py:85-87: nodes = None
    if (!prim_bounds || !node_bounds || !node_meta || !prim_order) return build_fail(PT_ERR_ARG, "null array");
    if (method != PT_BVH_SAH && method != PT_BVH_MIDDLE) return build_fail(PT_ERR_ARG, "split method %d", method);
    if (max_prim_per_node < 1) return build_fail(PT_ERR_ARG, "max_prim_per_node %d", max_prim_per_node);

    std::vector<BuildPrim> data((size_t)n);
    for (int64_t i = 0; i < n; i++) {
        BuildPrim &p = data[(size_t)i];
        p.idx = (int32_t)i;
        for (int k = 0; k < 3; k++) {
            p.b.lo[k] = prim_bounds[6 * i + k];
            p.b.hi[k] = prim_bounds[6 * i + 3 + k];
            p.c[k] = .5 * (p.b.lo[k] + p.b.hi[k]);   // bvh.py:35
        }
    }

    // _recursive_build numbers nodes in creation order (segment[2], bvh.py:162) and _flatten_tree
    // numbers them in pre-order (bvh.py:277-293): the same order, so linear nodes are emitted
    // directly.  An explicit task stack replaces the recursion (first child on top).
    std::vector<Task> todo;
    todo.push_back({0, n, -1, 1});
    int64_t count = 0, n_ordered = 0;
    int deepest = 0;
    const int n_buckets = 12;
    while (!todo.empty()) {
        const Task t = todo.back();
        todo.pop_back();
        const int64_t start = t.start, end = t.end;
        const int64_t me = count++;
        if (t.patch >= 0) node_meta[3 * t.patch] = (int32_t)me;
        if (t.depth > deepest) deepest = t.depth;

        Box bbox;
        for (int64_t i = start; i < end; i++) bbox.add(data[(size_t)i].b);
        for (int k = 0; k < 3; k++) {
            node_bounds[6 * me + k] = bbox.lo[k];
            node_bounds[6 * me + 3 + k] = bbox.hi[k];
        }
        const int64_t n_prim = end - start;
        auto make_leaf = [&]() {
            node_meta[3 * me] = (int32_t)n_ordered;
            node_meta[3 * me + 1] = (int32_t)n_prim;
            node_meta[3 * me + 2] = 0;
            for (int64_t i = start; i < end; i++) prim_order[n_ordered++] = data[(size_t)i].idx;
        };
        auto make_interior = [&](int dim, int64_t mid) {
            node_meta[3 * me] = -1;   // patched when the second child is created
            node_meta[3 * me + 1] = 0;
            node_meta[3 * me + 2] = dim;
            todo.push_back({mid, end, me, t.depth + 1});
            todo.push_back({start, mid, -1, t.depth + 1});
        };
        if (n_prim == 1) { make_leaf(); continue; }

        int64_t mid = (start + end) / 2;
        Box cb;
        for (int64_t i = start; i < end; i++) cb.add_point(data[(size_t)i].c);
        const int dim = cb.maximum_extent();
        if (cb.hi[dim] == cb.lo[dim]) {   // bvh.py:187-203
            if (n_prim <= max_prim_per_node) make_leaf();
            else make_interior(dim, mid);
            continue;
        }
        if (method == PT_BVH_MIDDLE) {
            const double pmid = .5 * (cb.hi[dim] + cb.lo[dim]);
            mid = partition_middle(data, start, end, pmid, dim);
        } else if (n_prim <= 4) {
            insertion_sort(data, start, end, dim);
        } else {
            int cnt[n_buckets] = {0};
            Box bb[n_buckets];
            const double diff = cb.hi[dim] - cb.lo[dim];
            for (int64_t i = start; i < end; i++) {
                const int b = bucket_of(data[(size_t)i].c[dim], cb.lo[dim], diff, n_buckets);
                if (b < 0 || b >= n_buckets)
                    return build_fail(PT_ERR_ARG, "non-finite centroid: the reference asserts here (bvh.py:227)");
                cnt[b]++;
                bb[b].add(data[(size_t)i].b);
            }
            double cost[n_buckets - 1];
            const double sa = bbox.surface_area();
            for (int i = 0; i < n_buckets - 1; i++) {
                Box b0, b1;
                int64_t c0 = 0, c1 = 0;
                for (int j = 0; j <= i; j++) { b0.add(bb[j]); c0 += cnt[j]; }
                for (int j = i + 1; j < n_buckets; j++) { b1.add(bb[j]); c1 += cnt[j]; }
                cost[i] = .125 + ((double)c0 * b0.surface_area() + (double)c1 * b1.surface_area()) / sa;
            }
            // np.argmin: first minimum; a NaN wins over everything (first NaN)
            int best = 0;
            for (int i = 0; i < n_buckets - 1; i++) {
                if (std::isnan(cost[i])) { best = i; break; }
                if (cost[i] < cost[best]) best = i;
            }
            if (n_prim > max_prim_per_node || cost[best] < (double)n_prim) {
                mid = partition_bucket(data, start, end, best, n_buckets, dim, cb);
                if (mid == start) mid += 1;
                else if (mid == end) mid -= 1;
            } else {
                return build_fail(PT_ERR_ARG,
                                  "max_prim_per_node > 4 reaches the leaf branch of bvh.py:261-265, which raises "
                                  "TypeError in the reference");
            }
        }
        if (mid == start || mid == end)
            return build_fail(PT_ERR_ARG, "empty side after the split: the reference asserts here (bvh.py:267-268)");
        make_interior(dim, mid);
    }
    *n_nodes = count;
    if (max_depth) *max_depth = deepest;
    return PT_OK;
}

// No C++ exception crosses the C ABI: an allocation failure on a huge input is PT_ERR_CAPACITY.
extern "C" int pt_bvh_build(const double *prim_bounds, int64_t n, int32_t max_prim_per_node, int32_t method,
                            double *node_bounds, int32_t *node_meta, int32_t *prim_order, int64_t *n_nodes,
                            int32_t *max_depth) {
    try {
        return bvh_build_impl(prim_bounds, n, max_prim_per_node, method, node_bounds, node_meta, prim_order, n_nodes,
                              max_depth);
    } catch (const std::bad_alloc &) {
        return build_fail(PT_ERR_CAPACITY, "out of host memory building a tree over %lld primitives", (long long)n);
    } catch (...) {
        return build_fail(PT_ERR_ARG, "unexpected C++ exception in pt_bvh_build");
    }
}
