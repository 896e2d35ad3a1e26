// pt_subdiv.cpp — host-native Loop subdivision with the reference's exact behaviour.
//
// Restates LoopSubdiv.__init__ and LoopSubdiv.refine (shape/loopsubdiv.py:195-247, 293-446) with its
// SDVertex / SDFace helpers (:79-128, :145-167): the same vertex and face ORDER (children of the old
// vertices first, then one vertex per edge in order of first appearance; four children per face), the
// same one-ring walks and therefore the same floating-point summation order, the limit-surface push
// at the end.  The output equals the reference's TriangleMesh (P in object space, vertexIndex) bit
// for bit, so a scene built from it gets the reference's BVH and primitive ids (pt_bvh_build).
// The vertex normals the reference also computes (:397-430) never reach the shading path (a mesh
// with N but no S shades with its geometric frame, triangle.py:325) and are not produced.
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <new>
#include <unordered_map>
#include <vector>

#include "../../include/pytracer_hip.h"

namespace {

thread_local char g_subdiv_err[256] = "";

int subdiv_fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_subdiv_err, sizeof(g_subdiv_err), fmt, ap);
    va_end(ap);
    return code;
}

struct P3 {
    double x, y, z;
};
inline P3 operator*(double s, const P3 &p) { return {s * p.x, s * p.y, s * p.z}; }
inline P3 operator+(const P3 &a, const P3 &b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }

struct SDVertex {   // loopsubdiv.py:54-65
    P3 P;
    int startFace = -1, child = -1;
    bool regular = false, boundary = false;
};
struct SDFace {     // loopsubdiv.py:130-140
    int v[3] = {-1, -1, -1};
    int f[3] = {-1, -1, -1};
    int children[4] = {-1, -1, -1, -1};
};

inline int NEXT(int i) { return (i + 1) % 3; }
inline int PREV(int i) { return (i + 2) % 3; }

struct Mesh {
    std::vector<SDVertex> v;
    std::vector<SDFace> f;
    bool bad = false;   // a walk left the manifold (the reference raises or loops there)

    int vnum(int face, int vert) {   // :145-149
        for (int i = 0; i < 3; i++)
            if (f[face].v[i] == vert) return i;
        bad = true;
        return 0;
    }
    int next_face(int face, int vert) { return f[face].f[vnum(face, vert)]; }
    int prev_face(int face, int vert) { return f[face].f[PREV(vnum(face, vert))]; }
    int next_vert(int face, int vert) { return f[face].v[NEXT(vnum(face, vert))]; }
    int prev_vert(int face, int vert) { return f[face].v[PREV(vnum(face, vert))]; }
    int other_vert(int face, int v0, int v1) {   // :163-167
        for (int i = 0; i < 3; i++)
            if (f[face].v[i] != v0 && f[face].v[i] != v1) return f[face].v[i];
        bad = true;
        return f[face].v[0];
    }
    int valence(int vert) {   // :79-101
        const size_t cap = f.size() + 2;
        const int start = v[vert].startFace;
        size_t guard = 0;
        int nf = 1;
        if (!v[vert].boundary) {
            int fc = next_face(start, vert);
            while (fc != start) {
                if (fc < 0 || ++guard > cap) { bad = true; return nf; }
                nf++;
                fc = next_face(fc, vert);
            }
            return nf;
        }
        int fc = next_face(start, vert);
        while (fc >= 0) {
            if (++guard > cap) { bad = true; return nf; }
            nf++;
            fc = next_face(fc, vert);
        }
        fc = prev_face(start, vert);
        while (fc >= 0) {
            if (++guard > cap) { bad = true; return nf; }
            nf++;
            fc = prev_face(fc, vert);
        }
        return nf + 1;
    }
    void one_ring(int vert, std::vector<P3> &ring) {   // :103-128
        ring.clear();
        const size_t cap = f.size() + 2;
        size_t guard = 0;
        const int start = v[vert].startFace;
        if (!v[vert].boundary) {
            int fc = start;
            ring.push_back(v[next_vert(fc, vert)].P);
            fc = next_face(fc, vert);
            while (fc != start) {
                if (fc < 0 || ++guard > cap) { bad = true; return; }
                ring.push_back(v[next_vert(fc, vert)].P);
                fc = next_face(fc, vert);
            }
        } else {
            int fc = start, f2 = next_face(fc, vert);
            while (f2 >= 0) {
                if (++guard > cap) { bad = true; return; }
                fc = f2;
                f2 = next_face(fc, vert);
            }
            ring.push_back(v[next_vert(fc, vert)].P);
            ring.push_back(v[prev_vert(fc, vert)].P);
            fc = prev_face(fc, vert);
            while (fc >= 0) {
                if (++guard > cap) { bad = true; return; }
                ring.push_back(v[prev_vert(fc, vert)].P);
                fc = prev_face(fc, vert);
            }
        }
    }
    // :263-269
    P3 weight_one_ring(int vert, double beta, std::vector<P3> &ring) {
        const int val = valence(vert);
        one_ring(vert, ring);
        P3 P = (1. - val * beta) * v[vert].P;
        for (const P3 &p : ring) P = P + beta * p;
        return P;
    }
    // :271-276
    P3 weight_boundary(int vert, double beta, std::vector<P3> &ring) {
        one_ring(vert, ring);
        P3 P = (1. - 2. * beta) * v[vert].P;
        if (ring.empty()) { bad = true; return P; }
        return P + beta * (ring.front() + ring.back());
    }
};

inline double beta_of(int valence) { return valence == 3 ? .1875 : 3. / (8. * valence); }            // :253-257
inline double gamma_of(int valence) { return 1. / (valence + 3. / (8. + beta_of(valence))); }          // :260-261

inline uint64_t edge_key(int a, int b) {   // SDEdge orders its ends (by id() there, by index here; only equality matters)
    const uint32_t lo = (uint32_t)(a < b ? a : b), hi = (uint32_t)(a < b ? b : a);
    return ((uint64_t)lo << 32) | hi;
}

}   // namespace

extern "C" const char *pt_subdiv_last_error(void) { return g_subdiv_err; }

static int loop_subdiv_impl(const double *P, int64_t nv, const int32_t *vi, int64_t nf, int32_t n_levels,
                            double *P_out, int64_t p_cap, int32_t *idx_out, int64_t idx_cap, int64_t *nv_out,
                            int64_t *nf_out) {
    g_subdiv_err[0] = 0;
    if (!nv_out || !nf_out) return subdiv_fail(PT_ERR_ARG, "null count pointers");
    *nv_out = *nf_out = 0;
    if (nv <= 0 || nf <= 0 || !P || !vi) return subdiv_fail(PT_ERR_ARG, "empty mesh");
    if (n_levels < 0 || n_levels > 10) return subdiv_fail(PT_ERR_ARG, "n_levels %d", n_levels);
    if (nv > 0x3FFFFFFF || nf > 0x3FFFFFFF) return subdiv_fail(PT_ERR_CAPACITY, "mesh too large");
    // nf * 4^levels faces come out: refuse what cannot be indexed with 32 bits before allocating for it
    if (((long double)nf) * powl(4.0L, (long double)n_levels) > (long double)0x3FFFFFFF)
        return subdiv_fail(PT_ERR_CAPACITY, "%lld faces x 4^%d levels exceed 2^30 triangles", (long long)nf, n_levels);
    for (int64_t k = 0; k < 3 * nf; k++)
        if (vi[k] < 0 || vi[k] >= nv) return subdiv_fail(PT_ERR_ARG, "vertex index %d out of range at %lld", vi[k], (long long)k);

    Mesh m;
    m.v.resize((size_t)nv);
    m.f.resize((size_t)nf);
    for (int64_t i = 0; i < nv; i++) m.v[(size_t)i].P = {P[3 * i], P[3 * i + 1], P[3 * i + 2]};
    // face → vertex refs; startFace = the LAST face that names the vertex (:204-210)
    for (int64_t i = 0; i < nf; i++)
        for (int j = 0; j < 3; j++) {
            m.f[(size_t)i].v[j] = vi[3 * i + j];
            m.v[(size_t)vi[3 * i + j]].startFace = (int)i;
        }
    // neighbour refs through the edge dictionary (:213-229)
    {
        struct E { int f0, f0edge; };
        std::unordered_map<uint64_t, E> edges;
        edges.reserve((size_t)nf * 2);
        for (int i = 0; i < (int)nf; i++)
            for (int en = 0; en < 3; en++) {
                const uint64_t key = edge_key(m.f[i].v[en], m.f[i].v[NEXT(en)]);
                auto it = edges.find(key);
                if (it != edges.end()) {
                    m.f[it->second.f0].f[it->second.f0edge] = i;
                    m.f[i].f[en] = it->second.f0;
                    edges.erase(it);
                } else {
                    edges[key] = {i, en};
                }
            }
    }
    // boundary / regular (:232-247)
    for (int i = 0; i < (int)nv; i++) {
        SDVertex &vx = m.v[i];
        if (vx.startFace < 0)
            return subdiv_fail(PT_ERR_ARG, "vertex %d belongs to no face: the reference raises AttributeError (loopsubdiv.py:235)", i);
        int fc = m.next_face(vx.startFace, i);
        size_t guard = 0;
        while (fc >= 0 && fc != vx.startFace) {
            if (++guard > m.f.size() + 2) return subdiv_fail(PT_ERR_ARG, "non-manifold fan around vertex %d", i);
            fc = m.next_face(fc, i);
        }
        vx.boundary = fc < 0;
        const int val = m.valence(i);
        vx.regular = (!vx.boundary && val == 6) || (vx.boundary && val == 4);
    }
    if (m.bad) return subdiv_fail(PT_ERR_ARG, "inconsistent mesh topology (the reference raises RuntimeError in vnum/other_vert)");

    std::vector<P3> ring;
    for (int level = 0; level < n_levels; level++) {
        Mesh n;
        const int ov = (int)m.v.size(), of = (int)m.f.size();
        n.v.reserve((size_t)ov + (size_t)of * 3 / 2 + 8);
        n.v.resize((size_t)ov);
        n.f.resize((size_t)of * 4);
        for (int i = 0; i < ov; i++) {          // :300-304
            m.v[i].child = i;
            n.v[i].regular = m.v[i].regular;
            n.v[i].boundary = m.v[i].boundary;
        }
        for (int i = 0; i < of; i++)            // :306-309
            for (int k = 0; k < 4; k++) m.f[i].children[k] = 4 * i + k;
        // even vertices (:313-324)
        for (int i = 0; i < ov; i++) {
            if (!m.v[i].boundary) {
                if (m.v[i].regular) n.v[i].P = m.weight_one_ring(i, .0625, ring);
                else n.v[i].P = m.weight_one_ring(i, beta_of(m.valence(i)), ring);
            } else {
                n.v[i].P = m.weight_boundary(i, .125, ring);
            }
        }
        // odd vertices, one per edge in order of first appearance (:327-349)
        std::unordered_map<uint64_t, int> edge_verts;
        edge_verts.reserve((size_t)of * 2);
        for (int i = 0; i < of; i++)
            for (int k = 0; k < 3; k++) {
                const int a = m.f[i].v[k], b = m.f[i].v[NEXT(k)];
                const uint64_t key = edge_key(a, b);
                if (edge_verts.find(key) != edge_verts.end()) continue;
                SDVertex nv_;
                nv_.regular = true;
                nv_.boundary = m.f[i].f[k] < 0;
                nv_.startFace = m.f[i].children[3];
                if (nv_.boundary) {
                    nv_.P = .5 * m.v[a].P + .5 * m.v[b].P;
                } else {
                    nv_.P = .375 * m.v[a].P + .375 * m.v[b].P + .125 * m.v[m.other_vert(i, a, b)].P +
                            .125 * m.v[m.other_vert(m.f[i].f[k], a, b)].P;
                }
                edge_verts[key] = (int)n.v.size();
                n.v.push_back(nv_);
            }
        // startFace of the even vertices (:354-356)
        for (int i = 0; i < ov; i++) {
            const int sf = m.v[i].startFace;
            n.v[i].startFace = m.f[sf].children[m.vnum(sf, i)];
        }
        // neighbour refs of the new faces (:359-369)
        for (int i = 0; i < of; i++) {
            const SDFace &ff = m.f[i];
            for (int k = 0; k < 3; k++) {
                n.f[ff.children[3]].f[k] = ff.children[NEXT(k)];
                n.f[ff.children[k]].f[NEXT(k)] = ff.children[3];
                int f2 = ff.f[k];
                n.f[ff.children[k]].f[k] = f2 < 0 ? -1 : m.f[f2].children[m.vnum(f2, ff.v[k])];
                f2 = ff.f[PREV(k)];
                n.f[ff.children[k]].f[PREV(k)] = f2 < 0 ? -1 : m.f[f2].children[m.vnum(f2, ff.v[k])];
            }
        }
        // vertex refs of the new faces (:372-380)
        for (int i = 0; i < of; i++) {
            const SDFace &ff = m.f[i];
            for (int k = 0; k < 3; k++) {
                n.f[ff.children[k]].v[k] = m.v[ff.v[k]].child;
                const int vert = edge_verts[edge_key(ff.v[k], ff.v[NEXT(k)])];
                n.f[ff.children[k]].v[NEXT(k)] = vert;
                n.f[ff.children[NEXT(k)]].v[k] = vert;
                n.f[ff.children[3]].v[k] = vert;
            }
        }
        if (m.bad) return subdiv_fail(PT_ERR_ARG, "inconsistent mesh topology at level %d", level);
        m.v.swap(n.v);
        m.f.swap(n.f);
    }
    // push to the limit surface (:387-395): all from the un-pushed positions
    std::vector<P3> limit(m.v.size());
    for (int i = 0; i < (int)m.v.size(); i++) {
        if (m.v[i].boundary) limit[i] = m.weight_boundary(i, .2, ring);
        else limit[i] = m.weight_one_ring(i, gamma_of(m.valence(i)), ring);
    }
    if (m.bad) return subdiv_fail(PT_ERR_ARG, "inconsistent mesh topology in the limit pass");
    // the tangent pass (:397-430) accumulates into S, T before they exist unless a boundary vertex came
    // first: a mesh whose first vertex is interior makes the reference raise UnboundLocalError
    if (!m.v[0].boundary)
        return subdiv_fail(PT_ERR_ARG, "vertex 0 is an interior vertex: LoopSubdiv.refine raises UnboundLocalError in "
                                       "its tangent pass (loopsubdiv.py:409), e.g. for every closed mesh");
    *nv_out = (int64_t)m.v.size();
    *nf_out = (int64_t)m.f.size();
    if (!P_out || !idx_out || p_cap < *nv_out || idx_cap < 3 * *nf_out)
        return subdiv_fail(PT_ERR_CAPACITY, "output needs %lld vertices and %lld indices", (long long)*nv_out,
                           (long long)(3 * *nf_out));
    for (size_t i = 0; i < m.v.size(); i++) {
        P_out[3 * i] = limit[i].x; P_out[3 * i + 1] = limit[i].y; P_out[3 * i + 2] = limit[i].z;
    }
    for (size_t i = 0; i < m.f.size(); i++)      // :433-439
        for (int j = 0; j < 3; j++) idx_out[3 * i + j] = m.f[i].v[j];
    return PT_OK;
}

// No C++ exception crosses the C ABI: an allocation failure (4^levels growth) is PT_ERR_CAPACITY.
extern "C" int pt_loop_subdiv(const double *P, int64_t nv, const int32_t *vi, int64_t nf, int32_t n_levels,
                              double *P_out, int64_t p_cap, int32_t *idx_out, int64_t idx_cap, int64_t *nv_out,
                              int64_t *nf_out) {
    try {
        return loop_subdiv_impl(P, nv, vi, nf, n_levels, P_out, p_cap, idx_out, idx_cap, nv_out, nf_out);
    } catch (const std::bad_alloc &) {
        return subdiv_fail(PT_ERR_CAPACITY, "out of host memory subdividing %lld faces %d times", (long long)nf, n_levels);
    } catch (...) {
        return subdiv_fail(PT_ERR_ARG, "unexpected C++ exception in pt_loop_subdiv");
    }
}
