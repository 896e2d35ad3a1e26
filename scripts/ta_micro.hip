// Micro-benchmark behind DESIGN.md "Trace: what was measured": how fast can a wave fetch one 64-byte record per
// lane at unrelated addresses (what a BVH node visit is)?
//   own : every lane issues four 16-byte loads of its own record (what k_trace does);
//   coop: four lanes fetch the four 16-byte quarters of ONE record with one instruction (so a load instruction
//         touches 16 lines instead of 64), the wave transposes through LDS and every lane reads its record back.
// Each lane follows a random permutation stored in the records (the next address depends on the data, like a
// traversal).  `active` of every 64 lanes take part (divergence).  Build: hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

struct Rec { double b[6]; int next; int pad[3]; };
static_assert(sizeof(Rec) == 64, "");

template <int NL>
__global__ __launch_bounds__(256) void k_own(const Rec *__restrict__ recs, const int *__restrict__ start, int steps,
                                             int active, double *__restrict__ out) {
    const long gid = (long)blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63;
    int i = start[gid];
    double acc = 0.;
    if (((lane * 37) & 63) < active) {
        for (int s = 0; s < steps; s++) {
            const double2 *p = (const double2 *)(recs + i);
            // NL of the record's four 16-byte quarters are loaded (the link is in the last one)
            const double2 d = p[3];
            double part = 0.;
            if (NL > 1) { const double2 a = p[0]; part += a.x + a.y; }
            if (NL > 2) { const double2 b = p[1]; part += b.x + b.y; }
            if (NL > 3) { const double2 c = p[2]; part += c.x + c.y; }
            acc += part;
            i = __double2loint(d.x);
        }
    }
    out[gid] = acc + i;
}

__global__ __launch_bounds__(256) void k_coop(const Rec *__restrict__ recs, const int *__restrict__ start, int steps,
                                              int active, double *__restrict__ out) {
    __shared__ double2 tile[4][4 * 64];      // per wave: [quarter][slot]
    __shared__ int idx[4][64];
    const long gid = (long)blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const bool on = ((lane * 37) & 63) < active;
    int i = start[gid];
    double acc = 0.;
    double2 *t = tile[w];
    int *ix = idx[w];
    const int q = lane & 3, s0 = lane >> 2;
    for (int s = 0; s < steps; s++) {
        ix[lane] = on ? i : -1;
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int j = ix[s0 + 16 * k];
            if (j >= 0) t[q * 64 + s0 + 16 * k] = ((const double2 *)(recs + j))[q];
        }
        __builtin_amdgcn_wave_barrier();
        if (on) {
            const double2 a = t[lane], b = t[64 + lane], c = t[128 + lane], d = t[192 + lane];
            acc += (a.x + a.y) + (b.x + b.y) + (c.x + c.y);
            i = __double2loint(d.x);
        }
        __builtin_amdgcn_wave_barrier();
    }
    out[gid] = acc + i;
}

int main(int argc, char **argv) {
    const int n_rec = argc > 1 ? atoi(argv[1]) : 5537;
    const int steps = 256;
    const long n = 256L * 256 * 24;          // resident-sized grid: 256 CUs x 6 blocks x 4 waves... x4 rounds
    std::vector<Rec> h(n_rec);
    std::vector<int> perm(n_rec);
    srand(1);
    for (int i = 0; i < n_rec; i++) perm[i] = i;
    for (int i = n_rec - 1; i > 0; i--) { int j = rand() % (i + 1); std::swap(perm[i], perm[j]); }
    for (int i = 0; i < n_rec; i++) {
        for (int k = 0; k < 6; k++) h[i].b[k] = 1e-3 * (i + k);
        h[i].next = perm[i]; h[i].pad[0] = h[i].pad[1] = h[i].pad[2] = 0;
    }
    std::vector<int> st(n);
    for (long i = 0; i < n; i++) st[i] = rand() % n_rec;
    Rec *d_r; int *d_s; double *d_o;
    hipMalloc(&d_r, n_rec * sizeof(Rec)); hipMalloc(&d_s, n * sizeof(int)); hipMalloc(&d_o, n * sizeof(double));
    hipMemcpy(d_r, h.data(), n_rec * sizeof(Rec), hipMemcpyHostToDevice);
    hipMemcpy(d_s, st.data(), n * sizeof(int), hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int acts[] = {64, 32, 18, 8};
    std::vector<double> o1(n), o2(n);
    for (int a : acts) {
        float ms[2] = {0, 0};
        for (int v = 0; v < 2; v++) {
            for (int rep = 0; rep < 3; rep++) {
                hipEventRecord(e0);
                if (v == 0) hipLaunchKernelGGL(k_own<4>, dim3(n / 256), dim3(256), 0, 0, d_r, d_s, steps, a, d_o);
                else hipLaunchKernelGGL(k_coop, dim3(n / 256), dim3(256), 0, 0, d_r, d_s, steps, a, d_o);
                hipEventRecord(e1); hipEventSynchronize(e1);
                hipEventElapsedTime(&ms[v], e0, e1);
            }
            hipMemcpy(v == 0 ? o1.data() : o2.data(), d_o, n * sizeof(double), hipMemcpyDeviceToHost);
        }
        long bad = 0;
        for (long i = 0; i < n; i++) bad += o1[i] != o2[i];
        const double fetches = (double)n * a / 64. * steps;
        float msn[4] = {0, 0, 0, 0};
        for (int nl = 1; nl <= 3; nl++)
            for (int rep = 0; rep < 3; rep++) {
                hipEventRecord(e0);
                if (nl == 1) hipLaunchKernelGGL(k_own<1>, dim3(n / 256), dim3(256), 0, 0, d_r, d_s, steps, a, d_o);
                if (nl == 2) hipLaunchKernelGGL(k_own<2>, dim3(n / 256), dim3(256), 0, 0, d_r, d_s, steps, a, d_o);
                if (nl == 3) hipLaunchKernelGGL(k_own<3>, dim3(n / 256), dim3(256), 0, 0, d_r, d_s, steps, a, d_o);
                hipEventRecord(e1); hipEventSynchronize(e1);
                hipEventElapsedTime(&msn[nl], e0, e1);
            }
        printf("   loads per step 1/2/3/4: %.1f / %.1f / %.1f / %.1f G fetch/s\n", fetches / msn[1] / 1e6,
               fetches / msn[2] / 1e6, fetches / msn[3] / 1e6, fetches / ms[0] / 1e6);
        printf("records %d active %2d/64: own %.3f ms (%.1f G fetch/s)  coop %.3f ms (%.1f G fetch/s)  mismatches %ld\n",
               n_rec, a, ms[0], fetches / ms[0] / 1e6, ms[1], fetches / ms[1] / 1e6, bad);
    }
    return 0;
}
