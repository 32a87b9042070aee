// One diffusion step y = Q x of a large BSR-3 operator (cfg3 shape: 2M block rows, ~19 blocks per row, banded after the
// breadth-first relabelling) -- which kernel shape reaches the HBM rate?
//   K0  the lane-group kernel of escat.hip (k_bsr3_step<7,0>): 21 lanes per row, each lane streams 12 B of a block
//   K1  K0 without the dependent x gather (x row taken from the lane index): what the rowptr -> col -> x chain costs
//   K2  staged: a workgroup copies the contiguous block range of RPB rows into LDS with 16-byte loads issued together,
//       gathers the x rows of those blocks into LDS, then LPR lanes per row compute out of LDS
//   build: hipcc -O3 --offload-arch=gfx950 -o step_bench step_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <algorithm>
#include <cmath>

__host__ __device__ inline unsigned hash32(unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

// pattern: points of a W x W grid numbered row by row (what a breadth-first relabelling of a surface kNN graph looks
// like: band ~ 2W), each linked to its `deg` nearest grid points of the 5 x 5 stencil
__constant__ int st_di[25] = {0, 0, 0, 1, -1, 1, 1, -1, -1, 0, 0, 2, -2, 1, 1, -1, -1, 2, 2, -2, -2, 2, 2, -2, -2};
__constant__ int st_dj[25] = {0, 1, -1, 0, 0, 1, -1, 1, -1, 2, -2, 0, 0, 2, -2, 2, -2, 1, -1, 1, -1, 2, -2, 2, -2};
__global__ void k_fill(const int *rowptr, int *col, float *blk, int nb, int W)
{
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= nb) return;
    const int e0 = rowptr[r], e1 = rowptr[r + 1];
    const int i = r / W, j = r % W;
    for (int e = e0; e < e1; ++e) {
        const int t = e - e0;
        long long c = (long long)(i + st_di[t]) * W + (j + st_dj[t]);
        c = c < 0 ? 0 : c >= nb ? nb - 1 : c;
        col[e] = (int)c;
        for (int q = 0; q < 9; ++q) blk[(size_t)e * 9 + q] = (hash32(e * 9u + q) & 0xffff) * (1.f / 65536.f) - 0.5f;
    }
}
__global__ void k_fillx(float *x, size_t n) { const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; if (i < n) x[i] = (hash32((unsigned)i * 7u + 1u) & 0xffff) * (1.f / 65536.f); }

template <int g, bool GATHER>
__global__ void __launch_bounds__(256)
k_step_lanes(const int *__restrict__ rowptr, const int *__restrict__ col, const float *__restrict__ blk,
             const float *__restrict__ xin, float *__restrict__ yout, int64_t nb)
{
    constexpr int GL = 3 * g, RPW = 64 / GL;
    const int lane = threadIdx.x & 63;
    const int sub = lane / GL, l = lane % GL, k = l / 3, a = l % 3;
    const int64_t waves = (int64_t)gridDim.x * 4;
    for (int64_t base = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW; base < nb; base += waves * RPW) {
        const int64_t brow = base + sub;
        const bool live = sub < RPW && brow < nb;
        float acc = 0.f;
        if (live) {
            const int e1 = rowptr[brow + 1];
            for (int e = rowptr[brow] + k; e < e1; e += g) {
                const float *b = blk + (int64_t)e * 9 + a * 3;
                const float *x = xin + (GATHER ? (int64_t)col[e] * 3 : (int64_t)(brow + k) * 3);
                acc += b[0] * x[0] + b[1] * x[1] + b[2] * x[2];
            }
        }
        float tot = 0.f;
#pragma unroll
        for (int kk = 0; kk < g; ++kk) tot += __shfl(acc, live ? sub * GL + 3 * kk + a : lane, 64);
        if (live && k == 0) yout[3 * brow + a] = tot;
    }
}

template <int RPB, int CAP, int VARIANT>
__global__ void __launch_bounds__(256)
k_step_lds(const int *__restrict__ rowptr, const int *__restrict__ col, const float *__restrict__ blk,
           const float *__restrict__ xin, float *__restrict__ yout, int nb, int64_t nnzb)
{
    constexpr int LPR = 256 / RPB;
    constexpr int NQ = (CAP * 9 + 3) / 4 + 1, QPT = (NQ + 255) / 256, EPT = (CAP + 255) / 256;
    extern __shared__ float lds[];
    float *vL = lds, *xL = lds + NQ * 4;
    const int trips = (nb + RPB - 1) / RPB;
    for (int trip = blockIdx.x; trip < trips; trip += gridDim.x) {
        const int r0 = trip * RPB, r1 = r0 + RPB < nb ? r0 + RPB : nb;
        const int e0 = rowptr[r0], e1 = rowptr[r1], ne = e1 - e0;
        const int64_t f0 = (int64_t)e0 * 9, f1 = (int64_t)e1 * 9, a0 = f0 & ~(int64_t)3;
        const int nq = (int)((f1 - a0 + 3) >> 2);
        int c[EPT];
#pragma unroll
        for (int j = 0; j < EPT; ++j) { const int i = threadIdx.x + 256 * j; c[j] = (VARIANT & 1) ? (r0 + (i >> 4)) : (i < ne ? col[e0 + i] : 0); }
        int rp = 0;
        if ((VARIANT & 2) && threadIdx.x <= r1 - r0) rp = rowptr[r0 + threadIdx.x] - e0;
        float4 v[QPT];
#pragma unroll
        for (int j = 0; j < QPT; ++j) {
            const int q = threadIdx.x + 256 * j;
            const int64_t gq = a0 + 4 * (int64_t)q;
            if (q < nq) {
                if (gq + 4 <= nnzb * 9) {
                    if (VARIANT & 4) { typedef float vf4 __attribute__((ext_vector_type(4))); const vf4 t = __builtin_nontemporal_load(reinterpret_cast<const vf4 *>(blk + gq)); v[j] = make_float4(t.x, t.y, t.z, t.w); }
                    else v[j] = *reinterpret_cast<const float4 *>(blk + gq);
                }
                else { v[j].x = blk[gq]; v[j].y = gq + 1 < nnzb * 9 ? blk[gq + 1] : 0.f; v[j].z = gq + 2 < nnzb * 9 ? blk[gq + 2] : 0.f; v[j].w = 0.f; }
            }
        }
        float xr[EPT][3];
#pragma unroll
        for (int j = 0; j < EPT; ++j) {
            if (VARIANT & 8) { const float4 t = *reinterpret_cast<const float4 *>(xin + (int64_t)c[j] * 4); xr[j][0] = t.x; xr[j][1] = t.y; xr[j][2] = t.z; }
            else { const float *xp = xin + (int64_t)c[j] * 3; xr[j][0] = xp[0]; xr[j][1] = xp[1]; xr[j][2] = xp[2]; }
        }
#pragma unroll
        for (int j = 0; j < QPT; ++j) { const int q = threadIdx.x + 256 * j; if (q < nq) *reinterpret_cast<float4 *>(vL + 4 * q) = v[j]; }
#pragma unroll
        for (int j = 0; j < EPT; ++j) {
            const int i = threadIdx.x + 256 * j;
            if (i < ne) { xL[3 * i] = xr[j][0]; xL[3 * i + 1] = xr[j][1]; xL[3 * i + 2] = xr[j][2]; }
        }
        int *rpL = reinterpret_cast<int *>(xL + CAP * 3);
        if ((VARIANT & 2) && threadIdx.x <= r1 - r0) rpL[threadIdx.x] = rp;
        __syncthreads();
        const int row = r0 + threadIdx.x / LPR, k = threadIdx.x % LPR;
        float a0s = 0.f, a1s = 0.f, a2s = 0.f;
        if (row < r1) {
            const int s0 = (VARIANT & 2) ? rpL[row - r0] : rowptr[row] - e0, s1 = (VARIANT & 2) ? rpL[row - r0 + 1] : rowptr[row + 1] - e0;
            const float *vb = vL + (int)(f0 - a0);
            for (int i = s0 + k; i < s1; i += LPR) {
                const float *b = vb + i * 9; const float *x = xL + 3 * i;
                const float x0 = x[0], x1 = x[1], x2 = x[2];
                a0s += b[0] * x0 + b[1] * x1 + b[2] * x2;
                a1s += b[3] * x0 + b[4] * x1 + b[5] * x2;
                a2s += b[6] * x0 + b[7] * x1 + b[8] * x2;
            }
        }
#pragma unroll
        for (int m = LPR / 2; m >= 1; m >>= 1) { a0s += __shfl_xor(a0s, m, 64); a1s += __shfl_xor(a1s, m, 64); a2s += __shfl_xor(a2s, m, 64); }
        if (row < r1 && k < 3) yout[3 * (int64_t)row + k] = k == 0 ? a0s : k == 1 ? a1s : a2s;
        __syncthreads();
    }
}

// K3: K2 software-pipelined -- while a trip is computed out of LDS, the next trip's values and x rows are in flight
// into registers and the trip after that has its columns loading; row offsets of the trip ride along into LDS.
template <int RPB, int CAP>
__global__ void __launch_bounds__(256)
k_step_pipe(const int *__restrict__ rowptr, const int *__restrict__ col, const float *__restrict__ blk,
            const float *__restrict__ xin, float *__restrict__ yout, int nb, int64_t nnzb)
{
    constexpr int LPR = 256 / RPB;
    constexpr int NQ = (CAP * 9 + 3) / 4 + 1, QPT = (NQ + 255) / 256, EPT = (CAP + 255) / 256;
    extern __shared__ float lds[];
    float *vL = lds, *xL = lds + NQ * 4;
    int *rpL = reinterpret_cast<int *>(xL + CAP * 3);
    const int trips = (nb + RPB - 1) / RPB, G = gridDim.x;
    struct Rg { int r0, r1, e0, e1; };
    auto range_of = [&](int trip) {
        Rg g{0, 0, 0, 0};
        if (trip < trips) { g.r0 = trip * RPB; g.r1 = g.r0 + RPB < nb ? g.r0 + RPB : nb; g.e0 = rowptr[g.r0]; g.e1 = rowptr[g.r1]; }
        return g;
    };
    auto load_cols = [&](const Rg &g, int (&c)[EPT]) {
#pragma unroll
        for (int j = 0; j < EPT; ++j) { const int i = threadIdx.x + 256 * j; c[j] = i < g.e1 - g.e0 ? col[g.e0 + i] : 0; }
    };
    float4 v[QPT]; float xr[EPT][3]; int rp = 0;
    auto issue = [&](const Rg &g, const int (&c)[EPT]) {
        const int64_t a0 = ((int64_t)g.e0 * 9) & ~(int64_t)3;
        const int nq = (int)(((int64_t)g.e1 * 9 - a0 + 3) >> 2);
#pragma unroll
        for (int j = 0; j < QPT; ++j) {
            const int q = threadIdx.x + 256 * j;
            const int64_t gq = a0 + 4 * (int64_t)q;
            if (q < nq) {
                if (gq + 4 <= nnzb * 9) { typedef float vf4 __attribute__((ext_vector_type(4))); const vf4 t = __builtin_nontemporal_load(reinterpret_cast<const vf4 *>(blk + gq)); v[j] = make_float4(t.x, t.y, t.z, t.w); }
                else { v[j].x = blk[gq]; v[j].y = gq + 1 < nnzb * 9 ? blk[gq + 1] : 0.f; v[j].z = gq + 2 < nnzb * 9 ? blk[gq + 2] : 0.f; v[j].w = 0.f; }
            }
        }
#pragma unroll
        for (int j = 0; j < EPT; ++j) {
            const float *xp = xin + (int64_t)c[j] * 3;
            xr[j][0] = xp[0]; xr[j][1] = xp[1]; xr[j][2] = xp[2];
        }
        if (threadIdx.x <= g.r1 - g.r0 && g.r1 > g.r0) rp = rowptr[g.r0 + threadIdx.x];
    };
    Rg g0 = range_of(blockIdx.x); int c0[EPT]; load_cols(g0, c0);
    issue(g0, c0);
    Rg g1 = range_of(blockIdx.x + G); int c1[EPT]; load_cols(g1, c1);
    for (int trip = blockIdx.x; trip < trips; trip += G) {
        const int ne = g0.e1 - g0.e0;
        const int64_t f0 = (int64_t)g0.e0 * 9, a0 = f0 & ~(int64_t)3;
        const int nq = (int)(((int64_t)g0.e1 * 9 - a0 + 3) >> 2);
#pragma unroll
        for (int j = 0; j < QPT; ++j) { const int q = threadIdx.x + 256 * j; if (q < nq) *reinterpret_cast<float4 *>(vL + 4 * q) = v[j]; }
#pragma unroll
        for (int j = 0; j < EPT; ++j) {
            const int i = threadIdx.x + 256 * j;
            if (i < ne) { xL[3 * i] = xr[j][0]; xL[3 * i + 1] = xr[j][1]; xL[3 * i + 2] = xr[j][2]; }
        }
        if (threadIdx.x <= g0.r1 - g0.r0) rpL[threadIdx.x] = rp - g0.e0;
        __syncthreads();
        issue(g1, c1);
        Rg g2 = range_of(trip + 2 * G); int c2[EPT]; load_cols(g2, c2);
        const int lr = threadIdx.x / LPR, row = g0.r0 + lr, k = threadIdx.x % LPR;
        float a0s = 0.f, a1s = 0.f, a2s = 0.f;
        if (row < g0.r1) {
            const int s0 = rpL[lr], s1 = rpL[lr + 1];
            const float *vb = vL + (int)(f0 - a0);
            for (int i = s0 + k; i < s1; i += LPR) {
                const float *b = vb + i * 9; const float *x = xL + 3 * i;
                const float x0 = x[0], x1 = x[1], x2 = x[2];
                a0s += b[0] * x0 + b[1] * x1 + b[2] * x2;
                a1s += b[3] * x0 + b[4] * x1 + b[5] * x2;
                a2s += b[6] * x0 + b[7] * x1 + b[8] * x2;
            }
        }
#pragma unroll
        for (int m = LPR / 2; m >= 1; m >>= 1) { a0s += __shfl_xor(a0s, m, 64); a1s += __shfl_xor(a1s, m, 64); a2s += __shfl_xor(a2s, m, 64); }
        if (row < g0.r1 && k < 3) yout[3 * (int64_t)row + k] = k == 0 ? a0s : k == 1 ? a1s : a2s;
        __syncthreads();
        g0 = g1; g1 = g2;
#pragma unroll
        for (int j = 0; j < EPT; ++j) c1[j] = c2[j];
    }
}

int main(int argc, char **argv)
{
    const int nb = argc > 1 ? atoi(argv[1]) : 2000000, band = argc > 2 ? atoi(argv[2]) : 1414;
    std::vector<int> rp(nb + 1);
    rp[0] = 0;
    for (int r = 0; r < nb; ++r) rp[r + 1] = rp[r] + 14 + hash32(r) % 11;
    const int64_t nnzb = rp[nb];
    int *rowptr, *col; float *blk, *x, *y0, *y1;
    hipMalloc(&rowptr, (nb + 1) * 4); hipMalloc(&col, nnzb * 4); hipMalloc(&blk, nnzb * 36);
    hipMalloc(&x, (size_t)nb * 16); hipMalloc(&y0, (size_t)nb * 12); hipMalloc(&y1, (size_t)nb * 12);
    hipMemcpy(rowptr, rp.data(), (nb + 1) * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_fill, dim3((nb + 255) / 256), dim3(256), 0, 0, rowptr, col, blk, nb, band);
    hipLaunchKernelGGL(k_fillx, dim3((4 * (size_t)nb + 255) / 256), dim3(256), 0, 0, x, 4 * (size_t)nb);
    hipDeviceSynchronize();
    const double bytes = 4.0 * (nb + 1) + 40.0 * nnzb + 24.0 * nb;
    printf("nb %d nnzb %lld bytes/step %.3f GB\n", nb, (long long)nnzb, bytes * 1e-9);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto time = [&](const char *name, auto launch, float *y) {
        float best = 1e30f, sum = 0.f;
        for (int rep = 0; rep < 12; ++rep) {
            hipEventRecord(e0); launch(y); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep >= 2) { best = std::min(best, ms); sum += ms; }
        }
        hipError_t er = hipGetLastError();
        printf("%-44s best %.3f ms  mean %.3f ms  %.0f GB/s (best)  %s\n", name, best, sum / 10, bytes / best * 1e-6, er == hipSuccess ? "" : hipGetErrorString(er));
    };
    auto diff = [&](const char *name) {
        std::vector<float> a((size_t)nb * 3), b((size_t)nb * 3);
        hipMemcpy(a.data(), y0, a.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(b.data(), y1, b.size() * 4, hipMemcpyDeviceToHost);
        double m = 0, s = 0; for (size_t i = 0; i < a.size(); ++i) { m = std::max(m, (double)std::fabs(a[i] - b[i])); s = std::max(s, (double)std::fabs(a[i])); }
        printf("   %s: max|diff| %.3g of max|y| %.3g\n", name, m, s);
    };
    const int cap_grid = 256 * 16;
    time("K0 lanes g=7 (escat.hip)", [&](float *y) { hipLaunchKernelGGL((k_step_lanes<7, true>), dim3(cap_grid), dim3(256), 0, 0, rowptr, col, blk, x, y, (int64_t)nb); }, y0);
    time("K1 lanes g=7, no dependent gather", [&](float *y) { hipLaunchKernelGGL((k_step_lanes<7, false>), dim3(cap_grid), dim3(256), 0, 0, rowptr, col, blk, x, y, (int64_t)nb); }, y1);
#define RUN_LDS(RPB, CAP, WGS, VAR)                                                                                         \
    {                                                                                                                  \
        constexpr int NQ = (CAP * 9 + 3) / 4 + 1;                                                                      \
        const size_t lb = (size_t)(NQ * 4 + CAP * 3 + RPB + 4) * 4;                                                    \
        auto kern = k_step_lds<RPB, CAP, VAR>;                                                                          \
        hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lb);                  \
        int occ = 0; hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kern, 256, lb);                                \
        const int wgs = WGS > 0 ? WGS : 256 * (occ + WGS);                                                             \
        char nm[96]; snprintf(nm, 96, "K2 staged RPB=%d CAP=%d var %d occ %d grid %d", RPB, CAP, VAR, occ, wgs);       \
        time(nm, [&](float *y) { hipLaunchKernelGGL(kern, dim3(wgs), dim3(256), lb, 0, rowptr, col, blk, x, y, nb, nnzb); }, y1); \
        diff(nm);                                                                                                      \
    }
#define RUN_PIPE(RPB, CAP, WGS)                                                                                        \
    {                                                                                                                  \
        constexpr int NQ = (CAP * 9 + 3) / 4 + 1;                                                                      \
        const size_t lb = (size_t)(NQ * 4 + CAP * 3 + RPB + 4) * 4;                                                    \
        auto kern = k_step_pipe<RPB, CAP>;                                                                             \
        hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lb);                  \
        int occ = 0; hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kern, 256, lb);                                \
        const int wgs = WGS > 0 ? WGS : 256 * (occ + WGS);                                                             \
        char nm[96]; snprintf(nm, 96, "K3 pipelined RPB=%d CAP=%d occ %d grid %d", RPB, CAP, occ, wgs);                \
        time(nm, [&](float *y) { hipLaunchKernelGGL(kern, dim3(wgs), dim3(256), lb, 0, rowptr, col, blk, x, y, nb, nnzb); }, y1); \
        diff(nm);                                                                                                      \
    }
    RUN_PIPE(16, 384, 0)
    RUN_PIPE(32, 768, 0)
    RUN_PIPE(32, 768, -1)
    RUN_LDS(32, 768, 0, 6)
    RUN_LDS(32, 768, -1, 6)
    RUN_LDS(32, 640, 0, 6)
    RUN_LDS(64, 1536, 0, 6)
    RUN_LDS(64, 1280, 0, 6)
    return 0;
}
