// LDS gather micro-benchmark: how many LDS cycles does one ds_read_b128 wave-instruction cost for a given
// lane -> 16-byte-slot pattern?  Used to check the bank-conflict model the SELL slot planner is built on.
//   build: hipcc -O3 --offload-arch=gfx950 -o lds_gather_bench lds_gather_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <string>
#include <algorithm>
#include <numeric>

#define ROWS 1024            // float4 rows in LDS (16 KiB)
#define ITERS 4096

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

// 4 patterns x 8 immediate offsets (same slot, different rows) per loop trip, no VALU work in between
#define RD8(INS, T, A)                                                                                     \
    { T v0, v1, v2, v3, v4, v5, v6, v7;                                                                    \
      asm volatile(INS " %0, %8\n" INS " %1, %8 offset:256\n" INS " %2, %8 offset:512\n"                   \
                   INS " %3, %8 offset:768\n" INS " %4, %8 offset:1024\n" INS " %5, %8 offset:1280\n"      \
                   INS " %6, %8 offset:1536\n" INS " %7, %8 offset:1792\n s_waitcnt lgkmcnt(0)\n"          \
                   : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3), "=&v"(v4), "=&v"(v5), "=&v"(v6), "=&v"(v7) : "v"(A)); }
#define WR8(INS, A, D)                                                                                     \
    asm volatile(INS " %0, %1\n" INS " %0, %1 offset:256\n" INS " %0, %1 offset:512\n"                     \
                 INS " %0, %1 offset:768\n" INS " %0, %1 offset:1024\n" INS " %0, %1 offset:1280\n"        \
                 INS " %0, %1 offset:1536\n" INS " %0, %1 offset:1792\n s_waitcnt lgkmcnt(0)\n" :: "v"(A), "v"(D) : "memory");

template <int W>  // W = bytes per lane: 16 (b128), 8 (b64), 4 (b32)
__global__ void __launch_bounds__(512) k_gather(const int *__restrict__ pat, float *out, int iters)
{
    __shared__ __align__(256) float4 x[ROWS + 128];
    for (int i = threadIdx.x; i < ROWS + 128; i += blockDim.x) x[i] = make_float4(i, 1.f, 2.f, 3.f);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    unsigned addr[4];
    for (int k = 0; k < 4; ++k) addr[k] = (unsigned)(size_t)(&x[0]) + 16u * (unsigned)(pat[k * 64 + lane] & (ROWS - 1));
    for (int it = 0; it < iters / 8; ++it) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (W == 16) RD8("ds_read_b128", f4, addr[k])
            else if (W == 8) RD8("ds_read_b64", f2, addr[k])
            else RD8("ds_read_b32", float, addr[k])
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x[threadIdx.x].x;
}

template <int W>
__global__ void __launch_bounds__(512) k_scatter(const int *__restrict__ pat, float *out, int iters)
{
    __shared__ __align__(256) float4 x[ROWS + 128];
    const int lane = threadIdx.x & 63;
    unsigned addr[4];
    for (int k = 0; k < 4; ++k) addr[k] = (unsigned)(size_t)(&x[0]) + 16u * (unsigned)(pat[k * 64 + lane] & (ROWS - 1));
    f4 d4 = {1.f, 2.f, 3.f, (float)lane}; float d1 = (float)lane;
    for (int it = 0; it < iters / 8; ++it) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (W == 16) WR8("ds_write_b128", addr[k], d4)
            else WR8("ds_write_b32", addr[k], d1)
        }
    }
    __syncthreads();
    out[blockIdx.x * blockDim.x + threadIdx.x] = x[threadIdx.x].x;
}

#define FCOLS 24
__global__ void __launch_bounds__(512) k_gather_file(const int *__restrict__ pat, float *out, int iters)
{
    __shared__ __align__(256) float4 x[2 * ROWS + 128];
    for (int i = threadIdx.x; i < 2 * ROWS + 128; i += blockDim.x) x[i] = make_float4(i, 1.f, 2.f, 3.f);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    unsigned addr[FCOLS];
#pragma unroll
    for (int k = 0; k < FCOLS; ++k) addr[k] = (unsigned)(size_t)(&x[0]) + 16u * (unsigned)(pat[k * 64 + lane] & (2 * ROWS - 1));
    for (int it = 0; it < iters / 8; ++it) {
#pragma unroll
        for (int k = 0; k < FCOLS; ++k) RD8("ds_read_b128", f4, addr[k])
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x[threadIdx.x].x;
}

static const int G0[16] = {0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27};
static int model_group(int lane) { int l = lane & 31; bool g0 = std::find(G0, G0 + 16, l) != G0 + 16; return (lane >> 5) * 2 + (g0 ? 0 : 1); }
static int model_rank(int lane) { int g = model_group(lane), r = 0; for (int q = 0; q < lane; ++q) r += model_group(q) == g; return r; }

int main(int argc, char **argv)
{
    int waves = argc > 1 ? atoi(argv[1]) : 8;
    struct Pat { std::string name; std::vector<int> rows; };
    std::vector<Pat> pats;
    auto mk = [&](const char *n, auto f) { Pat p; p.name = n; p.rows.resize(256); for (int k = 0; k < 4; ++k) for (int l = 0; l < 64; ++l) p.rows[k * 64 + l] = f(k, l); pats.push_back(p); };
    srand(1);
    mk("contiguous rows (row=lane)", [](int k, int l) { return l + 64 * k; });
    mk("all lanes one slot, distinct rows (16-way by any model)", [](int k, int l) { return 16 * l; });
    mk("all lanes one row (broadcast)", [](int k, int l) { return 5 + k; });
    mk("random rows", [](int k, int l) { return rand() % 500; });
    mk("model groups: slot = rank in {0-3,12-15,20-27}.. groups", [](int k, int l) { return model_rank(l) + 16 * (rand() % 31); });
    mk("contiguous 16-lane groups: slot = lane&15, random row", [](int k, int l) { return (l & 15) + 16 * (rand() % 31); });
    mk("8-lane groups: slot = (lane&7)*2 + random bit", [](int k, int l) { return ((l & 7) * 2 + (rand() & 1)) + 16 * (rand() % 31); });
    mk("8-lane groups distinct mod 8 slots (slot=(lane&7)+8*rand)", [](int k, int l) { return ((l & 7) + 8 * (rand() & 1)) + 16 * (rand() % 31); });
    mk("32-lane halves: slot = lane&15 twice (2 lanes/slot per half)", [](int k, int l) { return (l & 15) + 16 * ((l >> 4) + 4 * k); });
    mk("model groups, 2 per slot: slot = rank/2", [](int k, int l) { return model_rank(l) / 2 + 16 * (rand() % 31); });
    mk("contig groups, 2 per slot: slot = (lane&15)/2", [](int k, int l) { return (l & 15) / 2 + 16 * (rand() % 31); });
    int *dpat; float *dout;
    hipMalloc(&dpat, 256 * sizeof(int)); hipMalloc(&dout, 256 * 8 * 512 * sizeof(float));
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int ncu = prop.multiProcessorCount; const double ghz = prop.clockRate * 1e-6;
    printf("CUs %d clock %.2f GHz, %d waves per workgroup, one workgroup per CU\n", ncu, ghz, waves);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](const char *what, auto kern) {
        printf("-- %s: LDS cycles per wave-instruction (CU-level: time * clock / (iters*4*waves))\n", what);
        for (auto &p : pats) {
            hipMemcpy(dpat, p.rows.data(), 256 * sizeof(int), hipMemcpyHostToDevice);
            float best = 1e30f;
            for (int rep = 0; rep < 3; ++rep) {
                hipEventRecord(e0);
                hipLaunchKernelGGL(kern, dim3(ncu), dim3(64 * waves), 0, 0, dpat, dout, ITERS);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1); best = std::min(best, ms);
            }
            printf("   %6.2f  %s\n", best * 1e-3 * ghz * 1e9 / ((double)ITERS * 4 * waves), p.name.c_str());
        }
    };
    if (argc > 2) {      // file: int32 npat, then npat tables of FCOLS x 64 LDS rows; prints cycles per wave-instruction per table
        FILE *f = fopen(argv[2], "rb"); int np = 0;
        if (!f || fread(&np, 4, 1, f) != 1) { printf("cannot read %s\n", argv[2]); return 1; }
        int *dp; hipMalloc(&dp, FCOLS * 64 * sizeof(int));
        std::vector<int> t(FCOLS * 64); double sum = 0;
        for (int i = 0; i < np; ++i) {
            if (fread(t.data(), 4, FCOLS * 64, f) != FCOLS * 64) return 1;
            hipMemcpy(dp, t.data(), FCOLS * 64 * sizeof(int), hipMemcpyHostToDevice);
            float best = 1e30f;
            for (int rep = 0; rep < 3; ++rep) {
                hipEventRecord(e0);
                hipLaunchKernelGGL(k_gather_file, dim3(ncu), dim3(64 * waves), 0, 0, dp, dout, 1024);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1); best = std::min(best, ms);
            }
            const double cyc = best * 1e-3 * ghz * 1e9 / (1024.0 * FCOLS * waves);
            printf("   table %3d: %6.2f cycles per ds_read_b128\n", i, cyc); sum += cyc;
        }
        printf("mean %.2f over %d tables (%s)\n", sum / np, np, argv[2]);
        return 0;
    }
    run("ds_read_b128", k_gather<16>);
    run("ds_read_b64", k_gather<8>);
    run("ds_read_b32", k_gather<4>);
    run("ds_write_b128", k_scatter<16>);
    run("ds_write_b32", k_scatter<4>);
    return 0;
}
