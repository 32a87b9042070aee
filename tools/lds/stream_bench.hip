// HBM streaming micro-benchmark in the shape of the chain kernel's operator load: one persistent workgroup per CU,
// every wave pulls a contiguous run of "slots" (2304 B per slot and lane group: two 1-KiB dwordx4 pieces + one 256-B
// dword piece) into registers in one burst, then "computes" for a while.  Which variant reaches the HBM rate?
//   build: hipcc -O3 --offload-arch=gfx950 -o stream_bench stream_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <algorithm>

#define SLOTS 16
typedef float vf4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 nt4(const float *p) { const vf4 v = __builtin_nontemporal_load(reinterpret_cast<const vf4 *>(p)); return make_float4(v.x, v.y, v.z, v.w); }
template <int MODE, int NT>   // 0: ours (x4, x4, x1 per slot)  1: three x4 per slot (same bytes+) 2: ours with nt  3: ours, slots interleaved over 2 bursts
__global__ void __launch_bounds__(NT) k_stream(const float *__restrict__ src, float *out, int chunks_per_wg, size_t chunk_floats, int spin)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int NW = NT / 64;
    float acc = 0.f;
    for (int c = 0; c < chunks_per_wg; ++c) {
        const float *base = src + ((size_t)c * gridDim.x + blockIdx.x) * chunk_floats + (size_t)wave * SLOTS * 576;
        float4 p0[SLOTS], p1[SLOTS]; float p2[SLOTS];
#pragma unroll
        for (int d = 0; d < SLOTS; ++d) {
            const float *vb = base + d * 576;
            if (MODE == 2) {
                p0[d] = nt4(vb + lane * 4);
                p1[d] = nt4(vb + 256 + lane * 4);
                p2[d] = __builtin_nontemporal_load(vb + 512 + lane);
            } else {
                p0[d] = *reinterpret_cast<const float4 *>(vb + lane * 4);
                p1[d] = *reinterpret_cast<const float4 *>(vb + 256 + lane * 4);
                if (MODE == 1) p2[d] = reinterpret_cast<const float4 *>(vb + 512)[lane & 15].x; else p2[d] = vb[512 + lane];
            }
        }
#pragma unroll
        for (int d = 0; d < SLOTS; ++d) acc += p0[d].x + p0[d].y + p0[d].z + p0[d].w + p1[d].x + p1[d].y + p1[d].z + p1[d].w + p2[d];
        for (int s = 0; s < spin; ++s) { acc = fmaf(acc, 1.0001f, 0.5f); __builtin_amdgcn_s_sleep(8); }
        __syncthreads();
    }
    out[(size_t)blockIdx.x * NT + threadIdx.x] = acc;
}

int main(int argc, char **argv)
{
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int ncu = prop.multiProcessorCount;
    const int chunks = 64;
    float *src, *out;
    const size_t chunk16 = (size_t)16 * SLOTS * 576;           // floats per chunk at 16 waves
    hipMalloc(&src, (size_t)chunks * ncu * 2 * chunk16 * 4); hipMalloc(&out, (size_t)ncu * 2 * 1024 * 4);
    hipMemset(src, 0, (size_t)chunks * ncu * 2 * chunk16 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](const char *name, auto kern, int nt, int wg_per_cu, int spin) {
        const size_t cf = (size_t)(nt / 64) * SLOTS * 576;
        float best = 1e30f;
        for (int rep = 0; rep < 4; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(kern, dim3(ncu * wg_per_cu), dim3(nt), 0, 0, src, out, chunks, cf, spin);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); best = std::min(best, ms);
        }
        const double bytes = (double)chunks * ncu * wg_per_cu * cf * 4;
        printf("%-58s %7.1f us  %6.2f TB/s  %5.1f GB/s per CU   (%.0f KB per chunk)\n", name, best * 1e3, bytes / best * 1e-9, bytes / best * 1e-6 / ncu, cf * 4 / 1024.0);
    };
    run("ours: 512 thr, 1 WG/CU, x4+x4+x1 per slot", k_stream<0, 512>, 512, 1, 0);
    run("three x4 per slot", k_stream<1, 512>, 512, 1, 0);
    run("ours, nt loads", k_stream<2, 512>, 512, 1, 0);
    run("ours, 1024 thr (16 waves)", k_stream<0, 1024>, 1024, 1, 0);
    run("ours, 256 thr x 2 WG/CU", k_stream<0, 256>, 256, 2, 0);
    run("ours, 256 thr x 4 WG/CU", k_stream<0, 256>, 256, 4, 0);
    run("ours, 512 thr, compute pause ~ (spin 200)", k_stream<0, 512>, 512, 1, 200);
    run("nt, 1024 thr", k_stream<2, 1024>, 1024, 1, 0);
    return 0;
}
