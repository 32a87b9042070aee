// Per-graph divergences of the diffusion powers for the infogain scale selection
// (/root/reference/models/infogain.py:810-900 get_divergence_by_channel, with normalize_channels :780-807, get_floor
// :760-777 and models/nn_utilities.py:76-151 minmax_scale_tensor, :154-199 norm_tensor_to_prob_mass, :202-245
// get_mid_btw_min_and_2nd_low_vector_vals).
//
// The reference loops in Python over graphs x channels x steps; each (graph, channel, step) column of the powers is
//   1. shifted and scaled onto [floor', 1]:  v' = (v - (min - floor)) / (max - (min - floor)), where floor is either a
//      given constant or the midpoint of the lowest and the second-lowest DISTINCT value of the column (sign flipped
//      if negative),
//   2. divided by its sum (a probability vector),
//   3. compared with the same channel's last step T:  KL(q_T || p_t) = sum q (log q - log(p + 1e-12)),  or  L1.
//
// Here the powers arrive as the chain kernel wrote them, W[row][channel][step] (one launch of the all-powers plan, no
// (T+1, N, C) stack), and one workgroup owns one graph: its threads are (row group, column) pairs over the C*F
// contiguous floats of a row, so every pass over the graph's rows is a contiguous stream; the three passes (extrema,
// sum, divergence) re-read the graph from L2 (a 500-node graph with C=3, T=16 is 102 KB).  HBM-bound: one read of W.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include "../../include/escat.h"
#include "escat_internal.h"

#define IG_NT 256

__global__ void __launch_bounds__(IG_NT)
k_infogain_div(const float *__restrict__ W, const int *__restrict__ graph_ptr, int C, int F, int cpw, int start,
               int metric, int fixed_floor, float floor_val, float *__restrict__ out, int *__restrict__ status)
{
    __shared__ float s_a[IG_NT], s_b[IG_NT], s_c[IG_NT];        // partials per (row group, column)
    __shared__ float s_lo[IG_NT], s_rng[IG_NT], s_sum[IG_NT];   // per column: shifted minimum, range, sum of v'
    __shared__ int s_bad[IG_NT];
    const int b = blockIdx.x, c0 = blockIdx.y * cpw;
    const int nc = min(cpw, C - c0), Kw = nc * F, G = IG_NT / Kw;
    const int tid = threadIdx.x, k = tid % Kw, g = tid / Kw;
    const bool active = g < G;
    const int r0 = graph_ptr[b], rows = graph_ptr[b + 1] - r0;
    const int64_t K = (int64_t)C * F;
    const float *base = W + (int64_t)r0 * K + (int64_t)c0 * F;

    // pass 1: minimum, second-lowest distinct value, maximum
    float mn = INFINITY, sec = INFINITY, mx = -INFINITY;
    if (active)
        for (int r = g; r < rows; r += G) {
            const float v = base[r * K + k];
            if (v < mn) { sec = mn; mn = v; }
            else if (v > mn && v < sec) sec = v;
            mx = fmaxf(mx, v);
        }
    s_a[tid] = mn; s_b[tid] = sec; s_c[tid] = mx;
    __syncthreads();
    if (tid < Kw) {
        float m = s_a[tid], s = s_b[tid], x = s_c[tid];
        for (int q = 1; q < G; ++q) {
            const float m2 = s_a[q * Kw + tid], s2 = s_b[q * Kw + tid];
            const float between = (m2 != m) ? fmaxf(m, m2) : INFINITY;
            s = fminf(fminf(s, s2), between);
            m = fminf(m, m2);
            x = fmaxf(x, s_c[q * Kw + tid]);
        }
        int bad = rows <= 0;
        float fl = floor_val;
        if (!fixed_floor) {
            if (x - m < 1e-10f) bad = 1;                 // no variance: the reference's floor search returns 'error'
            fl = (m + s) / 2.f;
        }
        if (fl <= 0.f) fl = -fl;
        if (!(fl > 0.f)) bad = 1;                        // a zero floor is rejected by minmax_scale_tensor
        const float lo = m - fl, rng = x - lo;
        if (!(fabsf(rng) >= 1e-10f)) bad = 1;
        s_lo[tid] = lo; s_rng[tid] = rng; s_bad[tid] = bad;
    }
    __syncthreads();

    // pass 2: sum of the rescaled column
    const float lo = s_lo[k], rng = s_rng[k];
    float acc = 0.f;
    if (active)
        for (int r = g; r < rows; r += G) acc += (base[r * K + k] - lo) / rng;
    s_a[tid] = acc;
    __syncthreads();
    if (tid < Kw) {
        float s = s_a[tid];
        for (int q = 1; q < G; ++q) s += s_a[q * Kw + tid];
        s_sum[tid] = s;
    }
    __syncthreads();

    // pass 3: divergence of step t against the channel's last step
    const int t = k % F, kT = k - t + (F - 1);
    const bool wanted = active && t >= start && t < F - 1;
    acc = 0.f;
    if (wanted) {
        const float sum = s_sum[k], loT = s_lo[kT], rngT = s_rng[kT], sumT = s_sum[kT];
        for (int r = g; r < rows; r += G) {
            const float p = ((base[r * K + k] - lo) / rng) / sum;
            const float q = ((base[r * K + kT] - loT) / rngT) / sumT;
            if (metric == 0) acc += q > 0.f ? q * (logf(q) - logf(p + 1e-12f)) : 0.f;
            else acc += fabsf(p - q);
        }
    }
    s_b[tid] = acc;
    __syncthreads();
    if (tid < Kw && t >= start && t < F - 1) {
        float s = s_b[tid];
        for (int q = 1; q < G; ++q) s += s_b[q * Kw + tid];
        const int bad = s_bad[tid] | s_bad[kT];
        if (bad) atomicOr(status + b, 1);
        out[((int64_t)b * (F - 1 - start) + (t - start)) * C + c0 + k / F] = bad ? NAN : s;
    }
}

extern "C" int escat_infogain_divergence_f32(const float *W, const int32_t *graph_ptr, int64_t n_graphs, int C, int F,
                                             int start, int metric, int fixed_floor, float floor_val, float *out,
                                             int32_t *status, void *stream_)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (n_graphs < 0 || n_graphs > 0x7fffffff) return escat_fail(ESCAT_ERANGE, "n_graphs=%lld out of range", (long long)n_graphs);
    if (C < 1 || C > 65535) return escat_fail(ESCAT_ERANGE, "C=%d outside [1,65535]", C);
    if (F < 2 || F > IG_NT) return escat_fail(ESCAT_ERANGE, "F=%d outside [2,%d] (T+1 diffusion steps)", F, IG_NT);
    if (start < 0 || start >= F - 1) return escat_fail(ESCAT_EINVAL, "start=%d leaves no step in [start, %d)", start, F - 1);
    if (metric != 0 && metric != 1) return escat_fail(ESCAT_EINVAL, "metric must be 0 (kl) or 1 (l1)");
    if (n_graphs == 0) return ESCAT_OK;
    if (!W || !graph_ptr || !out || !status) return escat_fail(ESCAT_EINVAL, "null pointer");
    if (hipMemsetAsync(status, 0, (size_t)n_graphs * sizeof(int32_t), stream) != hipSuccess)
        return escat_fail(ESCAT_ELAUNCH, "hipMemsetAsync(status) failed");
    int cpw = IG_NT / F; if (cpw > C) cpw = C; if (cpw < 1) cpw = 1;
    dim3 grid((unsigned)n_graphs, (unsigned)((C + cpw - 1) / cpw));
    hipLaunchKernelGGL(k_infogain_div, grid, dim3(IG_NT), 0, stream, W, graph_ptr, C, F, cpw, start, metric, fixed_floor,
                       floor_val, out, status);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return escat_fail(ESCAT_ELAUNCH, "k_infogain_div: %s", hipGetErrorString(e));
    return ESCAT_OK;
}
