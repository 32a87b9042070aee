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

// partials of one column live at s[q * Kw + col], q < G.  Reads are issued eight at a time so that the one wave doing
// the reduction pays the LDS latency once per batch, not once per row group.
__device__ __forceinline__ float col_sum(const float *s, int col, int Kw, int G)
{
    float acc = s[col];
    for (int q0 = 1; q0 < G; q0 += 8) {
        float a[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] = s[min(q0 + j, G - 1) * Kw + col];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc += (q0 + j < G) ? a[j] : 0.f;
    }
    return acc;
}

// merges (minimum, second-lowest distinct value, maximum) triples; merging a triple twice changes nothing
__device__ __forceinline__ void col_extrema(const float *sa, const float *sb, const float *sc, int col, int Kw, int G,
                                            float &m, float &s, float &x)
{
    m = sa[col]; s = sb[col]; x = sc[col];
    for (int q0 = 1; q0 < G; q0 += 8) {
        float a[8], b[8], c[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int i = min(q0 + j, G - 1) * Kw + col;
            a[j] = sa[i]; b[j] = sb[i]; c[j] = sc[i];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float between = (a[j] != m) ? fmaxf(m, a[j]) : INFINITY;
            s = fminf(fminf(s, b[j]), between);
            m = fminf(m, a[j]);
            x = fmaxf(x, c[j]);
        }
    }
}

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
    auto see = [&](float v) {
        if (v < mn) { sec = mn; mn = v; }
        else if (v > mn && v < sec) sec = v;
        mx = fmaxf(mx, v);
    };
    if (active) {
        int r = g;
        for (; r + 3 * G < rows; r += 4 * G) {            // four independent loads in flight per thread
            const float v0 = base[r * K + k], v1 = base[(r + G) * K + k], v2 = base[(r + 2 * G) * K + k], v3 = base[(r + 3 * G) * K + k];
            see(v0); see(v1); see(v2); see(v3);
        }
        for (; r < rows; r += G) see(base[r * K + k]);
    }
    s_a[tid] = mn; s_b[tid] = sec; s_c[tid] = mx;
    __syncthreads();
    if (tid < Kw) {
        float m, s, x;
        col_extrema(s_a, s_b, s_c, tid, Kw, G, m, s, x);
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
    const float lo = s_lo[k], rng = s_rng[k], irng = 1.f / rng;
    float acc = 0.f;
    if (active) {
        int r = g;
        for (; r + 3 * G < rows; r += 4 * G) {
            const float v0 = base[r * K + k], v1 = base[(r + G) * K + k], v2 = base[(r + 2 * G) * K + k], v3 = base[(r + 3 * G) * K + k];
            acc += (v0 - lo) * irng; acc += (v1 - lo) * irng; acc += (v2 - lo) * irng; acc += (v3 - lo) * irng;
        }
        for (; r < rows; r += G) acc += (base[r * K + k] - lo) * irng;
    }
    s_a[tid] = acc;
    __syncthreads();
    if (tid < Kw) {
        s_sum[tid] = col_sum(s_a, tid, Kw, G);
    }
    __syncthreads();

    // pass 3: divergence of step t against the channel's last step
    const int t = k % F, kT = k - t + (F - 1);
    const bool wanted = active && t >= start && t < F - 1;
    acc = 0.f;
    if (wanted) {
        // one reciprocal per column instead of two divisions per element (<= 2 ulp from the reference's sequence), and
        // the hardware log2: its arguments are probabilities, far from 1, where its absolute error would show
        const float loT = s_lo[kT], ip = irng / s_sum[k], iq = (1.f / s_rng[kT]) / s_sum[kT];
        auto term = [&](float v, float vT) {
            const float p = (v - lo) * ip, q = (vT - loT) * iq;
            if (metric == 0) acc += q > 0.f ? q * (__log2f(q) - __log2f(p + 1e-12f)) : 0.f;
            else acc += fabsf(p - q);
        };
        int r = g;
        for (; r + 3 * G < rows; r += 4 * G) {
            const float v0 = base[r * K + k], v1 = base[(r + G) * K + k], v2 = base[(r + 2 * G) * K + k], v3 = base[(r + 3 * G) * K + k];
            const float t0 = base[r * K + kT], t1 = base[(r + G) * K + kT], t2 = base[(r + 2 * G) * K + kT], t3 = base[(r + 3 * G) * K + kT];
            term(v0, t0); term(v1, t1); term(v2, t2); term(v3, t3);
        }
        for (; r < rows; r += G) term(base[r * K + k], base[r * K + kT]);
    }
    s_b[tid] = acc;
    __syncthreads();
    if (tid < Kw && t >= start && t < F - 1) {
        float s = col_sum(s_b, tid, Kw, G);
        if (metric == 0) s *= 0.69314718055994531f;          // log2 -> natural log
        const int bad = s_bad[tid] | s_bad[kT];
        if (bad) atomicOr(status + b, 1);
        out[((int64_t)b * (F - 1 - start) + (t - start)) * C + c0 + k / F] = bad ? NAN : s;
    }
}


// Register-resident form for graphs of up to RG_MAXI * G rows (G = row groups of the workgroup): every thread keeps its
// <= RG_MAXI values of one column in registers, so the graph is read from HBM exactly once; the last step's
// probabilities q and log2 q go through LDS ([rows][channels of the workgroup]) for the threads of the other steps.
#ifndef RG_NT
#define RG_NT 1024
#define RG_MAXI 32
#define RG_WAVES 8
#endif

__global__ void __launch_bounds__(RG_NT, RG_WAVES)
k_infogain_div_reg(const float *__restrict__ W, const int *__restrict__ graph_ptr, int C, int F, int cpw, int start,
                   int metric, int fixed_floor, float floor_val, float *__restrict__ out, int *__restrict__ status)
{
    extern __shared__ float s_q[];                               // [rows][nc] q, then [rows][nc] log2 q
    __shared__ float s_a[RG_NT], s_b[RG_NT], s_c[RG_NT];
    __shared__ float s_lo[RG_NT], s_rng[RG_NT], s_sum[RG_NT];       // Kw can be anything up to RG_NT here
    __shared__ int s_bad[RG_NT];
    const int b = blockIdx.x, c0 = blockIdx.y * cpw;
    const int nc = min(cpw, C - c0), Kw = nc * F, G = RG_NT / Kw;
    const int tid = threadIdx.x, k = tid % Kw, g = tid / Kw;
    const int r0 = graph_ptr[b], rows = graph_ptr[b + 1] - r0;
    const int t = k % F, cl = k / F, kT = k - t + (F - 1);
    if (rows <= 0) {                                             // uniform: an empty graph has nothing to normalise
        if (tid < Kw && t >= start && t < F - 1) {
            out[((int64_t)b * (F - 1 - start) + (t - start)) * C + c0 + cl] = NAN;
            atomicOr(status + b, 1);
        }
        return;
    }
    const int64_t K = (int64_t)C * F;
    const float *base = W + (int64_t)r0 * K + (int64_t)c0 * F;
    // rows g, g+G, ... of this thread: every thread with g < G has `full` of them, some one more.  (The few threads with
    // g >= G compute on clamped copies; nothing they produce is read.)
    const int full = rows / G;                                   // uniform
    const bool extra = g < G && g + full * G < rows;

    // row i of the thread is at a uniform distance i * G * K from its first one: one vector offset, scalar strides
    // (32 clamped 64-bit addresses would double the register count); only the odd last row needs a clamp
    float v[RG_MAXI];
    const float *mine = base + (int64_t)(g < G ? g : 0) * K + k;      // g >= G: spare threads, kept inside the graph
    const int64_t GK = (int64_t)G * K;
    const float *last = base + (int64_t)min(g + full * G, rows - 1) * K + k;
#pragma unroll
    for (int i = 0; i < RG_MAXI; ++i) {
        if (i < full) v[i] = mine[i * GK];
        else if (i == full) v[i] = *last;
        else v[i] = 0.f;
    }

    float mn = INFINITY, sec = INFINITY, mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < RG_MAXI; ++i)
        if (i < full || (i == full && extra)) {                  // i < full is a scalar branch
            const float x = v[i], hi = fmaxf(mn, x);
            mn = fminf(mn, x);
            if (hi != mn) sec = fminf(sec, hi);                  // hi: the larger of (old minimum, x), a candidate unless equal
            mx = fmaxf(mx, x);
        }
    s_a[tid] = mn; s_b[tid] = sec; s_c[tid] = mx;
    __syncthreads();
    if (tid < Kw) {
        float m, s, x;
        col_extrema(s_a, s_b, s_c, tid, Kw, G, m, s, x);
        int bad = 0;
        float fl = floor_val;
        if (!fixed_floor) {
            if (x - m < 1e-10f) bad = 1;
            fl = (m + s) / 2.f;
        }
        if (fl <= 0.f) fl = -fl;
        if (!(fl > 0.f)) bad = 1;
        const float lo = m - fl, rng = x - lo;
        if (!(fabsf(rng) >= 1e-10f)) bad = 1;
        s_lo[tid] = lo; s_rng[tid] = rng; s_bad[tid] = bad;
    }
    __syncthreads();

    const float lo = s_lo[k], irng = 1.f / s_rng[k];
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < RG_MAXI; ++i)
        if (i < full || (i == full && extra)) acc += (v[i] - lo) * irng;
    s_a[tid] = acc;
    __syncthreads();
    if (tid < Kw) s_sum[tid] = col_sum(s_a, tid, Kw, G);
    __syncthreads();

    const float ip = irng / s_sum[k];
    float *s_lq = s_q + rows * nc;
    if (t == F - 1 && g < G) {
#pragma unroll
        for (int i = 0; i < RG_MAXI; ++i)
            if (i < full || (i == full && extra)) {
                const float q = (v[i] - lo) * ip;
                s_q[(g + i * G) * nc + cl] = q;
                s_lq[(g + i * G) * nc + cl] = q > 0.f ? __log2f(q) : 0.f;   // q == 0: the term is q * (finite) = 0
            }
    }
    __syncthreads();
    acc = 0.f;
    if (t >= start && t < F - 1 && g < G) {
#pragma unroll
        for (int i = 0; i < RG_MAXI; ++i)
            if (i < full || (i == full && extra)) {
                const float p = (v[i] - lo) * ip, q = s_q[(g + i * G) * nc + cl];
                if (metric == 0) acc += q * (s_lq[(g + i * G) * nc + cl] - __log2f(p + 1e-12f));
                else acc += fabsf(p - q);
            }
    }
    s_b[tid] = acc;
    __syncthreads();
    if (tid < Kw && t >= start && t < F - 1) {
        float s = col_sum(s_b, tid, Kw, G);
        if (metric == 0) s *= 0.69314718055994531f;
        const int bad = s_bad[tid] | s_bad[kT];
        if (bad) atomicOr(status + b, 1);
        out[((int64_t)b * (F - 1 - start) + (t - start)) * C + c0 + cl] = bad ? NAN : s;
    }
}

extern "C" int escat_infogain_divergence_f32(const float *W, const int32_t *graph_ptr, int64_t n_graphs, int C, int F,
                                             int start, int metric, int fixed_floor, float floor_val,
                                             int max_graph_rows, float *out, int32_t *status, void *stream_)
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
    if (max_graph_rows > 0) {
        // register-resident form: the most channels per workgroup for which every graph fits RG_MAXI rows per thread
        for (int nc = C < RG_NT / F ? C : RG_NT / F; nc >= 1; --nc) {
            const int G = RG_NT / (nc * F);
            const size_t lds = (size_t)max_graph_rows * nc * 2 * sizeof(float);
            if ((int64_t)G * RG_MAXI < max_graph_rows || lds > 96 * 1024) continue;
            dim3 grid((unsigned)n_graphs, (unsigned)((C + nc - 1) / nc));
            hipLaunchKernelGGL(k_infogain_div_reg, grid, dim3(RG_NT), lds, stream, W, graph_ptr, C, F, nc, start, metric,
                               fixed_floor, floor_val, out, status);
            hipError_t e = hipGetLastError();
            if (e != hipSuccess) return escat_fail(ESCAT_ELAUNCH, "k_infogain_div_reg: %s", hipGetErrorString(e));
            return ESCAT_OK;
        }
    }
    int cpw = IG_NT / F; if (cpw > C) cpw = C; if (cpw < 1) cpw = 1;
    dim3 grid((unsigned)n_graphs, (unsigned)((C + cpw - 1) / cpw));
    hipLaunchKernelGGL(k_infogain_div, grid, dim3(IG_NT), 0, stream, W, graph_ptr, C, F, cpw, start, metric, fixed_floor,
                       floor_val, out, status);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return escat_fail(ESCAT_ELAUNCH, "k_infogain_div: %s", hipGetErrorString(e));
    return ESCAT_OK;
}
