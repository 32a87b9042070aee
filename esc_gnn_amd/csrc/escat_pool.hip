// Neighbour pooling of per-edge values for ESCGNNModular._get_vector_invariants
// (/root/reference/models/escgnn_modular.py:431-502): per destination node and per column, mean / max / min / variance
// and lower quantiles (k-th smallest with k = floor((deg-1)*p), :484-497) of the values on the node's incoming edges.
// The reference computes the quantiles with a Python loop over nodes (one torch.kthvalue per node and level).
//
// Input: vals (E, W) with the edges grouped by destination (seg_ptr[n] .. seg_ptr[n+1]).  One thread per (node, column):
// it copies its <= 64 values into LDS, insertion-sorts them there together with their positions, and reads every
// statistic off the sorted list.  Columns are the fast index, so loads and stores are contiguous per node.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/escat.h"
#include "escat_internal.h"

#define POOL_NT 256
#define POOL_MAX_DEG 64

// stats bits
#define ST_MEAN 1
#define ST_MAX 2
#define ST_MIN 4
#define ST_VAR 8

__global__ void __launch_bounds__(POOL_NT)
k_pool_fwd(const float *__restrict__ vals, const int *__restrict__ seg_ptr, int64_t n_nodes, int W, int stats, int n_levels,
           const int *__restrict__ k_table, int table_ld, float *__restrict__ out, unsigned char *__restrict__ arg)
{
    extern __shared__ float lds[];                 // [POOL_MAX_DEG][POOL_NT] values, then the same of positions (bytes)
    float *sv = lds;
    unsigned char *sp = reinterpret_cast<unsigned char *>(lds + POOL_MAX_DEG * POOL_NT);
    const int tid = threadIdx.x;
    const int m = ((stats & ST_MEAN) ? 1 : 0) + ((stats & ST_MAX) ? 1 : 0) + ((stats & ST_MIN) ? 1 : 0) + ((stats & ST_VAR) ? 1 : 0) + n_levels;
    const int na = ((stats & ST_MAX) ? 1 : 0) + ((stats & ST_MIN) ? 1 : 0) + n_levels;
    const int64_t total = n_nodes * W;
    for (int64_t g = (int64_t)blockIdx.x * POOL_NT + tid; g < total; g += (int64_t)gridDim.x * POOL_NT) {
        const int64_t n = g / W; const int w = (int)(g - n * W);
        const int e0 = seg_ptr[n], cnt = seg_ptr[n + 1] - e0;
        float sum = 0.f, sq = 0.f;
        for (int i = 0; i < cnt; ++i) {             // insertion sort, ascending; equal values keep edge order
            const float x = vals[(int64_t)(e0 + i) * W + w];
            sum += x; sq += x * x;
            int j = i;
            while (j > 0 && sv[(j - 1) * POOL_NT + tid] > x) {
                sv[j * POOL_NT + tid] = sv[(j - 1) * POOL_NT + tid]; sp[j * POOL_NT + tid] = sp[(j - 1) * POOL_NT + tid]; --j;
            }
            sv[j * POOL_NT + tid] = x; sp[j * POOL_NT + tid] = (unsigned char)i;
        }
        float *o = out + n * (int64_t)m * W + w;
        unsigned char *a = arg ? arg + n * (int64_t)na * W + w : nullptr;
        int s = 0, sa = 0;
        const float mean = cnt > 0 ? sum / (float)cnt : 0.f;
        if (stats & ST_MEAN) { o[(int64_t)s * W] = mean; ++s; }
        if (stats & ST_MAX) {                        // first of equal maxima in edge order
            float v = 0.f; int p = 0;
            if (cnt > 0) { int q = cnt - 1; v = sv[q * POOL_NT + tid]; while (q > 0 && sv[(q - 1) * POOL_NT + tid] == v) --q; p = sp[q * POOL_NT + tid]; }
            o[(int64_t)s * W] = v; ++s; if (a) { a[(int64_t)sa * W] = (unsigned char)p; } ++sa;
        }
        if (stats & ST_MIN) {
            float v = 0.f; int p = 0;
            if (cnt > 0) { v = sv[tid]; p = sp[tid]; }
            o[(int64_t)s * W] = v; ++s; if (a) { a[(int64_t)sa * W] = (unsigned char)p; } ++sa;
        }
        if (stats & ST_VAR) {
            float v = 0.f;
            if (cnt > 0) { v = sq / (float)cnt - mean * mean; v = v > 0.f ? v : 0.f; }
            o[(int64_t)s * W] = v; ++s;
        }
        for (int l = 0; l < n_levels; ++l) {
            float v = 0.f; int p = 0;
            if (cnt > 0) { const int k = k_table[l * table_ld + cnt]; v = sv[k * POOL_NT + tid]; p = sp[k * POOL_NT + tid]; }
            o[(int64_t)s * W] = v; ++s; if (a) { a[(int64_t)sa * W] = (unsigned char)p; } ++sa;
        }
    }
}

// gradient w.r.t. the per-edge values: every (node, column) owns its segment's column, so no atomics
__global__ void __launch_bounds__(POOL_NT)
k_pool_bwd(const float *__restrict__ vals, const int *__restrict__ seg_ptr, int64_t n_nodes, int W, int stats, int n_levels,
           const float *__restrict__ gout, const unsigned char *__restrict__ arg, float *__restrict__ gvals)
{
    const int m = ((stats & ST_MEAN) ? 1 : 0) + ((stats & ST_MAX) ? 1 : 0) + ((stats & ST_MIN) ? 1 : 0) + ((stats & ST_VAR) ? 1 : 0) + n_levels;
    const int na = ((stats & ST_MAX) ? 1 : 0) + ((stats & ST_MIN) ? 1 : 0) + n_levels;
    const int64_t total = n_nodes * W;
    for (int64_t g = (int64_t)blockIdx.x * POOL_NT + threadIdx.x; g < total; g += (int64_t)gridDim.x * POOL_NT) {
        const int64_t n = g / W; const int w = (int)(g - n * W);
        const int e0 = seg_ptr[n], cnt = seg_ptr[n + 1] - e0;
        if (cnt <= 0) continue;
        const float *go = gout + n * (int64_t)m * W + w;
        const unsigned char *a = arg + n * (int64_t)na * W + w;
        int s = 0;
        float gmean = 0.f, gvar = 0.f;
        if (stats & ST_MEAN) { gmean = go[(int64_t)s * W]; ++s; }
        const int s_max = s; if (stats & ST_MAX) ++s;
        const int s_min = s; if (stats & ST_MIN) ++s;
        float mean = 0.f;
        if (stats & ST_VAR) {
            gvar = go[(int64_t)s * W]; ++s;
            float sum = 0.f, sq = 0.f;
            for (int i = 0; i < cnt; ++i) { const float x = vals[(int64_t)(e0 + i) * W + w]; sum += x; sq += x * x; }
            mean = sum / (float)cnt;
            if (sq / (float)cnt - mean * mean <= 0.f) gvar = 0.f;           // clamp(min=0) passes no gradient where it binds
        }
        const int s_q = s;
        const float inv = 1.f / (float)cnt;
        for (int i = 0; i < cnt; ++i) {
            float gacc = gmean * inv;
            if (stats & ST_VAR) gacc += gvar * 2.f * (vals[(int64_t)(e0 + i) * W + w] - mean) * inv;
            int sa = 0;
            if (stats & ST_MAX) { if (a[(int64_t)sa * W] == i) gacc += go[(int64_t)s_max * W]; ++sa; }
            if (stats & ST_MIN) { if (a[(int64_t)sa * W] == i) gacc += go[(int64_t)s_min * W]; ++sa; }
            for (int l = 0; l < n_levels; ++l, ++sa) if (a[(int64_t)sa * W] == i) gacc += go[(int64_t)(s_q + l) * W];
            gvals[(int64_t)(e0 + i) * W + w] = gacc;
        }
    }
}

// ---- fused form: cosines computed on the fly from the coefficients, nothing of size (E, W) is materialised.
// v (N, W, D): per node and wavelet a D-vector; in-edges of node n: src[seg_ptr[n] .. seg_ptr[n+1]).
// out (N, (1+m)*W) = [norm | mean | max | min | var | levels] (the reference's torch.cat, :504).
// 1 / max(sqrt(n2), 1e-8) with the hardware reciprocal square root (1 ulp; the clamp of the reference's norms,
// models/escgnn_modular.py:447-452, moves onto the squared norm).  One instruction instead of sqrt + IEEE division.
__device__ __forceinline__ float inv_norm(float n2) { return __builtin_amdgcn_rsqf(fmaxf(n2, 1e-16f)); }

template <int D>
__device__ __forceinline__ float cos_of(const float (&a)[D], float na, const float *__restrict__ b)
{
    float dot = 0.f, nb = 0.f;
#pragma unroll
    for (int c = 0; c < D; ++c) { const float x = b[c]; dot = fmaf(a[c], x, dot); nb = fmaf(x, x, nb); }
    return dot * inv_norm(nb) / na;
}

// gathers the cosines of a segment into this thread's LDS column, four edges in flight at a time (the loads of one edge
// depend on nothing but its index; issued one per loop trip they would cost a full L2/HBM latency each)
template <int D>
__device__ __forceinline__ void gather_cosines(const float *__restrict__ v, const int *__restrict__ src, int e0, int cnt, int W, int w,
                                               const float (&a)[D], float nac, float *sv, int tid, float &sum, float &sq)
{
    const float inac = 1.f / nac;
    for (int i0 = 0; i0 < cnt; i0 += 4) {
        float b[4][D];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + u < cnt ? i0 + u : cnt - 1;          // clamped: the extra lanes re-read the last edge
            const float *pb = v + ((int64_t)src[e0 + i] * W + w) * D;
#pragma unroll
            for (int c = 0; c < D; ++c) b[u][c] = pb[c];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (i0 + u < cnt) {
                float dot = 0.f, nb = 0.f;
#pragma unroll
                for (int c = 0; c < D; ++c) { dot = fmaf(a[c], b[u][c], dot); nb = fmaf(b[u][c], b[u][c], nb); }
                const float x = dot * inac * inv_norm(nb);
                sv[(i0 + u) * POOL_NT + tid] = x; sum += x; sq += x * x;
            }
        }
    }
}

template <int D>
__global__ void __launch_bounds__(POOL_NT)
k_vinv_fwd(const float *__restrict__ v, const int *__restrict__ src, const int *__restrict__ seg_ptr, int64_t n_nodes, int W,
           int stats, int n_levels, const int *__restrict__ k_table, int table_ld, float *__restrict__ out, unsigned char *__restrict__ arg,
           int cap)
{
    extern __shared__ float lds[];                 // [cap][POOL_NT] values, then as many position bytes
    float *sv = lds;
    unsigned char *sp = reinterpret_cast<unsigned char *>(lds + cap * POOL_NT);
    const int tid = threadIdx.x;
    const int m = ((stats & ST_MEAN) ? 1 : 0) + ((stats & ST_MAX) ? 1 : 0) + ((stats & ST_MIN) ? 1 : 0) + ((stats & ST_VAR) ? 1 : 0) + n_levels;
    const int na = ((stats & ST_MAX) ? 1 : 0) + ((stats & ST_MIN) ? 1 : 0) + n_levels;
    const int64_t total = n_nodes * W;
    for (int64_t g = (int64_t)blockIdx.x * POOL_NT + tid; g < total; g += (int64_t)gridDim.x * POOL_NT) {
        const int64_t n = g / W; const int w = (int)(g - n * W);
        float a[D]; float nrm = 0.f;
#pragma unroll
        for (int c = 0; c < D; ++c) { a[c] = v[g * D + c]; nrm = fmaf(a[c], a[c], nrm); }
        nrm = sqrtf(nrm);
        const float nac = nrm > 1e-8f ? nrm : 1e-8f;
        const int e0 = seg_ptr[n], cnt = seg_ptr[n + 1] - e0;
        float sum = 0.f, sq = 0.f;
        gather_cosines<D>(v, src, e0, cnt, W, w, a, nac, sv, tid, sum, sq);
        for (int i = 0; i < cnt; ++i) {             // insertion sort in place, ascending; equal values keep edge order
            const float x = sv[i * POOL_NT + tid];
            int j = i;
            while (j > 0 && sv[(j - 1) * POOL_NT + tid] > x) {
                sv[j * POOL_NT + tid] = sv[(j - 1) * POOL_NT + tid]; sp[j * POOL_NT + tid] = sp[(j - 1) * POOL_NT + tid]; --j;
            }
            sv[j * POOL_NT + tid] = x; sp[j * POOL_NT + tid] = (unsigned char)i;
        }
        float *o = out + n * (int64_t)(1 + m) * W + w;
        unsigned char *ar = arg ? arg + n * (int64_t)na * W + w : nullptr;
        o[0] = nrm;
        int s = 1, sa = 0;
        const float mean = cnt > 0 ? sum / (float)cnt : 0.f;
        if (stats & ST_MEAN) { o[(int64_t)s * W] = mean; ++s; }
        if (stats & ST_MAX) {
            float x = 0.f; int p = 0;
            if (cnt > 0) { int q = cnt - 1; x = sv[q * POOL_NT + tid]; while (q > 0 && sv[(q - 1) * POOL_NT + tid] == x) --q; p = sp[q * POOL_NT + tid]; }
            o[(int64_t)s * W] = x; ++s; if (ar) { ar[(int64_t)sa * W] = (unsigned char)p; } ++sa;
        }
        if (stats & ST_MIN) {
            float x = 0.f; int p = 0;
            if (cnt > 0) { x = sv[tid]; p = sp[tid]; }
            o[(int64_t)s * W] = x; ++s; if (ar) { ar[(int64_t)sa * W] = (unsigned char)p; } ++sa;
        }
        if (stats & ST_VAR) {
            float x = 0.f;
            if (cnt > 0) { x = sq / (float)cnt - mean * mean; x = x > 0.f ? x : 0.f; }
            o[(int64_t)s * W] = x; ++s;
        }
        for (int l = 0; l < n_levels; ++l) {
            float x = 0.f; int p = 0;
            if (cnt > 0) { const int k = k_table[l * table_ld + cnt]; x = sv[k * POOL_NT + tid]; p = sp[k * POOL_NT + tid]; }
            o[(int64_t)s * W] = x; ++s; if (ar) { ar[(int64_t)sa * W] = (unsigned char)p; } ++sa;
        }
    }
}

// dv (N, W, D) must be zero on entry: a coefficient collects from its own node's statistics and, as a neighbour, from
// those of every node it points to -- float atomics (each (node, wavelet) thread adds 2*D values per in-edge).
// TWO_PASS: this kernel stores the per-edge cosine gradients (gedge (E, W)) and each coefficient's gradient as a
// destination; k_vinv_bwd_src then adds what it receives as a source (edges grouped by source) -- no atomics.
template <int D, bool TWO_PASS>
__global__ void __launch_bounds__(POOL_NT)
k_vinv_bwd(const float *__restrict__ v, const int *__restrict__ src, const int *__restrict__ seg_ptr, int64_t n_nodes, int W,
           int stats, int n_levels, const float *__restrict__ gout, const unsigned char *__restrict__ arg, float *__restrict__ dv,
           float *__restrict__ gedge)
{
    extern __shared__ float lds[];
    float *sc = lds;                                 // [POOL_MAX_DEG][POOL_NT] cosines of this thread's segment
    const int tid = threadIdx.x;
    const int m = ((stats & ST_MEAN) ? 1 : 0) + ((stats & ST_MAX) ? 1 : 0) + ((stats & ST_MIN) ? 1 : 0) + ((stats & ST_VAR) ? 1 : 0) + n_levels;
    const int na = ((stats & ST_MAX) ? 1 : 0) + ((stats & ST_MIN) ? 1 : 0) + n_levels;
    const int64_t total = n_nodes * W;
    for (int64_t g = (int64_t)blockIdx.x * POOL_NT + tid; g < total; g += (int64_t)gridDim.x * POOL_NT) {
        const int64_t n = g / W; const int w = (int)(g - n * W);
        float a[D], da[D]; float nrm = 0.f;
#pragma unroll
        for (int c = 0; c < D; ++c) { a[c] = v[g * D + c]; nrm = fmaf(a[c], a[c], nrm); }
        nrm = sqrtf(nrm);
        const bool a_free = nrm > 1e-8f;             // the clamp passes gradient only where it does not bind
        const float nac = a_free ? nrm : 1e-8f;
        const float inac = 1.f / nac;
        const float *go = gout + n * (int64_t)(1 + m) * W + w;
        {
            const float gn = nrm > 0.f ? go[0] / nrm : 0.f;      // d|a|/da = a/|a| (0 at the origin)
#pragma unroll
            for (int c = 0; c < D; ++c) da[c] = gn * a[c];
        }
        const int e0 = seg_ptr[n], cnt = seg_ptr[n + 1] - e0;
        if (cnt > 0) {
            const unsigned char *ar = arg + n * (int64_t)na * W + w;
            float sum = 0.f, sq = 0.f;
            gather_cosines<D>(v, src, e0, cnt, W, w, a, nac, sc, tid, sum, sq);
            int s = 1;
            float gmean = 0.f, gvar = 0.f;
            if (stats & ST_MEAN) { gmean = go[(int64_t)s * W]; ++s; }
            const int s_max = s; if (stats & ST_MAX) ++s;
            const int s_min = s; if (stats & ST_MIN) ++s;
            const float mean = sum / (float)cnt;
            if (stats & ST_VAR) { gvar = go[(int64_t)s * W]; ++s; if (sq / (float)cnt - mean * mean <= 0.f) gvar = 0.f; }
            const int s_q = s;
            const float inv = 1.f / (float)cnt;
            // per-edge gradient of the statistics, built in place of the cosines: the dense part (mean, variance) first,
            // then each selected edge (max, min, quantiles) gets its share -- O(deg + stats) instead of O(deg * stats)
            for (int i = 0; i < cnt; ++i) {
                const float x = sc[i * POOL_NT + tid];
                sc[i * POOL_NT + tid] = gmean * inv + gvar * 2.f * (x - mean) * inv;
            }
            {
                int sa = 0;
                if (stats & ST_MAX) { sc[ar[(int64_t)sa * W] * POOL_NT + tid] += go[(int64_t)s_max * W]; ++sa; }
                if (stats & ST_MIN) { sc[ar[(int64_t)sa * W] * POOL_NT + tid] += go[(int64_t)s_min * W]; ++sa; }
                for (int l = 0; l < n_levels; ++l, ++sa) sc[ar[(int64_t)sa * W] * POOL_NT + tid] += go[(int64_t)(s_q + l) * W];
            }
            for (int i0 = 0; i0 < cnt; i0 += 4) {       // four edges in flight (see gather_cosines)
                float b[4][D]; int64_t jb[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int i = i0 + u < cnt ? i0 + u : cnt - 1;
                    jb[u] = ((int64_t)src[e0 + i] * W + w) * D;
#pragma unroll
                    for (int c = 0; c < D; ++c) b[u][c] = v[jb[u] + c];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (i0 + u >= cnt) continue;
                    const float gc = sc[(i0 + u) * POOL_NT + tid];
                    if (TWO_PASS) gedge[(int64_t)(e0 + i0 + u) * W + w] = gc;
                    if (gc == 0.f) continue;
                    // cos = <a,b> / (na nb): d/da = b/(na nb) - cos a/na^2 (second term only where |a| is not clamped), same for b
                    float nb = 0.f, dot = 0.f;
#pragma unroll
                    for (int c = 0; c < D; ++c) { nb = fmaf(b[u][c], b[u][c], nb); dot = fmaf(a[c], b[u][c], dot); }
                    const bool b_free = nb > 1e-16f;
                    const float inb = inv_norm(nb);
                    const float r = gc * inac * inb, x = dot * inac * inb;
                    const float ka = a_free ? gc * x * inac * inac : 0.f, kb = b_free ? gc * x * inb * inb : 0.f;
#pragma unroll
                    for (int c = 0; c < D; ++c) {
                        da[c] += r * b[u][c] - ka * a[c];
                        if (!TWO_PASS) atomicAdd(dv + jb[u] + c, r * a[c] - kb * b[u][c]);
                    }
                }
            }
        }
#pragma unroll
        for (int c = 0; c < D; ++c) { if (TWO_PASS) dv[g * D + c] = da[c]; else atomicAdd(dv + g * D + c, da[c]); }
    }
}

// second pass: what a coefficient receives as the SOURCE of edges.  out_edge[out_ptr[n] .. out_ptr[n+1]) = the edges
// (positions in the destination-grouped order) whose source is n; dst[e] = destination of edge e.
template <int D>
__global__ void __launch_bounds__(POOL_NT)
k_vinv_bwd_src(const float *__restrict__ v, const int *__restrict__ dst, const int *__restrict__ out_ptr,
               const int *__restrict__ out_edge, int64_t n_nodes, int W, const float *__restrict__ gedge, float *__restrict__ dv)
{
    const int64_t total = n_nodes * W;
    for (int64_t g = (int64_t)blockIdx.x * POOL_NT + threadIdx.x; g < total; g += (int64_t)gridDim.x * POOL_NT) {
        const int64_t n = g / W; const int w = (int)(g - n * W);
        const int p0 = out_ptr[n], p1 = out_ptr[n + 1];
        if (p0 == p1) continue;
        float b[D], db[D]; float nb = 0.f;
#pragma unroll
        for (int c = 0; c < D; ++c) { b[c] = v[g * D + c]; nb = fmaf(b[c], b[c], nb); db[c] = 0.f; }
        nb = sqrtf(nb);
        const bool b_free = nb > 1e-8f;
        const float inbc = 1.f / (b_free ? nb : 1e-8f);
        for (int q0 = p0; q0 < p1; q0 += 4) {
            float av[4][D], gcs[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int e = out_edge[q0 + u < p1 ? q0 + u : p1 - 1];
                gcs[u] = gedge[(int64_t)e * W + w];
                const float *pa = v + ((int64_t)dst[e] * W + w) * D;
#pragma unroll
                for (int c = 0; c < D; ++c) av[u][c] = pa[c];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float gc = gcs[u];
                if (q0 + u >= p1 || gc == 0.f) continue;
                float na = 0.f, dot = 0.f;
#pragma unroll
                for (int c = 0; c < D; ++c) { na = fmaf(av[u][c], av[u][c], na); dot = fmaf(av[u][c], b[c], dot); }
                const float ina = inv_norm(na);
                const float r = gc * ina * inbc, x = dot * ina * inbc;
                const float kb = b_free ? gc * x * inbc * inbc : 0.f;
#pragma unroll
                for (int c = 0; c < D; ++c) db[c] += r * av[u][c] - kb * b[c];
            }
        }
#pragma unroll
        for (int c = 0; c < D; ++c) dv[g * D + c] += db[c];
    }
}

static int pool_grid(int64_t work)
{
    int64_t b = (work + POOL_NT - 1) / POOL_NT;
    if (b < 1) b = 1;
    return (int)(b < 256 * 8 ? b : 256 * 8);
}

extern "C" int escat_neighbor_pool_f32(const float *vals, const int32_t *seg_ptr, int64_t n_nodes, int W, int max_deg,
                                       int stats, int n_levels, const int32_t *k_table, float *out, uint8_t *arg, void *stream)
{
    if (!vals || !seg_ptr || !out) return escat_fail(ESCAT_EINVAL, "NULL pointer");
    if (W < 1 || n_levels < 0 || (stats & ~15)) return escat_fail(ESCAT_EINVAL, "bad W / stats / n_levels");
    if (n_levels > 0 && !k_table) return escat_fail(ESCAT_EINVAL, "k_table is NULL");
    if (max_deg > POOL_MAX_DEG) return escat_fail(ESCAT_ERANGE, "max_deg=%d: at most %d incoming edges per node", max_deg, POOL_MAX_DEG);
    if (n_nodes <= 0) return n_nodes == 0 ? ESCAT_OK : escat_fail(ESCAT_EINVAL, "negative n_nodes");
    const size_t lds = (size_t)POOL_MAX_DEG * POOL_NT * 5;
    // (set on every call: the attribute is per device and the call costs microseconds)
    if (hipFuncSetAttribute((const void *)k_pool_fwd, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return escat_fail(ESCAT_ELAUNCH, "hipFuncSetAttribute(k_pool_fwd)");
    hipLaunchKernelGGL(k_pool_fwd, dim3(pool_grid(n_nodes * W)), dim3(POOL_NT), lds, (hipStream_t)stream, vals, seg_ptr, n_nodes, W,
                       stats, n_levels, k_table, POOL_MAX_DEG + 1, out, arg);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return escat_fail(ESCAT_ELAUNCH, "k_pool_fwd: %s", hipGetErrorString(e));
    return ESCAT_OK;
}

extern "C" int escat_neighbor_pool_bwd_f32(const float *vals, const int32_t *seg_ptr, int64_t n_nodes, int W, int stats,
                                           int n_levels, const float *gout, const uint8_t *arg, float *gvals, void *stream)
{
    if (!vals || !seg_ptr || !gout || !gvals) return escat_fail(ESCAT_EINVAL, "NULL pointer");
    const int na = ((stats & ST_MAX) ? 1 : 0) + ((stats & ST_MIN) ? 1 : 0) + n_levels;
    if (na > 0 && !arg) return escat_fail(ESCAT_EINVAL, "arg is NULL");
    if (W < 1 || n_levels < 0 || (stats & ~15)) return escat_fail(ESCAT_EINVAL, "bad W / stats / n_levels");
    if (n_nodes <= 0) return n_nodes == 0 ? ESCAT_OK : escat_fail(ESCAT_EINVAL, "negative n_nodes");
    hipLaunchKernelGGL(k_pool_bwd, dim3(pool_grid(n_nodes * W)), dim3(POOL_NT), 0, (hipStream_t)stream, vals, seg_ptr, n_nodes, W,
                       stats, n_levels, gout, arg, gvals);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return escat_fail(ESCAT_ELAUNCH, "k_pool_bwd: %s", hipGetErrorString(e));
    return ESCAT_OK;
}

extern "C" int escat_vector_invariants_f32(const float *v, const int32_t *src, const int32_t *seg_ptr, int64_t n_nodes, int W, int d,
                                           int max_deg, int stats, int n_levels, const int32_t *k_table, float *out, uint8_t *arg,
                                           void *stream)
{
    if (!v || !seg_ptr || !out || (!src && max_deg > 0)) return escat_fail(ESCAT_EINVAL, "NULL pointer");
    if (W < 1 || n_levels < 0 || (stats & ~15)) return escat_fail(ESCAT_EINVAL, "bad W / stats / n_levels");
    if (d != 2 && d != 3) return escat_fail(ESCAT_ERANGE, "d=%d: the fused kernel is built for d = 2, 3", d);
    if (n_levels > 0 && !k_table) return escat_fail(ESCAT_EINVAL, "k_table is NULL");
    if (max_deg > POOL_MAX_DEG) return escat_fail(ESCAT_ERANGE, "max_deg=%d: at most %d incoming edges per node", max_deg, POOL_MAX_DEG);
    if (n_nodes <= 0) return n_nodes == 0 ? ESCAT_OK : escat_fail(ESCAT_EINVAL, "negative n_nodes");
    const int cap = max_deg < 4 ? 4 : (max_deg + 3) & ~3;           // LDS column height: as low as the graph allows (occupancy)
    const size_t lds = (size_t)cap * POOL_NT * 5;
    {
        hipError_t e = d == 2 ? hipFuncSetAttribute((const void *)k_vinv_fwd<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)
                              : hipFuncSetAttribute((const void *)k_vinv_fwd<3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return escat_fail(ESCAT_ELAUNCH, "hipFuncSetAttribute(k_vinv_fwd)");
    }
    const int grid = pool_grid(n_nodes * W);
    if (d == 2) hipLaunchKernelGGL(k_vinv_fwd<2>, dim3(grid), dim3(POOL_NT), lds, (hipStream_t)stream, v, src, seg_ptr, n_nodes, W, stats, n_levels, k_table, POOL_MAX_DEG + 1, out, arg, cap);
    else        hipLaunchKernelGGL(k_vinv_fwd<3>, dim3(grid), dim3(POOL_NT), lds, (hipStream_t)stream, v, src, seg_ptr, n_nodes, W, stats, n_levels, k_table, POOL_MAX_DEG + 1, out, arg, cap);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return escat_fail(ESCAT_ELAUNCH, "k_vinv_fwd: %s", hipGetErrorString(e));
    return ESCAT_OK;
}

extern "C" int escat_vector_invariants_bwd_f32(const float *v, const int32_t *src, const int32_t *seg_ptr, int64_t n_nodes, int W,
                                               int d, int max_deg, int stats, int n_levels, const float *gout, const uint8_t *arg, float *dv,
                                               const int32_t *dst, const int32_t *out_ptr, const int32_t *out_edge, float *gedge,
                                               void *stream)
{
    if (!v || !seg_ptr || !gout || !dv) return escat_fail(ESCAT_EINVAL, "NULL pointer");
    const int na = ((stats & ST_MAX) ? 1 : 0) + ((stats & ST_MIN) ? 1 : 0) + n_levels;
    if (na > 0 && !arg) return escat_fail(ESCAT_EINVAL, "arg is NULL");
    if (W < 1 || n_levels < 0 || (stats & ~15)) return escat_fail(ESCAT_EINVAL, "bad W / stats / n_levels");
    if (d != 2 && d != 3) return escat_fail(ESCAT_ERANGE, "d=%d: the fused kernel is built for d = 2, 3", d);
    if (n_nodes <= 0) return n_nodes == 0 ? ESCAT_OK : escat_fail(ESCAT_EINVAL, "negative n_nodes");
    const bool two = gedge != nullptr;
    if (two && (!dst || !out_ptr || !out_edge)) return escat_fail(ESCAT_EINVAL, "the two-pass form needs dst, out_ptr and out_edge");
    if (max_deg > POOL_MAX_DEG) return escat_fail(ESCAT_ERANGE, "max_deg=%d: at most %d incoming edges per node", max_deg, POOL_MAX_DEG);
    const size_t lds = (size_t)(max_deg < 4 ? 4 : (max_deg + 3) & ~3) * POOL_NT * 4;
    const int which = (d == 2 ? 0 : 1) + (two ? 2 : 0);
    const void *fn = which == 0 ? (const void *)k_vinv_bwd<2, false> : which == 1 ? (const void *)k_vinv_bwd<3, false>
                   : which == 2 ? (const void *)k_vinv_bwd<2, true> : (const void *)k_vinv_bwd<3, true>;
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return escat_fail(ESCAT_ELAUNCH, "hipFuncSetAttribute(k_vinv_bwd)");
    const int grid = pool_grid(n_nodes * W);
    hipStream_t st = (hipStream_t)stream;
#define GO(DD, TP) hipLaunchKernelGGL((k_vinv_bwd<DD, TP>), dim3(grid), dim3(POOL_NT), lds, st, v, src, seg_ptr, n_nodes, W, stats, n_levels, gout, arg, dv, gedge)
    if (which == 0) GO(2, false); else if (which == 1) GO(3, false); else if (which == 2) GO(2, true); else GO(3, true);
#undef GO
    if (two) {
        if (d == 2) hipLaunchKernelGGL(k_vinv_bwd_src<2>, dim3(grid), dim3(POOL_NT), 0, st, v, dst, out_ptr, out_edge, n_nodes, W, gedge, dv);
        else        hipLaunchKernelGGL(k_vinv_bwd_src<3>, dim3(grid), dim3(POOL_NT), 0, st, v, dst, out_ptr, out_edge, n_nodes, W, gedge, dv);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return escat_fail(ESCAT_ELAUNCH, "k_vinv_bwd: %s", hipGetErrorString(e));
    return ESCAT_OK;
}
