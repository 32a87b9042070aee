// escat.hip -- gfx950 (MI355X, CDNA4) kernels + C ABI for ESc-GNN's
// diffusion-wavelet scattering chain.  See include/escat.h for the contract and
// DESIGN.md for the data layout and the roofline of each kernel.
//
// Kernel families
//   k_csr_step / k_bsr3_step      one product y = A x over the whole operator with
//                                 the wavelet bookkeeping fused into the epilogue
//                                 (used for graphs too large for LDS: cfg3).
//   k_chain_csr / k_chain_bsr3    one workgroup per diagonal block of the batch:
//                                 vectors ping-pong in LDS for all T steps, the
//                                 (rows, C, F) output slab is accumulated in LDS
//                                 and written once, fully coalesced.
// Lane mapping (wave = 64 lanes):
//   CSR   G lanes per row (G = 4/8/16 by mean row length), lane = one nonzero,
//         C accumulators per lane, xor-shuffle reduction inside the group.
//   BSR-3 3*g lanes per block row (g = 3 or 7 -> 9 or 21 lanes, 7 or 3 block rows
//         per wave, 63 live lanes): lane (k, a) owns row a of block k, loads its
//         12 contiguous bytes; a wave therefore reads 36 B x consecutive blocks
//         as one contiguous run.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
#include "../../include/escat.h"
#include "escat_internal.h"

// ------------------------------------------------------------------ errors
static thread_local char g_err[512] = "";
int escat_fail(int code, const char *fmt, ...)      // shared with escat_build.hip (escat_internal.h)
{
    va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof(g_err), fmt, ap); va_end(ap);
    return code;
}
#define fail escat_fail
extern "C" const char *escat_last_error(void) { return g_err; }
extern "C" int escat_abi_version(void) { return 1; }

#define CHECK_LAUNCH(what)                                                              \
    do { hipError_t e_ = hipGetLastError();                                             \
         if (e_ != hipSuccess) return fail(ESCAT_ELAUNCH, "%s: %s", what, hipGetErrorString(e_)); } while (0)

// ------------------------------------------------------------------ event tables (kernel arguments)
struct Ev { int step, slot, sign, chan; };          // sign: +1/-1; bit 8 set (|0x100) = first touch of (slot,chan): store
struct EvTable { int n; Ev e[ESCAT_MAX_EVENTS]; };  // whole chain (fused kernels)
#define STEP_MAX_EV 24
struct StepEv { int n; Ev e[STEP_MAX_EV]; };        // one step (multi-launch path)

#define EV_STORE 0x100
// encoding: bit0 = valid, bit1 = negative, bit8 = store
static inline int enc_sign(int sign, bool store) { return 1 | (sign < 0 ? 2 : 0) | (store ? EV_STORE : 0); }

// validates + encodes the host event list; marks the first event of every (slot, chan)
static int build_table(const escat_event *ev, int n_ev, int T, int F, int C, EvTable *tab)
{
    if (n_ev < 0 || n_ev > ESCAT_MAX_EVENTS) return fail(ESCAT_ERANGE, "n_events=%d exceeds ESCAT_MAX_EVENTS", n_ev);
    if (n_ev > 0 && !ev) return fail(ESCAT_EINVAL, "events is NULL");
    if (F < 1 || F > 64) return fail(ESCAT_ERANGE, "F=%d out of range [1,64]", F);
    static thread_local unsigned char seen[64 * ESCAT_MAX_CHANNELS];
    memset(seen, 0, sizeof(seen));
    tab->n = n_ev;
    for (int k = 0; k < n_ev; ++k) {
        if (ev[k].step < 0 || ev[k].step > T) return fail(ESCAT_EINVAL, "event %d: step %d outside [0,%d]", k, ev[k].step, T);
        if (k && ev[k].step < ev[k - 1].step) return fail(ESCAT_EINVAL, "events not sorted by step");
        if (ev[k].slot < 0 || ev[k].slot >= F) return fail(ESCAT_EINVAL, "event %d: slot %d outside [0,%d)", k, ev[k].slot, F);
        if (ev[k].chan >= C) return fail(ESCAT_EINVAL, "event %d: chan %d >= C=%d", k, ev[k].chan, C);
        if (ev[k].sign != 1 && ev[k].sign != -1) return fail(ESCAT_EINVAL, "event %d: sign must be +1/-1", k);
        const int c0 = ev[k].chan < 0 ? 0 : ev[k].chan, c1 = ev[k].chan < 0 ? C : ev[k].chan + 1;
        int touched = 0;
        for (int c = c0; c < c1; ++c) touched += seen[ev[k].slot * ESCAT_MAX_CHANNELS + c];
        if (touched != 0 && touched != c1 - c0)
            return fail(ESCAT_EINVAL, "event %d mixes touched and untouched channels of slot %d", k, ev[k].slot);
        const bool first = touched == 0;      // first touch of these OUT elements: store instead of accumulate
        for (int c = c0; c < c1; ++c) seen[ev[k].slot * ESCAT_MAX_CHANNELS + c] = 1;
        tab->e[k].step = ev[k].step; tab->e[k].slot = ev[k].slot; tab->e[k].chan = ev[k].chan;
        tab->e[k].sign = enc_sign(ev[k].sign, first);
    }
    for (int f = 0; f < F; ++f)
        for (int c = 0; c < C; ++c)
            if (!seen[f * ESCAT_MAX_CHANNELS + c])
                return fail(ESCAT_EINVAL, "OUT slot %d channel %d receives no event", f, c);
    return ESCAT_OK;
}

static int step_events(const EvTable &tab, int step, StepEv *se)
{
    se->n = 0;
    for (int k = 0; k < tab.n; ++k)
        if (tab.e[k].step == step) {
            if (se->n >= STEP_MAX_EV) return fail(ESCAT_ERANGE, "more than %d events at step %d", STEP_MAX_EV, step);
            se->e[se->n++] = tab.e[k];
        }
    return ESCAT_OK;
}

// ------------------------------------------------------------------ device helpers
template <int G> __device__ __forceinline__ float group_sum(float v)
{
#pragma unroll
    for (int m = G / 2; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}

// apply forward events of one step to element (row, c) whose fresh value is y
__device__ __forceinline__ void emit_one(float *o /* &OUT[row, c, 0] */, float y, int slot, int s)
{
    if (s & EV_STORE) o[slot] = (s & 2) ? -y : y;
    else o[slot] = (s & 2) ? o[slot] - y : o[slot] + y;
}

// ------------------------------------------------------------------ multi-launch path: scalar CSR
// MODE 0: y = A x.   MODE 1: forward step (emit y into OUT).   MODE 2: backward step (y += gathered G).
template <int C, int G, int MODE>
__global__ void __launch_bounds__(256)
k_csr_step(const int *__restrict__ rowptr, const int *__restrict__ col, const float *__restrict__ val,
           const float *__restrict__ xin, int ldx, float *__restrict__ yout, int ldy, int64_t n,
           StepEv ev, float *OUT, int64_t ld_row, int F)
{
    const int lg = threadIdx.x % G;
    const int64_t groups = (int64_t)gridDim.x * (256 / G);
    for (int64_t row = (int64_t)blockIdx.x * (256 / G) + threadIdx.x / G; row < n; row += groups) {
        float acc[C];
#pragma unroll
        for (int c = 0; c < C; ++c) acc[c] = 0.f;
        const int e1 = rowptr[row + 1];
        for (int e = rowptr[row] + lg; e < e1; e += G) {
            const float v = val[e];
            const float *x = xin + (int64_t)col[e] * ldx;
#pragma unroll
            for (int c = 0; c < C; ++c) acc[c] += v * x[c];
        }
#pragma unroll
        for (int c = 0; c < C; ++c) acc[c] = group_sum<G>(acc[c]);
#pragma unroll
        for (int c = 0; c < C; ++c) {
            if (c % G != lg) continue;          // lane c%G owns channel c
            float y = acc[c];
            if (MODE == 2) {
                const float *g = OUT + row * ld_row + (int64_t)c * F;
                for (int k = 0; k < ev.n; ++k)
                    if (ev.e[k].chan < 0 || ev.e[k].chan == c) y += (ev.e[k].sign & 2) ? -g[ev.e[k].slot] : g[ev.e[k].slot];
            }
            yout[row * ldy + c] = y;
            if (MODE == 1) {
                float *o = OUT + row * ld_row + (int64_t)c * F;
                for (int k = 0; k < ev.n; ++k)
                    if (ev.e[k].chan < 0 || ev.e[k].chan == c) emit_one(o, y, ev.e[k].slot, ev.e[k].sign);
            }
        }
    }
}

// elementwise: forward -> emit events of a step for the vector v itself (step 0);
// backward -> v = sum of gathered G (the Horner start, or the whole answer when T == 0).
template <int MODE>
__global__ void __launch_bounds__(256)
k_events_only(float *v, int ldv, int64_t n, int C, StepEv ev, float *OUT, int64_t ld_row, int F, int pad3)
{
    // pad3: v is a float4-per-node vector (scalar row r lives at (r/3)*4 + r%3); only used with C == 1
    const int64_t total = n * C;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t row = i / C; const int c = (int)(i % C);
        float *o = OUT + row * ld_row + (int64_t)c * F;
        const int64_t va = pad3 ? (row / 3) * 4 + row % 3 : row * ldv + c;
        if (MODE == 1) {
            const float y = v[va];
            for (int k = 0; k < ev.n; ++k)
                if (ev.e[k].chan < 0 || ev.e[k].chan == c) emit_one(o, y, ev.e[k].slot, ev.e[k].sign);
        } else {
            float y = 0.f;
            for (int k = 0; k < ev.n; ++k)
                if (ev.e[k].chan < 0 || ev.e[k].chan == c) y += (ev.e[k].sign & 2) ? -o[ev.e[k].slot] : o[ev.e[k].slot];
            v[va] = y;
        }
    }
}

// ------------------------------------------------------------------ multi-launch path: BSR-3
template <int g, int MODE>
__global__ void __launch_bounds__(256)
k_bsr3_step(const int *__restrict__ rowptr, const int *__restrict__ col, const float *__restrict__ blk,
            const float *__restrict__ xin, float *__restrict__ yout, int64_t nb,
            StepEv ev, float *OUT, int F)
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
                const float *x = xin + (int64_t)col[e] * 3;
                acc += b[0] * x[0] + b[1] * x[1] + b[2] * x[2];
            }
        }
        float tot = 0.f;
#pragma unroll
        for (int kk = 0; kk < g; ++kk) {
            const int src = live ? sub * GL + 3 * kk + a : lane;
            tot += __shfl(acc, src, 64);
        }
        if (live && k == 0) {
            const int64_t r = 3 * brow + a;
            float y = tot;
            if (MODE == 2) {
                const float *gg = OUT + r * F;
                for (int q = 0; q < ev.n; ++q) y += (ev.e[q].sign & 2) ? -gg[ev.e[q].slot] : gg[ev.e[q].slot];
            }
            yout[r] = y;
            if (MODE == 1) {
                float *o = OUT + r * F;
                for (int q = 0; q < ev.n; ++q) emit_one(o, y, ev.e[q].slot, ev.e[q].sign);
            }
        }
    }
}

// ------------------------------------------------------------------ multi-launch path on a packed operator (large graphs)
// "SELL-64, global columns": rows sorted by length inside windows of a few thousand rows and cut into slices of 64;
// slice values slot-major exactly as in the per-part pack (escat_sell), one 32-bit GLOBAL block column per slot
// (-1 = padding), perm = global block row served by each lane.  One wave owns one slice: all of its value/column
// loads are independent, fully coalesced 1-KiB / 256-B accesses (the CSR kernels above chain rowptr -> col -> x and
// are latency-bound), the only dependent access is the x gather.  Intermediate vectors are float4 per node so that a
// gather is one aligned 16-B load.
struct SellG { const float *val; const int *col32; const int *perm; const int *slice_doff; };

template <int MODE, bool XIN_PACKED, bool YOUT_PACKED>
__global__ void __launch_bounds__(256)
k_bsr3_step_sell(SellG sg, int n_slices, const float *__restrict__ xin, float *__restrict__ yout,
                 StepEv ev, float *OUT, int F)
{
    const int lane = threadIdx.x & 63;
    const int s = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (s >= n_slices) return;
    const int doff = sg.slice_doff[s];
    const int w = __builtin_amdgcn_readfirstlane(sg.slice_doff[s + 1] - doff);
    const int row = sg.perm[(int64_t)s * 64 + lane];
    const float *vb = sg.val + (int64_t)doff * 576;
    const int *cb = sg.col32 + (int64_t)doff * 64 + lane;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
    auto slot = [&](int d) {
        const int c = cb[(int64_t)d * 64];
        const float4 p0 = *reinterpret_cast<const float4 *>(vb + (int64_t)d * 576 + lane * 4);
        const float4 p1 = *reinterpret_cast<const float4 *>(vb + (int64_t)d * 576 + 256 + lane * 4);
        const float p2 = vb[(int64_t)d * 576 + 512 + lane];
        if (c >= 0) {
            float x0, x1, x2;
            if (XIN_PACKED) { const float *x = xin + (int64_t)c * 3; x0 = x[0]; x1 = x[1]; x2 = x[2]; }
            else { const float4 xv = *reinterpret_cast<const float4 *>(xin + (int64_t)c * 4); x0 = xv.x; x1 = xv.y; x2 = xv.z; }
            a0 = fmaf(p0.x, x0, a0); a0 = fmaf(p0.y, x1, a0); a0 = fmaf(p0.z, x2, a0);
            a1 = fmaf(p0.w, x0, a1); a1 = fmaf(p1.x, x1, a1); a1 = fmaf(p1.y, x2, a1);
            a2 = fmaf(p1.z, x0, a2); a2 = fmaf(p1.w, x1, a2); a2 = fmaf(p2, x2, a2);
        }
    };
    int d = 0;
    for (; d + 4 <= w; d += 4) { slot(d); slot(d + 1); slot(d + 2); slot(d + 3); }
    for (; d < w; ++d) slot(d);
    if (row < 0) return;
    float y[3] = {a0, a1, a2};
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const int64_t r = 3 * (int64_t)row + a;
        float v = y[a];
        if (MODE == 2) {
            const float *gg = OUT + r * F;
            for (int q = 0; q < ev.n; ++q) v += (ev.e[q].sign & 2) ? -gg[ev.e[q].slot] : gg[ev.e[q].slot];
        }
        if (YOUT_PACKED) yout[r] = v; else yout[(int64_t)row * 4 + a] = v;
        if (MODE == 1) {
            float *o = OUT + r * F;
            for (int q = 0; q < ev.n; ++q) emit_one(o, v, ev.e[q].slot, ev.e[q].sign);
        }
    }
}

// ------------------------------------------------------------------ fused per-part chains (vectors resident in LDS)
// LDS layout (floats): xa[rows*C] | xb[rows*C] | slab[rows*C*F]
// forward : slab accumulates OUT for the part, written once at the end.
// backward: slab holds the part's upstream gradient, read-only.
#define CHAIN_NT 256

template <int C, int G, bool BWD>
__global__ void __launch_bounds__(CHAIN_NT)
k_chain_csr(const int *__restrict__ rowptr, const int *__restrict__ col, const float *__restrict__ val,
            const int *__restrict__ part_ptr, const float *__restrict__ X /* fwd: x, bwd: G */,
            float *__restrict__ OUT /* fwd: OUT, bwd: dX */, int T, EvTable tab, int F)
{
    extern __shared__ __align__(16) float smem[];
    const int gs = part_ptr[blockIdx.x], rows = part_ptr[blockIdx.x + 1] - gs;
    const int nc = rows * C;
    float *xa = smem, *xb = smem + nc, *slab = smem + 2 * nc;
    const int tid = threadIdx.x;
    const int lg = tid % G, grp = tid / G;
    constexpr int NGRP = CHAIN_NT / G;

    if (!BWD) {
        const float *xg = X + (int64_t)gs * C;
        for (int i = tid; i < nc; i += CHAIN_NT) xa[i] = xg[i];
    } else {
        const float *gg = X + (int64_t)gs * C * F;
        for (int i = tid; i < nc * F; i += CHAIN_NT) slab[i] = gg[i];
    }
    __syncthreads();

    int p = BWD ? tab.n : 0;   // event cursor (uniform)
    if (!BWD) {
        // events at step 0 act on x itself
        int p1 = p; while (p1 < tab.n && tab.e[p1].step == 0) ++p1;
        for (int i = tid; i < nc; i += CHAIN_NT) {
            const int c = i % C; const float y = xa[i];
            for (int k = p; k < p1; ++k)
                if (tab.e[k].chan < 0 || tab.e[k].chan == c) emit_one(slab + (int64_t)i * F, y, tab.e[k].slot, tab.e[k].sign);
        }
        p = p1;
    } else {
        // Horner start: acc_T = sum of events at step T
        int p0 = p; while (p0 > 0 && tab.e[p0 - 1].step == T) --p0;
        for (int i = tid; i < nc; i += CHAIN_NT) {
            const int c = i % C; float y = 0.f;
            for (int k = p0; k < p; ++k)
                if (tab.e[k].chan < 0 || tab.e[k].chan == c) y += (tab.e[k].sign & 2) ? -slab[i * F + tab.e[k].slot] : slab[i * F + tab.e[k].slot];
            xa[i] = y;
        }
        p = p0;
    }
    __syncthreads();

    for (int t = 1; t <= T; ++t) {
        // forward: this product yields y_t, events at step t.  backward: yields acc_{T-t}, events at step T-t.
        const int st = BWD ? T - t : t;
        int q0, q1;
        if (!BWD) { q0 = p; q1 = p; while (q1 < tab.n && tab.e[q1].step == st) ++q1; p = q1; }
        else { q1 = p; q0 = p; while (q0 > 0 && tab.e[q0 - 1].step == st) --q0; p = q0; }
        for (int r = grp; r < rows; r += NGRP) {
            float acc[C];
#pragma unroll
            for (int c = 0; c < C; ++c) acc[c] = 0.f;
            const int e1 = rowptr[gs + r + 1];
            for (int e = rowptr[gs + r] + lg; e < e1; e += G) {
                const float v = val[e];
                const float *x = xa + (col[e] - gs) * C;
#pragma unroll
                for (int c = 0; c < C; ++c) acc[c] += v * x[c];
            }
#pragma unroll
            for (int c = 0; c < C; ++c) acc[c] = group_sum<G>(acc[c]);
#pragma unroll
            for (int c = 0; c < C; ++c) {
                if (c % G != lg) continue;
                float y = acc[c];
                float *o = slab + (r * C + c) * F;
                if (BWD) {
                    for (int k = q0; k < q1; ++k)
                        if (tab.e[k].chan < 0 || tab.e[k].chan == c) y += (tab.e[k].sign & 2) ? -o[tab.e[k].slot] : o[tab.e[k].slot];
                } else {
                    for (int k = q0; k < q1; ++k)
                        if (tab.e[k].chan < 0 || tab.e[k].chan == c) emit_one(o, y, tab.e[k].slot, tab.e[k].sign);
                }
                xb[r * C + c] = y;
            }
        }
        __syncthreads();
        float *s = xa; xa = xb; xb = s;
    }
    if (!BWD) {
        float *og = OUT + (int64_t)gs * C * F;
        for (int i = tid; i < nc * F; i += CHAIN_NT) og[i] = slab[i];
    } else {
        float *dg = OUT + (int64_t)gs * C;
        for (int i = tid; i < nc; i += CHAIN_NT) dg[i] = xa[i];
    }
}

template <int g, bool BWD>
__global__ void __launch_bounds__(CHAIN_NT)
k_chain_bsr3(const int *__restrict__ rowptr, const int *__restrict__ col, const float *__restrict__ blk,
             const int *__restrict__ part_ptr, const float *__restrict__ X, float *__restrict__ OUT,
             int T, EvTable tab, int F)
{
    extern __shared__ __align__(16) float smem[];
    constexpr int GL = 3 * g, RPW = 64 / GL, NW = CHAIN_NT / 64;
    const int gs = part_ptr[blockIdx.x], brows = part_ptr[blockIdx.x + 1] - gs;
    const int nc = brows * 3;
    float *xa = smem, *xb = smem + nc, *slab = smem + 2 * nc;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int sub = lane / GL, l = lane % GL, k = l / 3, a = l % 3;

    if (!BWD) {
        const float *xg = X + (int64_t)gs * 3;
        for (int i = tid; i < nc; i += CHAIN_NT) xa[i] = xg[i];
    } else {
        const float *gg = X + (int64_t)gs * 3 * F;
        for (int i = tid; i < nc * F; i += CHAIN_NT) slab[i] = gg[i];
    }
    __syncthreads();
    int p = BWD ? tab.n : 0;
    if (!BWD) {
        int p1 = p; while (p1 < tab.n && tab.e[p1].step == 0) ++p1;
        for (int i = tid; i < nc; i += CHAIN_NT) {
            const float y = xa[i];
            for (int q = p; q < p1; ++q) emit_one(slab + i * F, y, tab.e[q].slot, tab.e[q].sign);
        }
        p = p1;
    } else {
        int p0 = p; while (p0 > 0 && tab.e[p0 - 1].step == T) --p0;
        for (int i = tid; i < nc; i += CHAIN_NT) {
            float y = 0.f;
            for (int q = p0; q < p; ++q) y += (tab.e[q].sign & 2) ? -slab[i * F + tab.e[q].slot] : slab[i * F + tab.e[q].slot];
            xa[i] = y;
        }
        p = p0;
    }
    __syncthreads();

    for (int t = 1; t <= T; ++t) {
        const int st = BWD ? T - t : t;
        int q0, q1;
        if (!BWD) { q0 = p; q1 = p; while (q1 < tab.n && tab.e[q1].step == st) ++q1; p = q1; }
        else { q1 = p; q0 = p; while (q0 > 0 && tab.e[q0 - 1].step == st) --q0; p = q0; }
        for (int base = wv * RPW; base < brows; base += NW * RPW) {
            const int br = base + sub;
            const bool live = sub < RPW && br < brows;
            float acc = 0.f;
            if (live) {
                const int e1 = rowptr[gs + br + 1];
                for (int e = rowptr[gs + br] + k; e < e1; e += g) {
                    const float *b = blk + (int64_t)e * 9 + a * 3;
                    const float *x = xa + (col[e] - gs) * 3;
                    acc += b[0] * x[0] + b[1] * x[1] + b[2] * x[2];
                }
            }
            float tot = 0.f;
#pragma unroll
            for (int kk = 0; kk < g; ++kk) {
                const int src = live ? sub * GL + 3 * kk + a : lane;
                tot += __shfl(acc, src, 64);
            }
            if (live && k == 0) {
                const int r = 3 * br + a;
                float y = tot;
                float *o = slab + r * F;
                if (BWD) { for (int q = q0; q < q1; ++q) y += (tab.e[q].sign & 2) ? -o[tab.e[q].slot] : o[tab.e[q].slot]; }
                else     { for (int q = q0; q < q1; ++q) emit_one(o, y, tab.e[q].slot, tab.e[q].sign); }
                xb[r] = y;
            }
        }
        __syncthreads();
        float *s = xa; xa = xb; xb = s;
    }
    if (!BWD) {
        float *og = OUT + (int64_t)gs * 3 * F;
        for (int i = tid; i < nc * F; i += CHAIN_NT) og[i] = slab[i];
    } else {
        float *dg = OUT + (int64_t)gs * 3;
        for (int i = tid; i < nc; i += CHAIN_NT) dg[i] = xa[i];
    }
}

// ------------------------------------------------------------------ register-resident chains
// One thread owns one row of its part for the whole chain: the row's first DMAX
// entries (values + packed 16-bit local column offsets) live in VGPRs across all T
// steps, so the operator is read from HBM exactly ONCE per chain (from the packed
// SELL-64 copy below, fully coalesced); rows longer than DMAX stream their tail
// from L2.  Per step a thread gathers x from LDS and
// accumulates sequentially in ascending column order with fmaf -- the order of the
// reference's CPU product.  Vectors ping-pong in LDS, the output slab accumulates in
// LDS and is written once, coalesced.
struct __attribute__((packed, aligned(4))) f3 { float x, y, z; };

__device__ __forceinline__ int wave_max_i(int v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) { const int o = __shfl_xor(v, m, 64); v = o > v ? o : v; }
    return __builtin_amdgcn_readfirstlane(v);
}

// LDS vectors of these kernels are padded and kept in TWO copies: row r, channel c lives at r*CP + c (copy A) and
// at (R + copyB(r))*CP + c (copy B), CP = C rounded up to 1/2/4/8 floats so that a gather is one aligned
// ds_read_b32/b64/b128, R = rows + 1 rounded up to 16.  Copy B permutes the rows (bits 0-3 of the index rotate to the
// top of a 256-row block), so a row's 16-byte bank slot is (r & 15) in copy A and ((r >> 4) & 15) in copy B.  At pack
// time every gather is given the copy -- and every row's entries the slot order -- that keeps the 16 lanes of a
// ds_read_b128 lane group on distinct bank slots (k_sell_plan below); this cuts the measured LDS bank-conflict cycles
// of the random gather, the bound of the step loop.  One all-zero row (index `rows`, copy A) is the target of padded slots.
__host__ __device__ __forceinline__ int sell_padR(int rows) { return (rows + 1 + 15) & ~15; }
__host__ __device__ __forceinline__ int sell_copyB(int p, int R)
{
    int m = 8;
    if (R < 256) { m = 0; while ((2 << m) <= R) ++m; }       // floor(log2(R))
    if (m < 5) return p;
    const int blk = 1 << m, lim = (R >> m) << m;
    if (p >= lim) return p;
    const int lo = p & (blk - 1);
    return (p & ~(blk - 1)) | (lo >> 4) | ((lo & 15) << (m - 4));
}

// loads x (fwd) or the upstream-gradient slab (bwd) of a part into LDS, applies the
// step-0 events (fwd) / forms the Horner start (bwd); returns the event cursor.
template <bool BWD, int NT>
__device__ __forceinline__ int chain_prologue(float *xa, float *xb, float *slab, const float *__restrict__ IN,
                                              int64_t first_row, int rows, int R, int C, int CP, int F, int T, const EvTable &tab)
{
    const int tid = threadIdx.x;
    const int nc = rows * C;
    int p;
    if (tid < CP) { xa[rows * CP + tid] = 0.f; xb[rows * CP + tid] = 0.f; }
    if (!BWD) {
        const float *xg = IN + first_row * C;
        int p1 = 0; while (p1 < tab.n && tab.e[p1].step == 0) ++p1;
        for (int i = tid; i < nc; i += NT) {
            const int c = i % C, r = i / C; const float y = xg[i];
            xa[r * CP + c] = y;
            xa[(R + sell_copyB(r, R)) * CP + c] = y;
            for (int k = 0; k < p1; ++k)
                if (tab.e[k].chan < 0 || tab.e[k].chan == c) emit_one(slab + i * F, y, tab.e[k].slot, tab.e[k].sign);
        }
        p = p1;
    } else {
        const float *gg = IN + first_row * C * F;
        for (int i = tid; i < nc * F; i += NT) slab[i] = gg[i];
        __syncthreads();
        int p0 = tab.n; while (p0 > 0 && tab.e[p0 - 1].step == T) --p0;
        for (int i = tid; i < nc; i += NT) {
            const int c = i % C, r = i / C; float y = 0.f;
            for (int k = p0; k < tab.n; ++k)
                if (tab.e[k].chan < 0 || tab.e[k].chan == c) y += (tab.e[k].sign & 2) ? -slab[i * F + tab.e[k].slot] : slab[i * F + tab.e[k].slot];
            xa[r * CP + c] = y;
            xa[(R + sell_copyB(r, R)) * CP + c] = y;
        }
        p = p0;
    }
    __syncthreads();
    return p;
}

template <bool BWD, int NT>
__device__ __forceinline__ void chain_epilogue(const float *xa, const float *slab, float *__restrict__ RES, int64_t first_row,
                                               int rows, int C, int CP, int F)
{
    const int tid = threadIdx.x;
    const int nc = rows * C;
    if (!BWD) {
        float *og = RES + first_row * C * F;
        for (int i = tid; i < nc * F; i += NT) og[i] = slab[i];
    } else {
        float *dg = RES + first_row * C;
        for (int i = tid; i < nc; i += NT) dg[i] = xa[(i / C) * CP + i % C];
    }
}

// events of step `st`: [q0, q1) in the table; moves the cursor (fwd: up, bwd: down)
template <bool BWD>
__device__ __forceinline__ void step_range(const EvTable &tab, int st, int &p, int &q0, int &q1)
{
    if (!BWD) { q0 = p; q1 = p; while (q1 < tab.n && tab.e[q1].step == st) ++q1; p = q1; }
    else { q1 = p; q0 = p; while (q0 > 0 && tab.e[q0 - 1].step == st) --q0; p = q0; }
}

template <bool BWD>
__device__ __forceinline__ float apply_events(const EvTable &tab, int q0, int q1, int c, float y, float *o)
{
    if (BWD) {
        for (int k = q0; k < q1; ++k)
            if (tab.e[k].chan < 0 || tab.e[k].chan == c) y += (tab.e[k].sign & 2) ? -o[tab.e[k].slot] : o[tab.e[k].slot];
    } else {
        for (int k = q0; k < q1; ++k)
            if (tab.e[k].chan < 0 || tab.e[k].chan == c) emit_one(o, y, tab.e[k].slot, tab.e[k].sign);
    }
    return y;
}

// Packed operator for these kernels ("SELL-64": sliced ELLPACK, slice = one wave).
// Inside each part the rows are sorted by length (descending) and cut into slices
// of 64; slice s stores its first w_s = min(longest row, dmax) entries slot-major,
//   CSR   val[(doff_s + d) * 64 + lane]
//   BSR-3 val[(doff_s + d) * 576 + {0: lane*4 + 0..3 | 256: lane*4 + 0..3 | 512: lane}]
//         (the 9 floats of a block as two float4 pieces and one scalar piece)
//   lc[(poff_s + d/2) * 64 + lane] = local column of slot d | local column of slot d+1 << 16
// so that every load of a wave is one fully coalesced 256-B or 1-KiB access, and
// the sort makes w_s the length of (nearly) every row of the slice: few idle slots.
// Slots past a row's end hold value 0 and local column = rows of the part (the
// all-zero LDS row), so the step loop needs no per-lane predication.
// perm[s*64 + lane] is the row (local to the part) that lane serves, -1 = none.
struct SellDev {
    const float *val; const unsigned *lc; const int *perm;
    const int *slice_doff; const int *slice_poff; const int *part_slice_ptr;
    int dmax;
};

#define SLOT_GROUP 4      // slots issued together: their LDS gathers are in flight at once

template <int DMAX, int NT, bool BWD>
__global__ void __launch_bounds__(NT)
k_chain_bsr3_sell(const int *__restrict__ rowptr, const int *__restrict__ col, const float *__restrict__ blk,
                  const int *__restrict__ part_ptr, SellDev sd, const float *__restrict__ X, float *__restrict__ OUT,
                  int T, EvTable tab, int F)
{
    constexpr int GRP = SLOT_GROUP;
    extern __shared__ __align__(16) float smem[];
    const int gs = part_ptr[blockIdx.x], brows = part_ptr[blockIdx.x + 1] - gs;
    if (brows > NT) return;                       // host guarantees max_part_rows <= NT
    const int R = sell_padR(brows);
    const int nv = 2 * R * 4;                     // two copies of a float4-per-node vector (incl. the zero node)
    float *xa = smem, *xb = smem + nv, *slab = smem + 2 * nv;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int s0 = sd.part_slice_ptr[blockIdx.x], ns = sd.part_slice_ptr[blockIdx.x + 1] - s0;
    const unsigned zoff = (unsigned)brows * 4u;

    float b[DMAX * 9];
    unsigned lc[(DMAX + 1) / 2];
    int r = -1, e0 = 0, deg = 0, w = 0;
#pragma unroll
    for (int q = 0; q < DMAX * 9; ++q) b[q] = 0.f;
#pragma unroll
    for (int q = 0; q < (DMAX + 1) / 2; ++q) lc[q] = zoff | (zoff << 16);
    if (wave < ns) {
        const int s = __builtin_amdgcn_readfirstlane(s0 + wave);
        const int doff = __builtin_amdgcn_readfirstlane(sd.slice_doff[s]), poff = __builtin_amdgcn_readfirstlane(sd.slice_poff[s]);
        w = __builtin_amdgcn_readfirstlane(sd.slice_doff[s + 1] - doff);
        r = sd.perm[s * 64 + lane];
        if (r >= 0) { e0 = rowptr[gs + r]; deg = rowptr[gs + r + 1] - e0; }
        const float *vb = sd.val + (int64_t)doff * 576;
#pragma unroll
        for (int d = 0; d < DMAX; ++d) {
            if (d < w) {                          // wave-uniform
                const float4 p0 = *reinterpret_cast<const float4 *>(vb + d * 576 + lane * 4);
                const float4 p1 = *reinterpret_cast<const float4 *>(vb + d * 576 + 256 + lane * 4);
                const float p2 = vb[d * 576 + 512 + lane];
                b[9 * d + 0] = p0.x; b[9 * d + 1] = p0.y; b[9 * d + 2] = p0.z; b[9 * d + 3] = p0.w;
                b[9 * d + 4] = p1.x; b[9 * d + 5] = p1.y; b[9 * d + 6] = p1.z; b[9 * d + 7] = p1.w;
                b[9 * d + 8] = p2;
            }
        }
#pragma unroll
        for (int j2 = 0; j2 < (DMAX + 1) / 2; ++j2) {
            if (2 * j2 < w) {
                const unsigned v = sd.lc[(int64_t)(poff + j2) * 64 + lane];
                lc[j2] = ((v & 0xffffu) * 4u) | (((v >> 16) * 4u) << 16);      // float offsets into the LDS vector
            }
        }
    }
    const bool own = r >= 0;
    const int rB = own ? R + sell_copyB(r, R) : 0;

    int p = chain_prologue<BWD, NT>(xa, xb, slab, X, gs, brows, R, 3, 4, F, T, tab);
    for (int t = 1; t <= T; ++t) {
        int q0, q1;
        step_range<BWD>(tab, BWD ? T - t : t, p, q0, q1);
        float a0 = 0.f, a1 = 0.f, a2 = 0.f;
#pragma unroll
        for (int d0 = 0; d0 < DMAX; d0 += GRP) {
            if (d0 < w) {                         // wave-uniform: slice width, in groups
                float4 xv[GRP];
#pragma unroll
                for (int g = 0; g < GRP; ++g) {
                    const int d = d0 + g;
                    if (d < DMAX) {
                        const unsigned off = (d & 1) ? (lc[d / 2] >> 16) : (lc[d / 2] & 0xffffu);
                        xv[g] = *reinterpret_cast<const float4 *>(xa + off);
                    }
                }
#pragma unroll
                for (int g = 0; g < GRP; ++g) {
                    const int d = d0 + g;
                    if (d < DMAX) {
                        const float x0 = xv[g].x, x1 = xv[g].y, x2 = xv[g].z;
                        a0 = fmaf(b[9 * d + 0], x0, a0); a0 = fmaf(b[9 * d + 1], x1, a0); a0 = fmaf(b[9 * d + 2], x2, a0);
                        a1 = fmaf(b[9 * d + 3], x0, a1); a1 = fmaf(b[9 * d + 4], x1, a1); a1 = fmaf(b[9 * d + 5], x2, a1);
                        a2 = fmaf(b[9 * d + 6], x0, a2); a2 = fmaf(b[9 * d + 7], x1, a2); a2 = fmaf(b[9 * d + 8], x2, a2);
                        // keep the pad lane live: otherwise the gather is narrowed to ds_read_b96 (8 LDS cycles, 8-lane
                        // groups on 32 banks) instead of ds_read_b128 (4 cycles, the 16-lane groups the pack is planned for)
                        asm volatile("" :: "v"(xv[g].w));
                    }
                }
            }
        }
        if (own) {
            for (int e = e0 + sd.dmax; e < e0 + deg; ++e) {   // rows longer than the pack: tail from L2
                const float *bb = blk + (int64_t)e * 9;
                const float *x = xa + (col[e] - gs) * 4;
                const float x0 = x[0], x1 = x[1], x2 = x[2];
                a0 = fmaf(bb[0], x0, a0); a0 = fmaf(bb[1], x1, a0); a0 = fmaf(bb[2], x2, a0);
                a1 = fmaf(bb[3], x0, a1); a1 = fmaf(bb[4], x1, a1); a1 = fmaf(bb[5], x2, a1);
                a2 = fmaf(bb[6], x0, a2); a2 = fmaf(bb[7], x1, a2); a2 = fmaf(bb[8], x2, a2);
            }
            float *o = slab + 3 * r * F;
            a0 = apply_events<BWD>(tab, q0, q1, 0, a0, o);
            a1 = apply_events<BWD>(tab, q0, q1, 0, a1, o + F);
            a2 = apply_events<BWD>(tab, q0, q1, 0, a2, o + 2 * F);
            const float4 y = make_float4(a0, a1, a2, 0.f);          // one 16-B store per copy: 32 banks serve 8 lanes
            *reinterpret_cast<float4 *>(xb + 4 * r) = y;
            *reinterpret_cast<float4 *>(xb + 4 * rB) = y;
        }
        __syncthreads();
        float *sw = xa; xa = xb; xb = sw;
    }
    chain_epilogue<BWD, NT>(xa, slab, OUT, gs, brows, 3, 4, F);
}

template <int CP> struct XVec;
template <> struct XVec<1> { float v[1];
    __device__ __forceinline__ void load(const float *p) { v[0] = p[0]; }
    __device__ __forceinline__ void store(float *p) const { p[0] = v[0]; } };
template <> struct XVec<2> { float v[2];
    __device__ __forceinline__ void load(const float *p) { const float2 t = *reinterpret_cast<const float2 *>(p); v[0] = t.x; v[1] = t.y; }
    __device__ __forceinline__ void store(float *p) const { *reinterpret_cast<float2 *>(p) = make_float2(v[0], v[1]); } };
template <> struct XVec<4> { float v[4];
    __device__ __forceinline__ void load(const float *p) { const float4 t = *reinterpret_cast<const float4 *>(p); v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w; }
    __device__ __forceinline__ void store(float *p) const { *reinterpret_cast<float4 *>(p) = make_float4(v[0], v[1], v[2], v[3]); } };
template <> struct XVec<8> { float v[8];
    __device__ __forceinline__ void load(const float *p) {
        const float4 t = *reinterpret_cast<const float4 *>(p), u = *reinterpret_cast<const float4 *>(p + 4);
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w; v[4] = u.x; v[5] = u.y; v[6] = u.z; v[7] = u.w; }
    __device__ __forceinline__ void store(float *p) const {
        *reinterpret_cast<float4 *>(p) = make_float4(v[0], v[1], v[2], v[3]);
        *reinterpret_cast<float4 *>(p + 4) = make_float4(v[4], v[5], v[6], v[7]); } };

template <int C, int DMAX, int NT, bool BWD>
__global__ void __launch_bounds__(NT, 4)          // <= 128 VGPRs: two 512-thread (four 256-thread) workgroups per CU
k_chain_csr_sell(const int *__restrict__ rowptr, const int *__restrict__ col, const float *__restrict__ val,
                 const int *__restrict__ part_ptr, SellDev sd, const float *__restrict__ X, float *__restrict__ OUT,
                 int T, EvTable tab, int F)
{
    static_assert(DMAX % 2 == 0, "DMAX must be even (packed column offsets)");
    constexpr int CP = C == 1 ? 1 : (C == 2 ? 2 : (C <= 4 ? 4 : 8));
    extern __shared__ __align__(16) float smem[];
    const int gs = part_ptr[blockIdx.x], rows = part_ptr[blockIdx.x + 1] - gs;
    if (rows > NT) return;
    const int R = sell_padR(rows);
    const int nv = 2 * R * CP;                    // two copies, R is a multiple of 16
    float *xa = smem, *xb = smem + nv, *slab = smem + 2 * nv;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int s0 = sd.part_slice_ptr[blockIdx.x], ns = sd.part_slice_ptr[blockIdx.x + 1] - s0;
    const unsigned zoff = (unsigned)rows * (unsigned)CP;

    float v[DMAX];
    unsigned lc[DMAX / 2];
    int r = -1, e0 = 0, deg = 0, w = 0;
#pragma unroll
    for (int d = 0; d < DMAX; ++d) v[d] = 0.f;
#pragma unroll
    for (int q = 0; q < DMAX / 2; ++q) lc[q] = zoff | (zoff << 16);
    if (wave < ns) {
        const int s = s0 + wave;
        const int doff = sd.slice_doff[s], poff = sd.slice_poff[s];
        w = __builtin_amdgcn_readfirstlane(sd.slice_doff[s + 1] - doff);
        r = sd.perm[s * 64 + lane];
        if (r >= 0) { e0 = rowptr[gs + r]; deg = rowptr[gs + r + 1] - e0; }
#pragma unroll
        for (int d = 0; d < DMAX; ++d)
            if (d < w) v[d] = sd.val[(int64_t)(doff + d) * 64 + lane];
#pragma unroll
        for (int j2 = 0; j2 < DMAX / 2; ++j2)
            if (2 * j2 < w) {
                const unsigned u = sd.lc[(int64_t)(poff + j2) * 64 + lane];
                lc[j2] = ((u & 0xffffu) * (unsigned)CP) | (((u >> 16) * (unsigned)CP) << 16);
            }
    }
    const bool own = r >= 0;

    int p = chain_prologue<BWD, NT>(xa, xb, slab, X, gs, rows, R, C, CP, F, T, tab);
    for (int t = 1; t <= T; ++t) {
        int q0, q1;
        step_range<BWD>(tab, BWD ? T - t : t, p, q0, q1);
        float acc[C];
#pragma unroll
        for (int c = 0; c < C; ++c) acc[c] = 0.f;
#pragma unroll
        for (int d0 = 0; d0 < DMAX; d0 += SLOT_GROUP) {
            if (d0 < w) {
                XVec<CP> xv[SLOT_GROUP];
#pragma unroll
                for (int g = 0; g < SLOT_GROUP; ++g) {
                    const int d = d0 + g;
                    if (d < DMAX) xv[g].load(xa + ((d & 1) ? (lc[d / 2] >> 16) : (lc[d / 2] & 0xffffu)));
                }
#pragma unroll
                for (int g = 0; g < SLOT_GROUP; ++g) {
                    const int d = d0 + g;
                    if (d < DMAX) {
#pragma unroll
                        for (int c = 0; c < C; ++c) acc[c] = fmaf(v[d], xv[g].v[c], acc[c]);
                        if (C % 4 == 3) asm volatile("" :: "v"(xv[g].v[CP - 1]));   // 12-B tail: keep it a ds_read_b128, not b96
                    }
                }
            }
        }
        if (own) {
            for (int e = e0 + sd.dmax; e < e0 + deg; ++e) {
                const float vv = val[e];
                const float *x = xa + (col[e] - gs) * CP;
#pragma unroll
                for (int c = 0; c < C; ++c) acc[c] = fmaf(vv, x[c], acc[c]);
            }
            const int rB = R + sell_copyB(r, R);
            XVec<CP> y;
#pragma unroll
            for (int c = 0; c < CP; ++c) y.v[c] = 0.f;
#pragma unroll
            for (int c = 0; c < C; ++c) y.v[c] = apply_events<BWD>(tab, q0, q1, c, acc[c], slab + (r * C + c) * F);
            y.store(xb + r * CP); y.store(xb + rB * CP);       // whole padded rows: one wide store per copy
        }
        __syncthreads();
        float *sw = xa; xa = xb; xb = sw;
    }
    chain_epilogue<BWD, NT>(xa, slab, OUT, gs, rows, C, CP, F);
}

// ------------------------------------------------------------------ pack-time slot planning
// One wave per slice.  For every column k of the slice (one ds_read_b128 wave-instruction of the step loop) the 16
// lanes of each hardware lane group ({0-3,12-15,20-27}, {4-11,16-19,28-31}, +32) take turns: a lane picks, among its
// row's entries not yet placed, the (entry, copy) whose 16-byte bank slot is least loaded so far in this column.
// Output per kept entry e: slot_of_e[e] = its column k, lc_of_e[e] = the LDS row index it reads (copy A: local column,
// copy B: R + copyB(local column)); entries beyond dmax keep slot -1 (streamed tail).
__global__ void __launch_bounds__(64)
k_sell_plan(const int *__restrict__ rowptr, const int *__restrict__ col, const int *__restrict__ part_ptr,
            const int *__restrict__ slice_part, const int *__restrict__ perm, int dmax,
            int *__restrict__ slot_of_e, int *__restrict__ lc_of_e)
{
    __shared__ int cols[64][33];
    __shared__ volatile unsigned long long load[4];
    const int s = blockIdx.x, lane = threadIdx.x;
    const int part = slice_part[s];
    const int gs = part_ptr[part], rows = part_ptr[part + 1] - gs;
    const int R = sell_padR(rows);
    const int r = perm[(int64_t)s * 64 + lane];
    int e0 = 0, cnt = 0;
    if (r >= 0) { e0 = rowptr[gs + r]; cnt = rowptr[gs + r + 1] - e0; cnt = cnt < dmax ? cnt : dmax; }
    for (int j = 0; j < cnt; ++j) cols[lane][j] = col[e0 + j] - gs;
    unsigned rem = cnt >= 32 ? 0xffffffffu : ((1u << cnt) - 1u);
    const int w = wave_max_i(cnt);
    const int l32 = lane & 31;
    const bool g0 = l32 < 4 || (l32 >= 12 && l32 < 16) || (l32 >= 20 && l32 < 28);
    const int g = (lane >> 5) * 2 + (g0 ? 0 : 1);
    int turn = 0;                                  // rank of this lane inside its group, ascending lane order
    for (int q = 0; q < l32; ++q) {
        const bool q0 = q < 4 || (q >= 12 && q < 16) || (q >= 20 && q < 28);
        turn += (q0 == g0);
    }
    const unsigned long long gmask32 = g0 ? 0x0FF0F00FULL : 0xF00F0FF0ULL;
    const unsigned long long gmask = (lane >> 5) ? (gmask32 << 32) : gmask32;
    const int zslot = rows & 15;
    for (int k = 0; k < w; ++k) {
        const bool padded = k >= cnt;
        const unsigned long long pm = __ballot(padded);
        if (turn == 0) load[g] = (pm & gmask) ? (1ULL << (4 * zslot)) : 0ULL;
        for (int t = 0; t < 16; ++t) {
            if (turn == t && !padded) {
                unsigned long long L = load[g];
                int best = 99, bj = 0, bpos = 0;
                unsigned m = rem;
                while (m) {
                    const int j = __ffs(m) - 1;
                    m &= m - 1;
                    const int a = cols[lane][j];
                    const int b = R + sell_copyB(a, R);
                    const int cA = (int)((L >> (4 * (a & 15))) & 15ULL), cB = (int)((L >> (4 * (b & 15))) & 15ULL);
                    if (cA < best) { best = cA; bj = j; bpos = a; }
                    if (cB < best) { best = cB; bj = j; bpos = b; }
                    if (best == 0) break;
                }
                if (((L >> (4 * (bpos & 15))) & 15ULL) < 15ULL) L += 1ULL << (4 * (bpos & 15));
                load[g] = L;
                slot_of_e[e0 + bj] = k;
                lc_of_e[e0 + bj] = bpos;
                rem &= ~(1u << bj);
            }
        }
    }
}

// ------------------------------------------------------------------ orders 0/1/2 epilogue (ESCGNNModular._scatter)
__global__ void __launch_bounds__(256)
k_scatter_orders(const float *__restrict__ x0, const float *__restrict__ W1, int64_t rows, int F, int Wtot,
                 float *__restrict__ W)
{
    // one thread per output element; triu pair (i,j) of index q recovered by a short scan
    const int64_t total = rows * Wtot;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int64_t r = idx / Wtot; const int w = (int)(idx % Wtot);
        float v;
        if (w == 0) v = x0[r];
        else if (w <= F) v = W1[r * F + (w - 1)];
        else {
            int q = w - 1 - F, i = 0, len = F - 1;
            while (q >= len) { q -= len; ++i; --len; }
            v = W1[r * F + i] * W1[r * F + i + 1 + q];
        }
        W[idx] = v;
    }
}

__global__ void __launch_bounds__(256)
k_scatter_orders_bwd(const float *__restrict__ W1, const float *__restrict__ dW, int64_t rows, int F, int Wtot,
                     int second, float *__restrict__ dx0, float *__restrict__ dW1)
{
    const int64_t total = rows * F;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int64_t r = idx / F; const int f = (int)(idx % F);
        const float *d = dW + r * Wtot; const float *w1 = W1 + r * F;
        if (f == 0) dx0[r] = d[0];
        float g = d[1 + f];
        if (second) {
            int q = 1 + F;
            for (int i = 0; i < F; ++i)
                for (int j = i + 1; j < F; ++j, ++q) {
                    if (i == f) g += d[q] * w1[j];
                    else if (j == f) g += d[q] * w1[i];
                }
        }
        dW1[idx] = g;
    }
}

// ------------------------------------------------------------------ host side
static inline hipStream_t S(void *s) { return (hipStream_t)s; }

static int grid_for(int64_t work_items, int per_block)
{
    int64_t b = (work_items + per_block - 1) / per_block;
    if (b < 1) b = 1;
    const int64_t cap = 256 * 16;          // 256 CUs x 16 blocks, grid-stride beyond
    return (int)(b < cap ? b : cap);
}

static int pick_G(int64_t n, int64_t nnz)
{
    const double mean = n > 0 ? (double)nnz / (double)n : 0.0;
    return mean <= 6.0 ? 4 : (mean <= 24.0 ? 8 : 16);
}
static int pick_g(int64_t nb, int64_t nnzb)
{
    const double mean = nb > 0 ? (double)nnzb / (double)nb : 0.0;
    return mean <= 4.5 ? 3 : 7;
}

template <int C, int MODE>
static void launch_csr_step_G(int G, int grid_rows_hint, const int *rp, const int *ci, const float *v, const float *xin, int ldx,
                              float *y, int ldy, int64_t n, const StepEv &ev, float *OUT, int64_t ld_row, int F, hipStream_t st)
{
    (void)grid_rows_hint;
    if (G == 4)      hipLaunchKernelGGL((k_csr_step<C, 4, MODE>), dim3(grid_for(n, 64)), dim3(256), 0, st, rp, ci, v, xin, ldx, y, ldy, n, ev, OUT, ld_row, F);
    else if (G == 8) hipLaunchKernelGGL((k_csr_step<C, 8, MODE>), dim3(grid_for(n, 32)), dim3(256), 0, st, rp, ci, v, xin, ldx, y, ldy, n, ev, OUT, ld_row, F);
    else             hipLaunchKernelGGL((k_csr_step<C, 16, MODE>), dim3(grid_for(n, 16)), dim3(256), 0, st, rp, ci, v, xin, ldx, y, ldy, n, ev, OUT, ld_row, F);
}

template <int MODE>
static int launch_csr_step(int C, int G, const int *rp, const int *ci, const float *v, const float *xin, int ldx,
                           float *y, int ldy, int64_t n, const StepEv &ev, float *OUT, int64_t ld_row, int F, hipStream_t st)
{
    switch (C) {
#define CASE(c) case c: launch_csr_step_G<c, MODE>(G, 0, rp, ci, v, xin, ldx, y, ldy, n, ev, OUT, ld_row, F, st); break;
        CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(7) CASE(8)
#undef CASE
        default: return fail(ESCAT_ERANGE, "C=%d outside [1,%d]", C, ESCAT_MAX_CHANNELS);
    }
    return ESCAT_OK;
}

template <int MODE>
static void launch_bsr3_step(int g, const int *rp, const int *ci, const float *blk, const float *xin, float *y,
                             int64_t nb, const StepEv &ev, float *OUT, int F, hipStream_t st)
{
    if (g == 3) hipLaunchKernelGGL((k_bsr3_step<3, MODE>), dim3(grid_for(nb, 4 * 7)), dim3(256), 0, st, rp, ci, blk, xin, y, nb, ev, OUT, F);
    else        hipLaunchKernelGGL((k_bsr3_step<7, MODE>), dim3(grid_for(nb, 4 * 3)), dim3(256), 0, st, rp, ci, blk, xin, y, nb, ev, OUT, F);
}

static int check_csr(const void *rp, const void *ci, const void *v, int64_t n, int64_t nnz)
{
    if (!rp || (nnz > 0 && (!ci || !v))) return fail(ESCAT_EINVAL, "operator pointer is NULL");
    if (n < 0 || nnz < 0) return fail(ESCAT_EINVAL, "negative size");
    if (n >= (int64_t)1 << 31 || nnz >= (int64_t)1 << 31) return fail(ESCAT_ERANGE, "n or nnz >= 2^31 (int32 indices)");
    return ESCAT_OK;
}

extern "C" int escat_csr_spmm_f32(const int32_t *rowptr, const int32_t *col, const float *val, int64_t n, int64_t nnz,
                                  const float *X, int ldx, float *Y, int ldy, int C, void *stream)
{
    int rc = check_csr(rowptr, col, val, n, nnz); if (rc) return rc;
    if (!X || !Y) return fail(ESCAT_EINVAL, "X or Y is NULL");
    if (ldx < C || ldy < C) return fail(ESCAT_EINVAL, "leading dimension smaller than C");
    if (n == 0) return ESCAT_OK;
    StepEv ev; ev.n = 0;
    rc = launch_csr_step<0>(C, pick_G(n, nnz), rowptr, col, val, X, ldx, Y, ldy, n, ev, nullptr, 0, 0, S(stream));
    if (rc) return rc;
    CHECK_LAUNCH("escat_csr_spmm_f32");
    return ESCAT_OK;
}

extern "C" int escat_bsr3_spmv_f32(const int32_t *rowptr, const int32_t *col, const float *blk, int64_t nb, int64_t nnzb,
                                   const float *x, float *y, void *stream)
{
    int rc = check_csr(rowptr, col, blk, nb, nnzb); if (rc) return rc;
    if (3 * nb >= (int64_t)1 << 31) return fail(ESCAT_ERANGE, "3*nb >= 2^31");
    if (!x || !y) return fail(ESCAT_EINVAL, "x or y is NULL");
    if (nb == 0) return ESCAT_OK;
    StepEv ev; ev.n = 0;
    launch_bsr3_step<0>(pick_g(nb, nnzb), rowptr, col, blk, x, y, nb, ev, nullptr, 0, S(stream));
    CHECK_LAUNCH("escat_bsr3_spmv_f32");
    return ESCAT_OK;
}

extern "C" size_t escat_chain_workspace_bytes(int64_t n_rows, int C)
{
    if (n_rows < 0 || C < 1) return 0;
    const int cc = C < ESCAT_MAX_CHANNELS ? C : ESCAT_MAX_CHANNELS;
    // + 1/3 for the float4-per-node vectors of the packed BSR-3 path, + alignment slack
    return (size_t)2 * (size_t)n_rows * (size_t)cc * sizeof(float) * 4 / 3 + 64;
}

#define LDS_BUDGET (160 * 1024)

template <typename K> static int set_lds(K kern, size_t bytes)
{
    if (bytes > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e != hipSuccess) return fail(ESCAT_ELAUNCH, "hipFuncSetAttribute(%zu B LDS): %s", bytes, hipGetErrorString(e));
    }
    return ESCAT_OK;
}

template <int C, bool BWD>
static int launch_chain_csr_G(int G, const int *rp, const int *ci, const float *v, const int *pp, int n_parts, size_t lds,
                              const float *X, float *OUT, int T, const EvTable &tab, int F, hipStream_t st)
{
    int rc;
#define GO(GG)                                                                                         \
    rc = set_lds(k_chain_csr<C, GG, BWD>, lds); if (rc) return rc;                                     \
    hipLaunchKernelGGL((k_chain_csr<C, GG, BWD>), dim3(n_parts), dim3(CHAIN_NT), lds, st, rp, ci, v, pp, X, OUT, T, tab, F);
    if (G == 4) { GO(4) } else if (G == 8) { GO(8) } else { GO(16) }
#undef GO
    return ESCAT_OK;
}

template <bool BWD>
static int launch_chain_csr(int C, int G, const int *rp, const int *ci, const float *v, const int *pp, int n_parts, size_t lds,
                            const float *X, float *OUT, int T, const EvTable &tab, int F, hipStream_t st)
{
    switch (C) {
#define CASE(c) case c: return launch_chain_csr_G<c, BWD>(G, rp, ci, v, pp, n_parts, lds, X, OUT, T, tab, F, st);
        CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(7) CASE(8)
#undef CASE
    }
    return fail(ESCAT_ERANGE, "C=%d outside [1,%d]", C, ESCAT_MAX_CHANNELS);
}

// ---- register-resident launchers
static int chain_path_override()
{
    // debugging / A-B aid: ESCAT_CHAIN_PATH = reg | stream | multi | multi_packed forces a path (default: best that fits)
    const char *e = getenv("ESCAT_CHAIN_PATH");
    if (!e) return 0;
    if (!strcmp(e, "reg")) return 1;
    if (!strcmp(e, "stream")) return 2;
    if (!strcmp(e, "multi")) return 3;
    if (!strcmp(e, "multi_unpacked") || !strcmp(e, "multi_packed")) return 3;
    return 0;
}

#define LAUNCH_SELL(KERN, NTHREADS)                                                                    \
    do { int rc_ = set_lds(KERN, lds); if (rc_) return rc_;                                            \
         hipLaunchKernelGGL((KERN), dim3(n_parts), dim3(NTHREADS), lds, st, rp, ci, v, pp, sd, IN, RES, T, tab, F); } while (0)

static int sell_to_dev(const escat_sell *sell, SellDev *sd)
{
    if (!sell->val || !sell->lc || !sell->perm || !sell->slice_doff || !sell->slice_poff || !sell->part_slice_ptr)
        return fail(ESCAT_EINVAL, "escat_sell has a NULL member");
    if (((uintptr_t)sell->val & 15) != 0) return fail(ESCAT_EINVAL, "escat_sell.val must be 16-byte aligned");
    sd->val = sell->val; sd->lc = sell->lc; sd->perm = sell->perm; sd->slice_doff = sell->slice_doff;
    sd->slice_poff = sell->slice_poff; sd->part_slice_ptr = sell->part_slice_ptr; sd->dmax = sell->dmax;
    return ESCAT_OK;
}

template <int DMAX, bool BWD>
static int launch_bsr3_sell_d(int NT, const int *rp, const int *ci, const float *v, const int *pp, int n_parts, size_t lds,
                              const SellDev &sd, const float *IN, float *RES, int T, const EvTable &tab, int F, hipStream_t st)
{
    if (NT <= 256) LAUNCH_SELL((k_chain_bsr3_sell<DMAX, 256, BWD>), 256);
    else           LAUNCH_SELL((k_chain_bsr3_sell<DMAX, 512, BWD>), 512);
    return ESCAT_OK;
}
template <bool BWD>
static int launch_bsr3_sell(int NT, const int *rp, const int *ci, const float *v, const int *pp, int n_parts,
                            size_t lds, const SellDev &sd, const float *IN, float *RES, int T, const EvTable &tab, int F,
                            hipStream_t st)
{
    switch (sd.dmax) {
        case 8:  return launch_bsr3_sell_d<8, BWD>(NT, rp, ci, v, pp, n_parts, lds, sd, IN, RES, T, tab, F, st);
        case 12: return launch_bsr3_sell_d<12, BWD>(NT, rp, ci, v, pp, n_parts, lds, sd, IN, RES, T, tab, F, st);
        case 16: return launch_bsr3_sell_d<16, BWD>(NT, rp, ci, v, pp, n_parts, lds, sd, IN, RES, T, tab, F, st);
        case 22: return launch_bsr3_sell_d<22, BWD>(NT, rp, ci, v, pp, n_parts, lds, sd, IN, RES, T, tab, F, st);
    }
    return fail(ESCAT_EINVAL, "escat_sell.dmax=%d: BSR-3 packs must use 8, 12, 16 or 22", sd.dmax);
}

template <int C, int DMAX, bool BWD>
static int launch_csr_sell_d(int NT, const int *rp, const int *ci, const float *v, const int *pp, int n_parts, size_t lds,
                             const SellDev &sd, const float *IN, float *RES, int T, const EvTable &tab, int F, hipStream_t st)
{
    if (NT <= 256) LAUNCH_SELL((k_chain_csr_sell<C, DMAX, 256, BWD>), 256);
    else           LAUNCH_SELL((k_chain_csr_sell<C, DMAX, 512, BWD>), 512);
    return ESCAT_OK;
}
template <int C, bool BWD>
static int launch_csr_sell_c(int NT, const int *rp, const int *ci, const float *v, const int *pp, int n_parts,
                             size_t lds, const SellDev &sd, const float *IN, float *RES, int T, const EvTable &tab, int F,
                             hipStream_t st)
{
    switch (sd.dmax) {
        case 8:  return launch_csr_sell_d<C, 8, BWD>(NT, rp, ci, v, pp, n_parts, lds, sd, IN, RES, T, tab, F, st);
        case 16: return launch_csr_sell_d<C, 16, BWD>(NT, rp, ci, v, pp, n_parts, lds, sd, IN, RES, T, tab, F, st);
        case 24: return launch_csr_sell_d<C, 24, BWD>(NT, rp, ci, v, pp, n_parts, lds, sd, IN, RES, T, tab, F, st);
        case 32: return launch_csr_sell_d<C, 32, BWD>(NT, rp, ci, v, pp, n_parts, lds, sd, IN, RES, T, tab, F, st);
    }
    return fail(ESCAT_EINVAL, "escat_sell.dmax=%d: CSR packs must use 8, 16, 24 or 32", sd.dmax);
}
template <bool BWD>
static int launch_csr_sell(int C, int NT, const int *rp, const int *ci, const float *v, const int *pp,
                           int n_parts, size_t lds, const SellDev &sd, const float *IN, float *RES, int T, const EvTable &tab,
                           int F, hipStream_t st)
{
    switch (C) {
#define CASE(c) case c: return launch_csr_sell_c<c, BWD>(NT, rp, ci, v, pp, n_parts, lds, sd, IN, RES, T, tab, F, st);
        CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(7) CASE(8)
#undef CASE
    }
    return fail(ESCAT_ERANGE, "C=%d outside [1,%d]", C, ESCAT_MAX_CHANNELS);
}

// shared driver for the scalar chain, forward (BWD=false: IN = X, RES = OUT) and backward (IN = G, RES = dX)
template <bool BWD>
static int chain_csr(const int32_t *rowptr, const int32_t *col, const float *val, int64_t n, int64_t nnz,
                     const float *IN, int C, int T, const escat_event *events, int n_events, int F, float *RES,
                     const int32_t *part_ptr, int n_parts, int max_part_rows, const escat_sell *sell,
                     void *workspace, size_t workspace_bytes, void *stream)
{
    int rc = check_csr(rowptr, col, val, n, nnz); if (rc) return rc;
    if (!IN || !RES) return fail(ESCAT_EINVAL, "input or output pointer is NULL");
    if (C < 1 || C > ESCAT_MAX_CHANNELS) return fail(ESCAT_ERANGE, "C=%d outside [1,%d]", C, ESCAT_MAX_CHANNELS);
    if (T < 0 || T > 4096) return fail(ESCAT_ERANGE, "T=%d outside [0,4096]", T);
    EvTable tab;
    rc = build_table(events, n_events, T, F, C, &tab); if (rc) return rc;
    if (n == 0) return ESCAT_OK;
    hipStream_t st = S(stream);
    const int G = pick_G(n, nnz);

    const int force = chain_path_override();
    if (part_ptr && n_parts > 0 && max_part_rows > 0 && force < 3) {
        const size_t lds = (size_t)max_part_rows * C * (2 + F) * sizeof(float);
        if (lds <= LDS_BUDGET) {
            const int CP = C == 1 ? 1 : (C == 2 ? 2 : (C <= 4 ? 4 : 8));
            const size_t nv = (size_t)2 * sell_padR(max_part_rows) * CP;
            const size_t lds_s = (2 * nv + (size_t)max_part_rows * C * F) * sizeof(float);
            if (max_part_rows <= 512 && sell && force != 2 && lds_s <= LDS_BUDGET) {
                SellDev sd;
                rc = sell_to_dev(sell, &sd); if (rc) return rc;
                rc = launch_csr_sell<BWD>(C, max_part_rows, rowptr, col, val, part_ptr, n_parts, lds_s, sd, IN, RES, T, tab, F, st);
            } else
                rc = launch_chain_csr<BWD>(C, G, rowptr, col, val, part_ptr, n_parts, lds, IN, RES, T, tab, F, st);
            if (rc) return rc;
            CHECK_LAUNCH("k_chain_csr");
            return ESCAT_OK;
        }
    }
    // multi-launch path
    const size_t need = (size_t)2 * n * C * sizeof(float);
    if (T > 0 && (!workspace || workspace_bytes < need))
        return fail(ESCAT_EWORKSPACE, "workspace %zu B < %zu B needed", workspace_bytes, need);
    float *wa = (float *)workspace, *wb = wa ? wa + n * C : nullptr;
    const int64_t ld_row = (int64_t)C * F;
    StepEv se;
    if (!BWD) {
        rc = step_events(tab, 0, &se); if (rc) return rc;
        if (se.n) hipLaunchKernelGGL((k_events_only<1>), dim3(grid_for(n * C, 256)), dim3(256), 0, st,
                                     const_cast<float *>(IN), C, n, C, se, RES, ld_row, F, 0);
        const float *xin = IN;
        for (int t = 1; t <= T; ++t) {
            rc = step_events(tab, t, &se); if (rc) return rc;
            float *y = (t & 1) ? wa : wb;
            rc = launch_csr_step<1>(C, G, rowptr, col, val, xin, C, y, C, n, se, RES, ld_row, F, st); if (rc) return rc;
            xin = y;
        }
    } else {
        float *Gm = const_cast<float *>(IN);
        rc = step_events(tab, T, &se); if (rc) return rc;
        float *acc = T == 0 ? RES : wa;
        hipLaunchKernelGGL((k_events_only<2>), dim3(grid_for(n * C, 256)), dim3(256), 0, st, acc, C, n, C, se, Gm, ld_row, F, 0);
        const float *xin = acc;
        for (int t = T; t >= 1; --t) {
            rc = step_events(tab, t - 1, &se); if (rc) return rc;
            float *y = (t == 1) ? RES : (((T - t) & 1) ? wa : wb);
            rc = launch_csr_step<2>(C, G, rowptr, col, val, xin, C, y, C, n, se, Gm, ld_row, F, st); if (rc) return rc;
            xin = y;
        }
        if (T == 0) { /* answer already in RES */ }
    }
    CHECK_LAUNCH("k_csr_step chain");
    return ESCAT_OK;
}

extern "C" int escat_chain_csr_f32(const int32_t *rowptr, const int32_t *col, const float *val, int64_t n, int64_t nnz,
                                   const float *X, int C, int T, const escat_event *events, int n_events, int F, float *OUT,
                                   const int32_t *part_ptr, int n_parts, int max_part_rows, const escat_sell *sell,
                                   void *workspace, size_t workspace_bytes, void *stream)
{
    return chain_csr<false>(rowptr, col, val, n, nnz, X, C, T, events, n_events, F, OUT, part_ptr, n_parts, max_part_rows, sell,
                            workspace, workspace_bytes, stream);
}

extern "C" int escat_chain_csr_bwd_f32(const int32_t *rowptr_t, const int32_t *col_t, const float *val_t, int64_t n, int64_t nnz,
                                       const float *G, int C, int T, const escat_event *events, int n_events, int F, float *dX,
                                       const int32_t *part_ptr, int n_parts, int max_part_rows, const escat_sell *sell,
                                       void *workspace, size_t workspace_bytes, void *stream)
{
    return chain_csr<true>(rowptr_t, col_t, val_t, n, nnz, G, C, T, events, n_events, F, dX, part_ptr, n_parts, max_part_rows, sell,
                           workspace, workspace_bytes, stream);
}

template <bool BWD>
static int chain_bsr3(const int32_t *rowptr, const int32_t *col, const float *blk, int64_t nb, int64_t nnzb,
                      const float *IN, int T, const escat_event *events, int n_events, int F, float *RES,
                      const int32_t *part_ptr, int n_parts, int max_part_rows, const escat_sell *sell,
                      void *workspace, size_t workspace_bytes, void *stream)
{
    int rc = check_csr(rowptr, col, blk, nb, nnzb); if (rc) return rc;
    if (3 * nb >= (int64_t)1 << 31) return fail(ESCAT_ERANGE, "3*nb >= 2^31");
    if (!IN || !RES) return fail(ESCAT_EINVAL, "input or output pointer is NULL");
    if (T < 0 || T > 4096) return fail(ESCAT_ERANGE, "T=%d outside [0,4096]", T);
    EvTable tab;
    rc = build_table(events, n_events, T, F, 1, &tab); if (rc) return rc;
    if (nb == 0) return ESCAT_OK;
    hipStream_t st = S(stream);
    const int g = pick_g(nb, nnzb);
    const int64_t n = 3 * nb;

    const int force = chain_path_override();
    if (part_ptr && n_parts > 0 && max_part_rows > 0 && force < 3) {
        const size_t lds = (size_t)max_part_rows * 3 * (2 + F) * sizeof(float);
        const size_t lds_s = ((size_t)2 * 2 * sell_padR(max_part_rows) * 4 + (size_t)max_part_rows * 3 * F) * sizeof(float);
        if (lds <= LDS_BUDGET && lds_s <= LDS_BUDGET && max_part_rows <= 512 && sell && force != 2) {
            SellDev sd;
            rc = sell_to_dev(sell, &sd); if (rc) return rc;
            rc = launch_bsr3_sell<BWD>(max_part_rows, rowptr, col, blk, part_ptr, n_parts, lds_s, sd, IN, RES, T, tab, F, st);
            if (rc) return rc;
            CHECK_LAUNCH("k_chain_bsr3_sell");
            return ESCAT_OK;
        }
        if (lds <= LDS_BUDGET) {
#define GO(gg)                                                                                          \
            rc = set_lds(k_chain_bsr3<gg, BWD>, lds); if (rc) return rc;                                \
            hipLaunchKernelGGL((k_chain_bsr3<gg, BWD>), dim3(n_parts), dim3(CHAIN_NT), lds, st, rowptr, col, blk, part_ptr, IN, RES, T, tab, F);
            if (g == 3) { GO(3) } else { GO(7) }
#undef GO
            CHECK_LAUNCH("k_chain_bsr3");
            return ESCAT_OK;
        }
    }
    if (sell && sell->col32 && T > 0) {
        // packed multi-launch path: float4-per-node intermediate vectors
        if (!sell->val || !sell->perm || !sell->slice_doff) return fail(ESCAT_EINVAL, "escat_sell has a NULL member");
        if (((uintptr_t)sell->val & 15) != 0) return fail(ESCAT_EINVAL, "escat_sell.val must be 16-byte aligned");
        const size_t needp = (size_t)2 * nb * 4 * sizeof(float);
        if (!workspace || workspace_bytes < needp || ((uintptr_t)workspace & 15) != 0)
            return fail(ESCAT_EWORKSPACE, "workspace %zu B < %zu B needed (16-byte aligned)", workspace_bytes, needp);
        SellG sg = {sell->val, sell->col32, sell->perm, sell->slice_doff};
        const int ns = sell->n_slices;
        const dim3 grid((ns + 3) / 4), blk256(256);
        float *pa = (float *)workspace, *pb = pa + nb * 4;
        StepEv se;
        if (!BWD) {
            rc = step_events(tab, 0, &se); if (rc) return rc;
            if (se.n) hipLaunchKernelGGL((k_events_only<1>), dim3(grid_for(n, 256)), dim3(256), 0, st,
                                         const_cast<float *>(IN), 1, n, 1, se, RES, (int64_t)F, F, 0);
            const float *xin = IN;
            for (int t = 1; t <= T; ++t) {
                rc = step_events(tab, t, &se); if (rc) return rc;
                float *y = (t & 1) ? pa : pb;
                if (t == 1) hipLaunchKernelGGL((k_bsr3_step_sell<1, true, false>), grid, blk256, 0, st, sg, ns, xin, y, se, RES, F);
                else        hipLaunchKernelGGL((k_bsr3_step_sell<1, false, false>), grid, blk256, 0, st, sg, ns, xin, y, se, RES, F);
                xin = y;
            }
        } else {
            float *Gm = const_cast<float *>(IN);
            rc = step_events(tab, T, &se); if (rc) return rc;
            hipLaunchKernelGGL((k_events_only<2>), dim3(grid_for(n, 256)), dim3(256), 0, st, pa, 1, n, 1, se, Gm, (int64_t)F, F, 1);
            const float *xin = pa;
            for (int t = T; t >= 1; --t) {
                rc = step_events(tab, t - 1, &se); if (rc) return rc;
                float *y = (t == 1) ? RES : (((T - t) & 1) ? pa : pb);
                if (t == 1) hipLaunchKernelGGL((k_bsr3_step_sell<2, false, true>), grid, blk256, 0, st, sg, ns, xin, y, se, Gm, F);
                else        hipLaunchKernelGGL((k_bsr3_step_sell<2, false, false>), grid, blk256, 0, st, sg, ns, xin, y, se, Gm, F);
                xin = y;
            }
        }
        CHECK_LAUNCH("k_bsr3_step_sell chain");
        return ESCAT_OK;
    }
    const size_t need = (size_t)2 * n * sizeof(float);
    if (T > 0 && (!workspace || workspace_bytes < need))
        return fail(ESCAT_EWORKSPACE, "workspace %zu B < %zu B needed", workspace_bytes, need);
    float *wa = (float *)workspace, *wb = wa ? wa + n : nullptr;
    StepEv se;
    if (!BWD) {
        rc = step_events(tab, 0, &se); if (rc) return rc;
        if (se.n) hipLaunchKernelGGL((k_events_only<1>), dim3(grid_for(n, 256)), dim3(256), 0, st,
                                     const_cast<float *>(IN), 1, n, 1, se, RES, (int64_t)F, F, 0);
        const float *xin = IN;
        for (int t = 1; t <= T; ++t) {
            rc = step_events(tab, t, &se); if (rc) return rc;
            float *y = (t & 1) ? wa : wb;
            launch_bsr3_step<1>(g, rowptr, col, blk, xin, y, nb, se, RES, F, st);
            xin = y;
        }
    } else {
        float *Gm = const_cast<float *>(IN);
        rc = step_events(tab, T, &se); if (rc) return rc;
        float *acc = T == 0 ? RES : wa;
        hipLaunchKernelGGL((k_events_only<2>), dim3(grid_for(n, 256)), dim3(256), 0, st, acc, 1, n, 1, se, Gm, (int64_t)F, F, 0);
        const float *xin = acc;
        for (int t = T; t >= 1; --t) {
            rc = step_events(tab, t - 1, &se); if (rc) return rc;
            float *y = (t == 1) ? RES : (((T - t) & 1) ? wa : wb);
            launch_bsr3_step<2>(g, rowptr, col, blk, xin, y, nb, se, Gm, F, st);
            xin = y;
        }
    }
    CHECK_LAUNCH("k_bsr3_step chain");
    return ESCAT_OK;
}

extern "C" int escat_chain_bsr3_f32(const int32_t *rowptr, const int32_t *col, const float *blk, int64_t nb, int64_t nnzb,
                                    const float *x, int T, const escat_event *events, int n_events, int F, float *OUT,
                                    const int32_t *part_ptr, int n_parts, int max_part_rows, const escat_sell *sell,
                                    void *workspace, size_t workspace_bytes, void *stream)
{
    return chain_bsr3<false>(rowptr, col, blk, nb, nnzb, x, T, events, n_events, F, OUT, part_ptr, n_parts, max_part_rows, sell,
                             workspace, workspace_bytes, stream);
}

extern "C" int escat_chain_bsr3_bwd_f32(const int32_t *rowptr_t, const int32_t *col_t, const float *blk_t, int64_t nb, int64_t nnzb,
                                        const float *G, int T, const escat_event *events, int n_events, int F, float *dx,
                                        const int32_t *part_ptr, int n_parts, int max_part_rows, const escat_sell *sell,
                                        void *workspace, size_t workspace_bytes, void *stream)
{
    return chain_bsr3<true>(rowptr_t, col_t, blk_t, nb, nnzb, G, T, events, n_events, F, dx, part_ptr, n_parts, max_part_rows, sell,
                            workspace, workspace_bytes, stream);
}

extern "C" int escat_scatter_orders_f32(const float *x0, const float *W1, int64_t rows, int F, int second_order, float *W,
                                        void *stream)
{
    if (!x0 || !W1 || !W) return fail(ESCAT_EINVAL, "NULL pointer");
    if (F < 1 || F > 64) return fail(ESCAT_ERANGE, "F=%d out of range [1,64]", F);
    if (rows <= 0) return rows == 0 ? ESCAT_OK : fail(ESCAT_EINVAL, "negative rows");
    const int Wtot = 1 + F + (second_order ? F * (F - 1) / 2 : 0);
    hipLaunchKernelGGL(k_scatter_orders, dim3(grid_for(rows * Wtot, 256)), dim3(256), 0, S(stream), x0, W1, rows, F, Wtot, W);
    CHECK_LAUNCH("k_scatter_orders");
    return ESCAT_OK;
}

extern "C" int escat_scatter_orders_bwd_f32(const float *W1, const float *dW, int64_t rows, int F, int second_order,
                                            float *dx0, float *dW1, void *stream)
{
    if (!W1 || !dW || !dx0 || !dW1) return fail(ESCAT_EINVAL, "NULL pointer");
    if (F < 1 || F > 64) return fail(ESCAT_ERANGE, "F=%d out of range [1,64]", F);
    if (rows <= 0) return rows == 0 ? ESCAT_OK : fail(ESCAT_EINVAL, "negative rows");
    const int Wtot = 1 + F + (second_order ? F * (F - 1) / 2 : 0);
    hipLaunchKernelGGL(k_scatter_orders_bwd, dim3(grid_for(rows * F, 256)), dim3(256), 0, S(stream), W1, dW, rows, F, Wtot,
                       second_order, dx0, dW1);
    CHECK_LAUNCH("k_scatter_orders_bwd");
    return ESCAT_OK;
}

extern "C" int escat_sell_plan_slots(const int32_t *rowptr, const int32_t *col, const int32_t *part_ptr,
                                     const int32_t *slice_part, const int32_t *perm, int n_slices, int dmax,
                                     int32_t *slot_of_e, int32_t *lc_of_e, void *stream)
{
    if (!rowptr || !col || !part_ptr || !slice_part || !perm || !slot_of_e || !lc_of_e) return fail(ESCAT_EINVAL, "NULL pointer");
    if (dmax < 1 || dmax > 32) return fail(ESCAT_ERANGE, "dmax=%d outside [1,32]", dmax);
    if (n_slices <= 0) return n_slices == 0 ? ESCAT_OK : fail(ESCAT_EINVAL, "negative n_slices");
    hipLaunchKernelGGL(k_sell_plan, dim3(n_slices), dim3(64), 0, S(stream), rowptr, col, part_ptr, slice_part, perm, dmax,
                       slot_of_e, lc_of_e);
    CHECK_LAUNCH("k_sell_plan");
    return ESCAT_OK;
}

