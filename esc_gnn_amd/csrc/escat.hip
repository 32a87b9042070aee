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
    if (F < 1 || F > ESCAT_MAX_SLOTS) return fail(ESCAT_ERANGE, "F=%d out of range [1,%d]", F, ESCAT_MAX_SLOTS);
    static thread_local unsigned char seen[ESCAT_MAX_SLOTS * ESCAT_MAX_CHANNELS];
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
// VP = 3: one value per channel, val (nnz, 3) (the signed scalar operator of a factored vector operator, see OpCsrV3);
// VP = 4: the same operator as |value| (nnz,) + the three signs in bits 29..31 of col (8 B per nonzero instead of 16).
template <int C, int G, int MODE, int VP = 1>
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
        // four entries per lane and trip: their index/value loads go out together, then their x gathers together --
        // two memory round trips per four entries instead of per entry (the step is latency-bound otherwise).  Slots past
        // the row's end re-read the trip's first entry with a zero value.
        constexpr int U = 4;
        for (int e = rowptr[row] + lg; e < e1; e += G * U) {
            unsigned cw[U]; float vv[U][VP == 3 ? 3 : 1];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const bool ok = e + u * G < e1;
                const int eu = ok ? e + u * G : e;
                cw[u] = (unsigned)col[eu];
                if constexpr (VP == 3) {
#pragma unroll
                    for (int c = 0; c < 3; ++c) vv[u][c] = ok ? val[(int64_t)eu * 3 + c] : 0.f;
                } else vv[u][0] = ok ? val[eu] : 0.f;
            }
            float xr[U][C];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const float *x = xin + (int64_t)(VP == 4 ? (cw[u] & 0x1fffffffu) : cw[u]) * ldx;
#pragma unroll
                for (int c = 0; c < C; ++c) xr[u][c] = x[c];
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    float v;
                    if constexpr (VP == 4) v = __uint_as_float(__float_as_uint(vv[u][0]) | ((cw[u] << (2 - c)) & 0x80000000u));
                    else if constexpr (VP == 3) v = vv[u][c % 3];
                    else v = vv[u][0];
                    acc[c] += v * xr[u][c];
                }
            }
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
// Every part owns `nsl` slices (nsl = max part rows / 64, rounded up; trailing ones may be empty).  Inside a part the
// rows are sorted by length (descending) and cut into slices of 64; slice s stores its first w_s = min(longest row,
// dmax) entries slot-major,
//   CSR   val[(doff_s + d) * 64 + lane]
//   BSR-3 val[(doff_s + d) * 576 + {0: lane*4 + 0..3 | 256: lane*4 + 0..3 | 512: lane}]
//         (the 9 floats of a block as two float4 pieces and one scalar piece:
//          b00 b10 b01 b11 | b02 b12 b20 b21 | b22)
//   lc[(poff_s + d/2) * 64 + lane] = LDS row of slot d | LDS row of slot d+1 << 16, for d < w_s rounded up to 4
// so that every load of a wave is one fully coalesced 256-B or 1-KiB access, and the sort makes w_s the length of
// (nearly) every row of the slice: few idle slots.  Slots past a row's end hold value 0 and LDS row = rows of the part
// (the all-zero row), so the step loop needs no per-lane predication.
//   lane_meta[(s*64 + lane)*3 + {0,1,2}] = { row | tail_n << 10,  tail_e0,  slice word }:
//       row = the row (local to the part) the lane serves, 0x3ff = none; tail_e0 / tail_n = the row's entries beyond
//       dmax in the canonical arrays (streamed from L2 every step); slice words of lanes 0..4 = doff_s, poff_s, w_s,
//       first row of the part, rows of the part -- ONE 12-byte load per lane brings everything a wave needs to start
//       a part, with no dependent second hop.
struct SellDev {
    const float *val; const unsigned *lc; const int *lane_meta;
    const int *part_list, *part_row0;             // optional: the launch covers parts part_list[i], whose rows of the
    int dmax, nsl;                                // dense operands start at part_row0[i] (a batch drawn from a store)
    int orders;                                   // forward only: 0 = the bank; 1 = [x | bank]; 2 = [x | bank | pair products]
    int planes;                                   // CSR packs: 1 = one value per entry, 3 = one per channel (OpCsrV3)
    const float *frames;                          // OpCsrV3 only, optional: (n, 3, 3) local frames; IN and RES are then in
};                                                // global coordinates (the chain runs D Pt^t D^T instead of Pt^t)

#define SLOT_GROUP 4      // granule of the packed LDS-row pairs (slots are padded to it)
#define GATHER_AHEAD 3     // LDS gathers in flight ahead of the FMAs

// debug build only (-DESCAT_STAMPS, see tools/stamps.py): wall-clock stamps of the phases of every part
#ifdef ESCAT_STAMPS
__device__ unsigned long long *g_stamps = nullptr;
#define STAMP(part, i) do { if (g_stamps && threadIdx.x == 0) g_stamps[(size_t)(part) * 16 + (i)] = wall_clock64(); } while (0)
#define STAMPW(part, i, tid) do { if (g_stamps && threadIdx.x == (tid)) g_stamps[(size_t)(part) * 16 + (i)] = wall_clock64(); } while (0)
extern "C" int escat_debug_set_stamps(void *p) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), &p, sizeof(p)); }
#else
#define STAMP(part, i)
#define STAMPW(part, i, tid)
#endif

typedef float vf4 __attribute__((ext_vector_type(4)));
typedef float vf2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void lds_void_t;

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

// ---- the operator of one row, in registers.  Two flavours with one interface:
//   load(sd, doff, poff, w, lane)  issue the (coalesced) loads of this lane's slots
//   gather(xa, w, acc)          acc += sum over slots of value * x[LDS row]   (one step)
//   tail(...)                   entries beyond dmax, streamed
template <int DMAX> struct OpBsr3 {
    static constexpr bool FRAMES = false;
    static constexpr int C = 3, CP = 4, MINW = 1;
    // a block's 9 floats as the three pieces they are loaded in, ordered for packed fp32 math:
    //   b0 = (b00, b10, b01, b11)   b1 = (b02, b12, b20, b21)   b2 = b22       (bij = row i, column j of the block)
    // so that (y0, y1) += (b0j, b1j) * xj is one v_pk_fma_f32 per column j (xj broadcast by op_sel) and
    // (b20, b21) * (x0, x1) another: 5 VALU instructions per block instead of 9.
    vf4 b0[DMAX], b1[DMAX];
    float b2[DMAX];
    unsigned lc[(DMAX + 1) / 2];
    // Every slot register is written by exactly one UNCONDITIONAL load per part: a slot past the slice's width re-reads
    // the last real slot (same address for the whole wave: served by the cache) and is never consumed, gather() stops at
    // the width.  Written with a branch per slot ("load if d < w, else keep / zero") the compiler merges loaded and
    // kept values through register copies that need the data -- a vmcnt(0) after every slot, i.e. twenty serial HBM
    // round trips per part instead of one burst (measured with tools/stamps.py: 16 us vs the ~9 us the bytes take).
    __device__ __forceinline__ void load(const SellDev &sd, int doff, int poff, int w, int lane)
    {
        const float *vb = sd.val + (int64_t)doff * 576;
        const int wl = w > 0 ? w - 1 : 0;         // (an empty slice reads the pack's padding slot)
        const int pl = w > 0 ? ((w + SLOT_GROUP - 1) / SLOT_GROUP) * (SLOT_GROUP / 2) - 1 : 0;
#pragma unroll
        for (int d = 0; d < DMAX; ++d) {
            const int dd = d < wl ? d : wl;       // wave-uniform
            b0[d] = *reinterpret_cast<const vf4 *>(vb + dd * 576 + lane * 4);
            b1[d] = *reinterpret_cast<const vf4 *>(vb + dd * 576 + 256 + lane * 4);
            b2[d] = vb[dd * 576 + 512 + lane];
        }
#pragma unroll
        for (int j2 = 0; j2 < (DMAX + 1) / 2; ++j2)
            lc[j2] = sd.lc[(int64_t)(poff + (j2 < pl ? j2 : pl)) * 64 + lane];
    }
    // One step of this lane's row.  gather() dispatches (wave-uniform) on the slice's width, rounded up to 2, to a
    // straight-line body for exactly that many slots: the LDS gathers run GATHER_AHEAD slots ahead of the FMAs in a ring
    // of registers, so a wave always has gathers in flight (two waves per SIMD cannot hide LDS latency by themselves),
    // and without branches in the body the compiler can count them (s_waitcnt lgkmcnt(N), not 0).  The odd slot of a
    // rounded-up width re-reads a real block against the all-zero row.
    __device__ __forceinline__ unsigned row_of(int d) const { return (d & 1) ? (lc[d / 2] >> 16) : (lc[d / 2] & 0xffffu); }
    template <int W>
    __device__ __forceinline__ void gather_w(const float *xa, vf2 &a01, vf2 &t2, float &a2) const
    {
        vf4 xv[GATHER_AHEAD + 1];
#pragma unroll
        for (int d = 0; d < GATHER_AHEAD && d < W; ++d) xv[d] = *reinterpret_cast<const vf4 *>(xa + row_of(d) * 4u);   // a row is 4 floats
#pragma unroll
        for (int d = 0; d < W; ++d) {
            if (d + GATHER_AHEAD < W)
                xv[(d + GATHER_AHEAD) % (GATHER_AHEAD + 1)] = *reinterpret_cast<const vf4 *>(xa + row_of(d + GATHER_AHEAD) * 4u);
            const vf4 x = xv[d % (GATHER_AHEAD + 1)];
            a01 = __builtin_elementwise_fma(b0[d].xy, x.xx, a01);
            a01 = __builtin_elementwise_fma(b0[d].zw, x.yy, a01);
            a01 = __builtin_elementwise_fma(b1[d].xy, x.zz, a01);
            t2 = __builtin_elementwise_fma(b1[d].zw, x.xy, t2);
            a2 = fmaf(b2[d], x.z, a2);
            // keep the pad lane live: otherwise the gather is narrowed to ds_read_b96 (8 LDS cycles, 8-lane groups on
            // 32 banks) instead of ds_read_b128 (4 cycles, the 16-lane groups the pack is planned for)
            asm volatile("" :: "v"(x.w));
        }
    }
    template <int W>
    __device__ __forceinline__ void dispatch(int w2, const float *xa, vf2 &a01, vf2 &t2, float &a2) const
    {
        if constexpr (W <= DMAX) {
            if (w2 == W / 2) gather_w<W>(xa, a01, t2, a2);
            else dispatch<W + 2>(w2, xa, a01, t2, a2);
        }
    }
    __device__ __forceinline__ void gather(const float *xa, int w, float (&a)[3]) const
    {
        vf2 a01 = {0.f, 0.f}, t2 = {0.f, 0.f};    // (y0, y1) and the two-column partial sums of y2
        float a2 = 0.f;
        dispatch<2>((w + 1) >> 1, xa, a01, t2, a2);
        a[0] += a01.x; a[1] += a01.y; a[2] += a2 + (t2.x + t2.y);
    }
    static __device__ __forceinline__ void tail(const float *__restrict__ blk, const int *__restrict__ col, int e0, int n,
                                                int gs, const float *xa, float (&a)[3])
    {
        for (int e = e0; e < e0 + n; ++e) {       // rows longer than the pack: from L2, every step
            const float *bb = blk + (int64_t)e * 9;
            const float *x = xa + (col[e] - gs) * 4;
            const float x0 = x[0], x1 = x[1], x2 = x[2];
            a[0] = fmaf(bb[0], x0, a[0]); a[0] = fmaf(bb[1], x1, a[0]); a[0] = fmaf(bb[2], x2, a[0]);
            a[1] = fmaf(bb[3], x0, a[1]); a[1] = fmaf(bb[4], x1, a[1]); a[1] = fmaf(bb[5], x2, a[1]);
            a[2] = fmaf(bb[6], x0, a[2]); a[2] = fmaf(bb[7], x1, a[2]); a[2] = fmaf(bb[8], x2, a[2]);
        }
    }
};

template <int CC, int DMAX> struct OpCsr {
    static_assert(DMAX % 2 == 0, "DMAX must be even (packed LDS rows)");
    static constexpr bool FRAMES = false;
    static constexpr int C = CC, CP = CC == 1 ? 1 : (CC == 2 ? 2 : (CC <= 4 ? 4 : 8));
    static constexpr int MINW = 4;                // <= 128 VGPRs: two 512-thread (four 256-thread) workgroups per CU
    float v[DMAX];
    unsigned lc[DMAX / 2];
    __device__ __forceinline__ void load(const SellDev &sd, int doff, int poff, int w, int lane)   // see OpBsr3::load
    {
        const int wl = w > 0 ? w - 1 : 0;
        const int pl = w > 0 ? ((w + SLOT_GROUP - 1) / SLOT_GROUP) * (SLOT_GROUP / 2) - 1 : 0;
#pragma unroll
        for (int d = 0; d < DMAX; ++d) v[d] = sd.val[(int64_t)(doff + (d < wl ? d : wl)) * 64 + lane];
#pragma unroll
        for (int j2 = 0; j2 < DMAX / 2; ++j2)
            lc[j2] = sd.lc[(int64_t)(poff + (j2 < pl ? j2 : pl)) * 64 + lane];     // LDS rows, two per register
    }
    __device__ __forceinline__ unsigned row_of(int d) const { return (d & 1) ? (lc[d / 2] >> 16) : (lc[d / 2] & 0xffffu); }
    template <int W>
    __device__ __forceinline__ void gather_w(const float *xa, float (&acc)[CC]) const      // see OpBsr3::gather_w
    {
        XVec<CP> xv[GATHER_AHEAD + 1];
#pragma unroll
        for (int d = 0; d < GATHER_AHEAD && d < W; ++d) xv[d].load(xa + row_of(d) * (unsigned)CP);
#pragma unroll
        for (int d = 0; d < W; ++d) {
            if (d + GATHER_AHEAD < W) xv[(d + GATHER_AHEAD) % (GATHER_AHEAD + 1)].load(xa + row_of(d + GATHER_AHEAD) * (unsigned)CP);
            const XVec<CP> &x = xv[d % (GATHER_AHEAD + 1)];
#pragma unroll
            for (int c = 0; c < CC; ++c) acc[c] = fmaf(v[d], x.v[c], acc[c]);
            if (CC % 4 == 3) asm volatile("" :: "v"(x.v[CP - 1]));   // 12-B tail: keep it a ds_read_b128, not b96
        }
    }
    template <int W>
    __device__ __forceinline__ void dispatch(int w2, const float *xa, float (&acc)[CC]) const
    {
        if constexpr (W <= DMAX) {
            if (w2 == W / 2) gather_w<W>(xa, acc);
            else dispatch<W + 2>(w2, xa, acc);
        }
    }
    __device__ __forceinline__ void gather(const float *xa, int w, float (&acc)[CC]) const { dispatch<2>((w + 1) >> 1, xa, acc); }
    static __device__ __forceinline__ void tail(const float *__restrict__ val, const int *__restrict__ col, int e0, int n,
                                                int gs, const float *xa, float (&acc)[CC])
    {
        for (int e = e0; e < e0 + n; ++e) {
            const float vv = val[e];
            const float *x = xa + (col[e] - gs) * CP;
#pragma unroll
            for (int c = 0; c < CC; ++c) acc[c] = fmaf(vv, x[c], acc[c]);
        }
    }
};


// CSR with one value PER CHANNEL (three of them): y[i][c] = sum_j v[i,j][c] * x[j][c].  This is the vector operator in its
// factored form: Q = D Pt D^T with D = blockdiag(O_i) and Pt_c[i,j] = p_ij * sign<O_i[:,c], O_j[:,c]> (what
// process_pyg_data's column_dot alignment builds, process_pyg_data.py:306-313, :492), so Q^t = D Pt^t D^T and the chain
// runs in the local frames on P's pattern with 12 B of values per nonzero instead of 36.  Same LDS layout and gathers
// as OpCsr<3>; pack layout val[((doff_s + d) * 3 + c) * 64 + lane] (escat_sell.val_planes = 3).
template <int DMAX, bool SIGNED = false> struct OpCsrV3 {
    static_assert(DMAX % 2 == 0, "DMAX must be even (packed LDS rows)");
    static constexpr int C = 3, CP = 4;
    static constexpr int MINW = 4;                // <= 128 VGPRs (60 of them values): two 512-thread workgroups per CU
    static constexpr int AHEAD = 2;               // gathers in flight (one fewer than OpCsr: the ring costs 4 VGPRs a stage)
    static constexpr bool FRAMES = true;          // SellDev.frames != NULL: rotate in and out of the local frames in LDS
    float v[DMAX][3];
    unsigned lc[DMAX / 2];
    __device__ __forceinline__ void load(const SellDev &sd, int doff, int poff, int w, int lane)   // see OpBsr3::load
    {
        const int wl = w > 0 ? w - 1 : 0;
        const int pl = w > 0 ? ((w + SLOT_GROUP - 1) / SLOT_GROUP) * (SLOT_GROUP / 2) - 1 : 0;
#pragma unroll
        for (int j2 = 0; j2 < DMAX / 2; ++j2)
            lc[j2] = sd.lc[(int64_t)(poff + (j2 < pl ? j2 : pl)) * 64 + lane];
        if constexpr (!SIGNED) {
#pragma unroll
            for (int d = 0; d < DMAX; ++d) {
                const float *p = sd.val + (int64_t)(doff + (d < wl ? d : wl)) * 192 + lane;
                v[d][0] = p[0]; v[d][1] = p[64]; v[d][2] = p[128];
            }
        } else {          // one plane of magnitudes from HBM; the signs ride in bits 11..13 of the slot's LDS-row field
            float m[DMAX];
#pragma unroll
            for (int d = 0; d < DMAX; ++d) m[d] = sd.val[(int64_t)(doff + (d < wl ? d : wl)) * 64 + lane];
#pragma unroll
            for (int d = 0; d < DMAX; ++d) {
                const unsigned f = field(d), mb = __float_as_uint(m[d]);
#pragma unroll
                for (int c = 0; c < 3; ++c) v[d][c] = __uint_as_float(mb | ((f << (20 - c)) & 0x80000000u));
            }
        }
    }
    __device__ __forceinline__ unsigned field(int d) const { return (d & 1) ? (lc[d / 2] >> 16) : (lc[d / 2] & 0xffffu); }
    __device__ __forceinline__ unsigned row_of(int d) const { return SIGNED ? (field(d) & 0x7ffu) : field(d); }
    template <int W>
    __device__ __forceinline__ void gather_w(const float *xa, float (&acc)[3]) const
    {
        XVec<4> xv[AHEAD + 1];
#pragma unroll
        for (int d = 0; d < AHEAD && d < W; ++d) xv[d].load(xa + row_of(d) * 4u);
#pragma unroll
        for (int d = 0; d < W; ++d) {
            if (d + AHEAD < W) xv[(d + AHEAD) % (AHEAD + 1)].load(xa + row_of(d + AHEAD) * 4u);
            const XVec<4> &x = xv[d % (AHEAD + 1)];
#pragma unroll
            for (int c = 0; c < 3; ++c) acc[c] = fmaf(v[d][c], x.v[c], acc[c]);
            asm volatile("" :: "v"(x.v[3]));          // keep the gather a ds_read_b128
        }
    }
    template <int W>
    __device__ __forceinline__ void dispatch(int w2, const float *xa, float (&acc)[3]) const
    {
        if constexpr (W <= DMAX) {
            if (w2 == W / 2) gather_w<W>(xa, acc);
            else dispatch<W + 2>(w2, xa, acc);
        }
    }
    __device__ __forceinline__ void gather(const float *xa, int w, float (&acc)[3]) const { dispatch<2>((w + 1) >> 1, xa, acc); }
    static __device__ __forceinline__ void tail(const float *__restrict__ val, const int *__restrict__ col, int e0, int n,
                                                int gs, const float *xa, float (&acc)[3])
    {
        for (int e = e0; e < e0 + n; ++e) {
            if constexpr (SIGNED) {          // global arrays in the packed form of k_csr_step<.., 4>
                const unsigned cw = (unsigned)col[e], vb = __float_as_uint(val[e]);
                const float *x = xa + ((int)(cw & 0x1fffffffu) - gs) * 4;
#pragma unroll
                for (int c = 0; c < 3; ++c) acc[c] = fmaf(__uint_as_float(vb | ((cw << (2 - c)) & 0x80000000u)), x[c], acc[c]);
            } else {
                const float *x = xa + (col[e] - gs) * 4;
#pragma unroll
                for (int c = 0; c < 3; ++c) acc[c] = fmaf(val[(int64_t)e * 3 + c], x[c], acc[c]);
            }
        }
    }
};


typedef const __attribute__((address_space(1))) void glb_void_t;

// n floats global -> LDS without passing through registers (LDS-DMA, one dword per lane, 256 B per wave-instruction);
// completion is counted in vmcnt like any load.
template <int NT>
__device__ __forceinline__ void dma_in(float *dst, const float *__restrict__ src, int n)
{
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int i0 = wave * 64; i0 < n; i0 += NT) {
        const int i = i0 + lane;
        if (i < n) __builtin_amdgcn_global_load_lds((glb_void_t *)(src + i), (lds_void_t *)(dst + i0), 4, 0, 0);
    }
}

// builds the first vector of a part from its staged input (fwd: x in `inb`; bwd: the upstream-gradient slab), applies
// the step-0 events (fwd) / forms the Horner start (bwd); returns the event cursor.  Ends with a barrier.
// Fs / so: row stride of the slab and offset of slot 0 in it (Fs = F, so = 0; the fused-orders forward keeps x itself
// in front of each row's bank: Fs = F + 1, so = 1).
template <bool BWD, int NT>
__device__ __forceinline__ int chain_prologue(float *xa, float *xb, float *slab, const float *inb, int rows, int R, int C, int CP,
                                              int F, int T, const EvTable &tab, int Fs = -1, int so = 0)
{
    if (Fs < 0) Fs = F;
    const int tid = threadIdx.x;
    const int nc = rows * C;
    int p;
    if (tid < CP) { xa[rows * CP + tid] = 0.f; xb[rows * CP + tid] = 0.f; }
    if (!BWD) {
        int p1 = 0; while (p1 < tab.n && tab.e[p1].step == 0) ++p1;
        for (int i = tid; i < nc; i += NT) {
            const int c = i % C, r = i / C; const float y = inb[i];
            xa[r * CP + c] = y;
            xa[(R + sell_copyB(r, R)) * CP + c] = y;
            if (so) slab[i * Fs] = y;
            for (int k = 0; k < p1; ++k)
                if (tab.e[k].chan < 0 || tab.e[k].chan == c) emit_one(slab + i * Fs + so, y, tab.e[k].slot, tab.e[k].sign);
        }
        p = p1;
    } else {
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

// in-place rotation of a part's 3-vectors in LDS by the rows' local frames (global, L1/L2-resident: 36 B per row):
// buf[r * rs + c * cs + f] <- sum_c' O_r[c][c'] (TR: O_r[c'][c]) buf[r * rs + c' * cs + f]; rows are independent.
template <bool TR, int NT>
__device__ __forceinline__ void rotate_rows(float *buf, const float *__restrict__ fr, int rows, int F, int rs, int cs)
{
    for (int r = threadIdx.x; r < rows; r += NT) {             // one thread per row: its frame is loaded once
        const float *o = fr + r * 9;
        const float o0 = o[0], o1 = o[1], o2 = o[2], o3 = o[3], o4 = o[4], o5 = o[5], o6 = o[6], o7 = o[7], o8 = o[8];
        float *x = buf + r * rs;
        for (int f = 0; f < F; ++f) {
            const float x0 = x[f], x1 = x[cs + f], x2 = x[2 * cs + f];
            if (TR) {
                x[f]          = fmaf(o6, x2, fmaf(o3, x1, o0 * x0));
                x[cs + f]     = fmaf(o7, x2, fmaf(o4, x1, o1 * x0));
                x[2 * cs + f] = fmaf(o8, x2, fmaf(o5, x1, o2 * x0));
            } else {
                x[f]          = fmaf(o2, x2, fmaf(o1, x1, o0 * x0));
                x[cs + f]     = fmaf(o5, x2, fmaf(o4, x1, o3 * x0));
                x[2 * cs + f] = fmaf(o8, x2, fmaf(o7, x1, o6 * x0));
            }
        }
    }
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

// forward epilogue with the orders 0/1/2 of ESCGNNModular._scatter fused in (models/escgnn_modular.py:333-341): writes
// (rows*C, Wtot) = [x | bank | bank_i * bank_j, i < j in triu order] straight from the LDS slab, so the bank itself
// never goes to HBM and back.  The slab rows are [x | bank] in this mode (chain_prologue puts x there).
// One thread per row-channel expands its slab row [x | bank] to Wtot values in an LDS tile (the two vector buffers are
// dead by now), then the tile goes out as a contiguous run; no per-element index arithmetic.
template <int NT>
__device__ __forceinline__ void chain_epilogue_orders(const float *slab, float *tile, int tile_floats, float *__restrict__ RES,
                                                      int64_t first_row, int rows, int C, int F, int second)
{
    const int wtot = 1 + F + (second ? F * (F - 1) / 2 : 0), Fs = F + 1;
    const int nrc = rows * C, per = tile_floats / wtot;
    float *og = RES + first_row * C * wtot;
    for (int c0 = 0; c0 < nrc; c0 += per) {
        const int n = nrc - c0 < per ? nrc - c0 : per;
        for (int rc = threadIdx.x; rc < n; rc += NT) {
            const float *srow = slab + (c0 + rc) * Fs;
            float *o = tile + rc * wtot;
            for (int f = 0; f <= F; ++f) o[f] = srow[f];
            if (second) {
                int q = 1 + F;
                for (int i = 0; i < F; ++i) {
                    const float a = srow[1 + i];
                    for (int j = i + 1; j < F; ++j) o[q++] = a * srow[1 + j];
                }
            }
        }
        __syncthreads();
        float *dst = og + (int64_t)c0 * wtot;
        for (int i = threadIdx.x; i < n * wtot; i += NT) dst[i] = tile[i];
        __syncthreads();
    }
}

// ---- the register-resident chain kernel: persistent workgroups, software-pipelined over parts.
// A workgroup walks parts blockIdx.x, + gridDim.x, ...  While part g steps, the metadata of part g' = g + gridDim.x
// is fetched (one 12-B load per lane) and its input is copied into LDS by DMA; the moment g's last step is done --
// its operator registers are dead -- the operator loads of g' are issued, and g's epilogue (slab -> HBM) and g''s
// prologue (input -> first vector) run underneath them.  What stays exposed per part is the operator stream itself
// (a CU pulls ~12-24 GB/s from HBM); the metadata hops, the input latency and the output stores are hidden.
//   LDS: xa | xb (two copies each) | slab | inb   -- inb = staged x (fwd) or the second slab (bwd, `early`);
//   bwd without room for two slabs (`early` = 0) copies the next input in after the last step instead.
template <class Op, int NT, bool BWD, bool ORD>     // ORD: forward with the orders epilogue (a separate instantiation:
__global__ void __launch_bounds__(NT, Op::MINW)      // compiled into the plain kernel, its code costs the step loop registers)
k_chain_sell(const int *__restrict__ col, const float *__restrict__ val, SellDev sd, const float *__restrict__ IN,
             float *__restrict__ RES, int T, EvTable tab, int F, int n_parts, int max_rows, int early)
{
    constexpr int C = Op::C, CP = Op::CP;
    extern __shared__ __align__(16) float smem[];
    const int nv = 2 * sell_padR(max_rows) * CP;                 // two copies per vector, a multiple of 32 floats
    const int so = ORD ? 1 : 0, Fs = F + so;                    // fused orders: slab rows are [x | bank]
    const int slab_sz = (max_rows * C * Fs + 3) & ~3;
    float *xa = smem, *xb = smem + nv, *slab = smem + 2 * nv;
    float *inb = (BWD && !early) ? slab : slab + slab_sz;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int in_w = BWD ? C * F : C;                            // input floats per row

    int li = blockIdx.x;                          // index into the launch's list of parts
    if (li >= n_parts) return;
    int part = sd.part_list ? sd.part_list[li] : li;            // part of the packed operator
    int row0 = sd.part_row0 ? sd.part_row0[li] : -1;            // first row in IN / RES (-1: the operator's own)
    Op op;
    // first part: nothing to hide behind
    int rt, te0, sw;
    {
        const int *lm = sd.lane_meta + ((int64_t)(part * sd.nsl + wave) * 64 + lane) * 3;
        rt = lm[0]; te0 = lm[1]; sw = lm[2];
        const int gs0 = __builtin_amdgcn_readlane(sw, 3), rows0 = __builtin_amdgcn_readlane(sw, 4);
        if (row0 < 0) row0 = gs0;
        dma_in<NT>(BWD ? slab : inb, IN + (int64_t)row0 * in_w, rows0 * in_w);
    }
    int prev_row0 = 0, prev_rows = -1, prev_gs = 0;   // part whose result is still in LDS (epilogue pending)
    for (;;) {
        // which part comes next (read now, needed after the prologue)
        const int liN = li + gridDim.x;
        const bool has_next = liN < n_parts;      // uniform
        int partN = liN, row0N = -1;
        if (has_next && sd.part_list) partN = sd.part_list[liN];
        if (has_next && sd.part_row0) row0N = sd.part_row0[liN];

        const int doff = __builtin_amdgcn_readlane(sw, 0), poff = __builtin_amdgcn_readlane(sw, 1);
        const int w = __builtin_amdgcn_readlane(sw, 2);
        const int gs = __builtin_amdgcn_readlane(sw, 3), rows = __builtin_amdgcn_readlane(sw, 4);
        // the operator registers are dead here: refill them ...
        STAMP(li, 0);
        op.load(sd, doff, poff, w, lane);
        STAMP(li, 1);
        // ... and store the previous part's result underneath the stream
        if (prev_rows >= 0) {
            if constexpr (Op::FRAMES) if (sd.frames) {     // back to global coordinates (uniform branch)
                if (!BWD) rotate_rows<false, NT>(slab, sd.frames + (int64_t)prev_gs * 9, prev_rows, Fs, 3 * Fs, Fs);   // ORD: [x | bank]
                else rotate_rows<false, NT>(xa, sd.frames + (int64_t)prev_gs * 9, prev_rows, 1, CP, 1);
                __syncthreads();
            }
            if (ORD) chain_epilogue_orders<NT>(slab, smem, 2 * nv, RES, prev_row0, prev_rows, C, F, sd.orders == 2);
            else chain_epilogue<BWD, NT>(xa, slab, RES, prev_row0, prev_rows, C, CP, F);
        }
        STAMP(li, 2);

        const int R = sell_padR(rows);
        const int r = (rt & 0x3ff) == 0x3ff ? -1 : (rt & 0x3ff);
        const int tn = (int)((unsigned)rt >> 10);
        const bool own = r >= 0;
        const int rB = own ? R + sell_copyB(r, R) : 0;
        // input staged?  (first part, or the late copy of the single-slab backward: wait for it -- and, loads being
        // returned in order, for the operator issued after it; otherwise it landed many steps ago)
        if (prev_rows < 0 || (BWD && !early)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                          // every wave's copy has landed; the epilogue's LDS reads are done
        STAMP(li, 3);
        if constexpr (Op::FRAMES) if (sd.frames) {         // the staged input, into the local frames
            if (!BWD) rotate_rows<true, NT>(inb, sd.frames + (int64_t)gs * 9, rows, 1, 3, 1);
            else rotate_rows<true, NT>(slab, sd.frames + (int64_t)gs * 9, rows, F, 3 * F, F);
            __syncthreads();
        }
        int p = chain_prologue<BWD, NT>(xa, xb, slab, BWD ? slab : inb, rows, R, C, CP, F, T, tab, Fs, so);
        STAMP(li, 4);

        int rtN = 0, te0N = 0, swN = 0;
        if (has_next) {                           // metadata of the next part: lands during the first step
            const int *lm = sd.lane_meta + ((int64_t)(partN * sd.nsl + wave) * 64 + lane) * 3;
            rtN = lm[0]; te0N = lm[1]; swN = lm[2];
        }
        bool staged = !(has_next && (!BWD || early));   // "nothing to stage early"
        for (int t = 1; t <= T; ++t) {
            int q0, q1;
            step_range<BWD>(tab, BWD ? T - t : t, p, q0, q1);
            float acc[C];
#pragma unroll
            for (int c = 0; c < C; ++c) acc[c] = 0.f;
#ifdef ESCAT_STAMPS
            if (t == 5) { STAMPW(li, 8, 0); STAMPW(li, 12, NT - 64); }
#endif
            op.gather(xa, w, acc);
#ifdef ESCAT_STAMPS
            if (t == 5) { STAMPW(li, 9, 0); STAMPW(li, 13, NT - 64); }
#endif
            if (own) {
                if (tn) Op::tail(val, col, te0, tn, gs, xa, acc);
                XVec<CP> y;
#pragma unroll
                for (int c = 0; c < CP; ++c) y.v[c] = 0.f;
#pragma unroll
                for (int c = 0; c < C; ++c) y.v[c] = apply_events<BWD>(tab, q0, q1, c, acc[c], slab + (r * C + c) * Fs + so);
                y.store(xb + r * CP); y.store(xb + rB * CP);       // whole padded rows: one wide store per copy
            }
#ifdef ESCAT_STAMPS
            if (t == 5) { STAMPW(li, 10, 0); STAMPW(li, 14, NT - 64); }
#endif
            __syncthreads();
            float *tmp = xa; xa = xb; xb = tmp;
#ifdef ESCAT_STAMPS
            if (t == 1) STAMP(li, 5);
            if (t == 5) { STAMPW(li, 11, 0); STAMPW(li, 15, NT - 64); }
#endif
            if (!staged) {                        // after the first step: start the DMA of the next part's input
                const int gsN = __builtin_amdgcn_readlane(swN, 3), rowsN = __builtin_amdgcn_readlane(swN, 4);
                dma_in<NT>(inb, IN + (int64_t)(row0N < 0 ? gsN : row0N) * in_w, rowsN * in_w);
                staged = true;
            }
        }
        prev_row0 = row0; prev_rows = rows; prev_gs = gs;
        STAMP(li, 6);
        if (!has_next) break;
        {
            const int gsN = __builtin_amdgcn_readlane(swN, 3), rowsN = __builtin_amdgcn_readlane(swN, 4);
            if (row0N < 0) row0N = gsN;
            // fwd / early bwd with T == 0 never reached the in-loop DMA; the single-slab bwd copies now (its slab is dead)
            if (!staged || (BWD && !early)) dma_in<NT>(inb, IN + (int64_t)row0N * in_w, rowsN * in_w);
            if (!BWD || early) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // staged long ago: costs nothing
        }
        STAMP(li, 7);
        li = liN; part = partN; row0 = row0N; rt = rtN; te0 = te0N; sw = swN;
        if (BWD && early) { float *tmp = slab; slab = inb; inb = tmp; }
    }
    if constexpr (Op::FRAMES) if (sd.frames) {
        if (!BWD) rotate_rows<false, NT>(slab, sd.frames + (int64_t)prev_gs * 9, prev_rows, Fs, 3 * Fs, Fs);   // ORD: [x | bank]
        else rotate_rows<false, NT>(xa, sd.frames + (int64_t)prev_gs * 9, prev_rows, 1, CP, 1);
        __syncthreads();
    }
    if (ORD) chain_epilogue_orders<NT>(slab, smem, 2 * nv, RES, prev_row0, prev_rows, C, F, sd.orders == 2);
    else chain_epilogue<BWD, NT>(xa, slab, RES, prev_row0, prev_rows, C, CP, F);
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
k_scatter_orders_bwd(const float *__restrict__ W1, int64_t ldw1, const float *__restrict__ dW, int64_t rows, int F, int Wtot,
                     int second, float *__restrict__ dx0, float *__restrict__ dW1)
{
    const int64_t total = rows * F;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int64_t r = idx / F; const int f = (int)(idx % F);
        const float *d = dW + r * Wtot; const float *w1 = W1 + r * ldw1;
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

// Tiled forms for the bank sizes that occur (F <= 9): one thread per row builds / consumes its Wtot values in registers,
// a wave-private LDS tile turns the (rows, Wtot) side into whole-line 16-byte accesses.  The element-wise kernels above
// re-derive the pair index per element and move 4 bytes per lane: 1.8 TB/s; these run at the copy rate.
template <int F, bool SECOND>
__global__ void __launch_bounds__(256)
k_orders_fwd_t(const float *__restrict__ x0, const float *__restrict__ W1, int64_t rows, float *__restrict__ W)
{
    constexpr int WT = 1 + F + (SECOND ? F * (F - 1) / 2 : 0);
    __shared__ __align__(16) float tile[4][64 * WT];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float *t = tile[wave];
    for (int64_t base = (int64_t)blockIdx.x * 256; base < rows; base += (int64_t)gridDim.x * 256) {
        const int64_t wb = base + wave * 64, r = wb + lane;
        if (r < rows) {
            float w[F];
#pragma unroll
            for (int f = 0; f < F; ++f) w[f] = W1[r * F + f];
            float *o = t + lane * WT;
            o[0] = x0[r];
#pragma unroll
            for (int f = 0; f < F; ++f) o[1 + f] = w[f];
            if (SECOND) {
                int q = 1 + F;
#pragma unroll
                for (int i = 0; i < F; ++i)
#pragma unroll
                    for (int j = i + 1; j < F; ++j) o[q++] = w[i] * w[j];
            }
        }
        __syncthreads();
        const int64_t nv = rows - wb < 64 ? (rows - wb > 0 ? rows - wb : 0) : 64;      // valid rows of this wave's tile
        const int n4 = (int)(nv * WT) / 4;
        float4 *dst4 = reinterpret_cast<float4 *>(W + wb * WT);                        // wb*WT*4 B: a multiple of 16
        const float4 *src4 = reinterpret_cast<const float4 *>(t);
        for (int i = lane; i < n4; i += 64) dst4[i] = src4[i];
        for (int i = n4 * 4 + lane; i < (int)(nv * WT); i += 64) W[wb * WT + i] = t[i];
        __syncthreads();
    }
}

template <int F, bool SECOND>
__global__ void __launch_bounds__(256)
k_orders_bwd_t(const float *__restrict__ W1, int64_t ldw1, const float *__restrict__ dW, int64_t rows,
               float *__restrict__ dx0, float *__restrict__ dW1)
{
    constexpr int WT = 1 + F + (SECOND ? F * (F - 1) / 2 : 0);
    __shared__ __align__(16) float tile[4][64 * WT];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float *t = tile[wave];
    for (int64_t base = (int64_t)blockIdx.x * 256; base < rows; base += (int64_t)gridDim.x * 256) {
        const int64_t wb = base + wave * 64, r = wb + lane;
        const int64_t nv = rows - wb < 64 ? (rows - wb > 0 ? rows - wb : 0) : 64;
        const int n4 = (int)(nv * WT) / 4;
        const float4 *src4 = reinterpret_cast<const float4 *>(dW + wb * WT);
        float4 *dst4 = reinterpret_cast<float4 *>(t);
        for (int i = lane; i < n4; i += 64) dst4[i] = src4[i];
        for (int i = n4 * 4 + lane; i < (int)(nv * WT); i += 64) t[i] = dW[wb * WT + i];
        __syncthreads();
        if (r < rows) {
            const float *d = t + lane * WT;
            float g[F], w[F];
#pragma unroll
            for (int f = 0; f < F; ++f) { g[f] = d[1 + f]; w[f] = SECOND ? W1[r * ldw1 + f] : 0.f; }
            if (SECOND) {
                int q = 1 + F;
#pragma unroll
                for (int i = 0; i < F; ++i)
#pragma unroll
                    for (int j = i + 1; j < F; ++j) { const float dq = d[q++]; g[i] = fmaf(dq, w[j], g[i]); g[j] = fmaf(dq, w[i], g[j]); }
            }
            dx0[r] = d[0];
#pragma unroll
            for (int f = 0; f < F; ++f) dW1[r * F + f] = g[f];
        }
        __syncthreads();
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
    if (const char *e = getenv("ESCAT_CSR_G")) {        // tuning aid: lanes per row of the lane-group CSR kernels
        const int g = atoi(e);
        if (g == 4 || g == 8 || g == 16) return g;
    }
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
                           float *y, int ldy, int64_t n, const StepEv &ev, float *OUT, int64_t ld_row, int F, hipStream_t st,
                           int planes = 1)
{
    if (planes == 3 || planes == 4) {
        if (C != 3) return fail(ESCAT_EINVAL, "one value per channel (val_planes = 3, 4) needs C = 3, got %d", C);
#define GO(GG, VV, RPB) hipLaunchKernelGGL((k_csr_step<3, GG, MODE, VV>), dim3(grid_for(n, RPB)), dim3(256), 0, st, rp, ci, v, xin, ldx, y, ldy, n, ev, OUT, ld_row, F)
        if (planes == 3) { if (G == 4) GO(4, 3, 64); else if (G == 8) GO(8, 3, 32); else GO(16, 3, 16); }
        else             { if (G == 4) GO(4, 4, 64); else if (G == 8) GO(8, 4, 32); else GO(16, 4, 16); }
#undef GO
        return ESCAT_OK;
    }
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

#define LIST_NEEDS_REG "escat_sell.part_list / .orders / .val_planes = 3 need the register-resident path (a packed operator whose parts have <= 512 rows and fit LDS)"

static int sell_to_dev(const escat_sell *sell, SellDev *sd)
{
    if (!sell->val || !sell->lc || !sell->lane_meta) return fail(ESCAT_EINVAL, "escat_sell has a NULL member");
    if (((uintptr_t)sell->val & 15) != 0) return fail(ESCAT_EINVAL, "escat_sell.val must be 16-byte aligned");
    if (sell->slices_per_part != 4 && sell->slices_per_part != 8)
        return fail(ESCAT_EINVAL, "escat_sell.slices_per_part=%d: must be 4 (parts <= 256 rows) or 8", sell->slices_per_part);
    if ((sell->part_list != nullptr) != (sell->part_row0 != nullptr) || (sell->part_list && sell->n_list < 0))
        return fail(ESCAT_EINVAL, "escat_sell.part_list and part_row0 go together (n_list >= 0)");
    sd->val = sell->val; sd->lc = sell->lc; sd->lane_meta = sell->lane_meta; sd->dmax = sell->dmax; sd->nsl = sell->slices_per_part;
    sd->part_list = sell->part_list; sd->part_row0 = sell->part_row0;
    if (sell->orders < 0 || sell->orders > 2) return fail(ESCAT_EINVAL, "escat_sell.orders=%d: 0, 1 or 2", sell->orders);
    sd->orders = sell->orders;
    if (sell->val_planes != 0 && sell->val_planes != 1 && sell->val_planes != 3 && sell->val_planes != 4)
        return fail(ESCAT_EINVAL, "escat_sell.val_planes=%d: 0/1 (one value per entry), 3 (one per channel) or 4 (|value| + sign bits)", sell->val_planes);
    sd->planes = sell->val_planes >= 3 ? sell->val_planes : 1;
    sd->frames = sell->frames;
    if (sd->frames && sd->planes < 3) return fail(ESCAT_EINVAL, "escat_sell.frames goes with val_planes = 3");
    return ESCAT_OK;
}

static int cu_count()
{
    static int cus[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (!cus[dev]) {
        hipDeviceProp_t prop;
        cus[dev] = hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    return cus[dev];
}

// LDS of a k_chain_sell workgroup: two vectors of two copies, the slab, and the staged input (fwd: x; bwd: a second
// slab if `want` workgroups per CU still fit -- *early = 1 -- else none)
static size_t sell_lds_bytes(int C, int CP, int max_rows, int F, bool bwd, int want, int *early, int orders = 0)
{
    const size_t nv = (size_t)2 * sell_padR(max_rows) * CP;
    const size_t slab = ((size_t)max_rows * C * (F + (!bwd && orders ? 1 : 0)) + 3) & ~(size_t)3;
    const size_t base = (2 * nv + slab) * sizeof(float);
    if (!bwd) { *early = 1; return base + ((((size_t)max_rows * C + 3) & ~(size_t)3)) * sizeof(float); }
    const size_t two = base + slab * sizeof(float);
    if (two * want <= LDS_BUDGET) { *early = 1; return two; }
    *early = 0;
    return base;
}

template <class Op, int NT, bool BWD, bool ORD>
static int launch_sell_k2(const int *ci, const float *v, int n_parts, int max_rows, const SellDev &sd, const float *IN,
                          float *RES, int T, const EvTable &tab, int F, hipStream_t st)
{
    int early = 0;
    const int want = Op::MINW == 1 ? 1 : (NT == 512 ? 2 : 4);
    const size_t lds = sell_lds_bytes(Op::C, Op::CP, max_rows, F, BWD, want, &early, sd.orders);
    if (lds > LDS_BUDGET) return fail(ESCAT_ERANGE, "register-resident chain needs %zu B of LDS", lds);
    auto kern = k_chain_sell<Op, NT, BWD, ORD>;
    int rc = set_lds(kern, lds); if (rc) return rc;
    // resident workgroups per CU: cached per instantiation, LDS size and device (thread_local: no locking needed)
    static thread_local size_t cached_lds = (size_t)-1; static thread_local int cached_per_cu = 0, cached_dev = -1;
    int dev_now = 0;
    (void)hipGetDevice(&dev_now);
    if (cached_lds != lds || cached_dev != dev_now) {
        int per = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per, kern, NT, lds) != hipSuccess || per < 1) per = 1;
        cached_per_cu = per; cached_lds = lds; cached_dev = dev_now;
    }
    const int64_t resident = (int64_t)cached_per_cu * cu_count();
    const int grid = (int)(n_parts < resident ? n_parts : resident);           // persistent: one wave of workgroups
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NT), lds, st, ci, v, sd, IN, RES, T, tab, F, n_parts, max_rows, early);
    return ESCAT_OK;
}

template <class Op, int NT, bool BWD>
static int launch_sell_k(const int *ci, const float *v, int n_parts, int max_rows, const SellDev &sd, const float *IN,
                         float *RES, int T, const EvTable &tab, int F, hipStream_t st)
{
    if (sd.nsl * 64 != NT) return fail(ESCAT_EINVAL, "escat_sell.slices_per_part=%d does not match parts of <= %d rows", sd.nsl, max_rows);
    if (!BWD && sd.orders) return launch_sell_k2<Op, NT, false, true>(ci, v, n_parts, max_rows, sd, IN, RES, T, tab, F, st);
    return launch_sell_k2<Op, NT, BWD, false>(ci, v, n_parts, max_rows, sd, IN, RES, T, tab, F, st);
}

template <int DMAX, bool BWD>
static int launch_bsr3_sell_d(int max_rows, const int *ci, const float *v, int n_parts, const SellDev &sd, const float *IN,
                              float *RES, int T, const EvTable &tab, int F, hipStream_t st)
{
    if (max_rows <= 256) return launch_sell_k<OpBsr3<DMAX>, 256, BWD>(ci, v, n_parts, max_rows, sd, IN, RES, T, tab, F, st);
    return launch_sell_k<OpBsr3<DMAX>, 512, BWD>(ci, v, n_parts, max_rows, sd, IN, RES, T, tab, F, st);
}
template <bool BWD>
static int launch_bsr3_sell(int max_rows, const int *ci, const float *v, int n_parts, const SellDev &sd, const float *IN,
                            float *RES, int T, const EvTable &tab, int F, hipStream_t st)
{
    switch (sd.dmax) {
        case 8:  return launch_bsr3_sell_d<8, BWD>(max_rows, ci, v, n_parts, sd, IN, RES, T, tab, F, st);
        case 12: return launch_bsr3_sell_d<12, BWD>(max_rows, ci, v, n_parts, sd, IN, RES, T, tab, F, st);
        case 16: return launch_bsr3_sell_d<16, BWD>(max_rows, ci, v, n_parts, sd, IN, RES, T, tab, F, st);
        case 20: return launch_bsr3_sell_d<20, BWD>(max_rows, ci, v, n_parts, sd, IN, RES, T, tab, F, st);
        case 22: return launch_bsr3_sell_d<22, BWD>(max_rows, ci, v, n_parts, sd, IN, RES, T, tab, F, st);
    }
    return fail(ESCAT_EINVAL, "escat_sell.dmax=%d: BSR-3 packs must use 8, 12, 16, 20 or 22", sd.dmax);
}

template <int C, int DMAX, bool BWD>
static int launch_csr_sell_d(int max_rows, const int *ci, const float *v, int n_parts, const SellDev &sd, const float *IN,
                             float *RES, int T, const EvTable &tab, int F, hipStream_t st)
{
    if (max_rows <= 256) return launch_sell_k<OpCsr<C, DMAX>, 256, BWD>(ci, v, n_parts, max_rows, sd, IN, RES, T, tab, F, st);
    return launch_sell_k<OpCsr<C, DMAX>, 512, BWD>(ci, v, n_parts, max_rows, sd, IN, RES, T, tab, F, st);
}
template <int C, bool BWD>
static int launch_csr_sell_c(int max_rows, const int *ci, const float *v, int n_parts, const SellDev &sd, const float *IN,
                             float *RES, int T, const EvTable &tab, int F, hipStream_t st)
{
    switch (sd.dmax) {
        case 8:  return launch_csr_sell_d<C, 8, BWD>(max_rows, ci, v, n_parts, sd, IN, RES, T, tab, F, st);
        case 16: return launch_csr_sell_d<C, 16, BWD>(max_rows, ci, v, n_parts, sd, IN, RES, T, tab, F, st);
        case 20: return launch_csr_sell_d<C, 20, BWD>(max_rows, ci, v, n_parts, sd, IN, RES, T, tab, F, st);
        case 24: return launch_csr_sell_d<C, 24, BWD>(max_rows, ci, v, n_parts, sd, IN, RES, T, tab, F, st);
        case 32: return launch_csr_sell_d<C, 32, BWD>(max_rows, ci, v, n_parts, sd, IN, RES, T, tab, F, st);
    }
    return fail(ESCAT_EINVAL, "escat_sell.dmax=%d: CSR packs must use 8, 16, 20, 24 or 32", sd.dmax);
}
template <int DMAX, bool BWD>
static int launch_csrv3_sell_d(int max_rows, const int *ci, const float *v, int n_parts, const SellDev &sd, const float *IN,
                               float *RES, int T, const EvTable &tab, int F, hipStream_t st)
{
    if (sd.planes == 4) {
        if (max_rows <= 256) return launch_sell_k<OpCsrV3<DMAX, true>, 256, BWD>(ci, v, n_parts, max_rows, sd, IN, RES, T, tab, F, st);
        return launch_sell_k<OpCsrV3<DMAX, true>, 512, BWD>(ci, v, n_parts, max_rows, sd, IN, RES, T, tab, F, st);
    }
    if (max_rows <= 256) return launch_sell_k<OpCsrV3<DMAX>, 256, BWD>(ci, v, n_parts, max_rows, sd, IN, RES, T, tab, F, st);
    return launch_sell_k<OpCsrV3<DMAX>, 512, BWD>(ci, v, n_parts, max_rows, sd, IN, RES, T, tab, F, st);
}
template <bool BWD>
static int launch_csr_sell(int C, int max_rows, const int *ci, const float *v, int n_parts, const SellDev &sd,
                           const float *IN, float *RES, int T, const EvTable &tab, int F, hipStream_t st)
{
    if (sd.planes >= 3) {
        if (C != 3) return fail(ESCAT_EINVAL, "a pack with one value per channel (val_planes = 3, 4) needs C = 3, got %d", C);
        switch (sd.dmax) {
            case 8:  return launch_csrv3_sell_d<8, BWD>(max_rows, ci, v, n_parts, sd, IN, RES, T, tab, F, st);
            case 16: return launch_csrv3_sell_d<16, BWD>(max_rows, ci, v, n_parts, sd, IN, RES, T, tab, F, st);
            case 20: return launch_csrv3_sell_d<20, BWD>(max_rows, ci, v, n_parts, sd, IN, RES, T, tab, F, st);
            case 24: return launch_csrv3_sell_d<24, BWD>(max_rows, ci, v, n_parts, sd, IN, RES, T, tab, F, st);
        }
        return fail(ESCAT_EINVAL, "escat_sell.dmax=%d: per-channel CSR packs must use 8, 16, 20 or 24", sd.dmax);
    }
    switch (C) {
#define CASE(c) case c: return launch_csr_sell_c<c, BWD>(max_rows, ci, v, n_parts, sd, IN, RES, T, tab, F, st);
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
    // one value per channel: val is (nnz, 3).  Runs on the register-resident path (packed) or the multi-launch path (a
    // descriptor-only escat_sell: val = NULL, val_planes = 3); the LDS-streaming lane-group kernel has no such form.
    const int planes = sell && sell->val_planes >= 3 ? sell->val_planes : 1;
    const bool packed = sell && sell->val != nullptr;
    if (planes >= 3 && C != 3) return fail(ESCAT_EINVAL, "one value per channel (val_planes = 3, 4) needs C = 3, got %d", C);
    if (sell && !packed && (sell->part_list || sell->orders || sell->frames)) return fail(ESCAT_EINVAL, LIST_NEEDS_REG);
    if (part_ptr && n_parts > 0 && max_part_rows > 0 && force < 3 && (planes == 1 || packed)) {
        const size_t lds = (size_t)max_part_rows * C * (2 + F) * sizeof(float);
        if (lds <= LDS_BUDGET) {
            const int CP = C == 1 ? 1 : (C == 2 ? 2 : (C <= 4 ? 4 : 8));
            int early_ = 0;
            const size_t lds_s = sell_lds_bytes(C, CP, max_part_rows, F, BWD, max_part_rows <= 256 ? 4 : 2, &early_, sell ? sell->orders : 0);
            if (max_part_rows <= 512 && packed && force != 2 && lds_s <= LDS_BUDGET) {
                SellDev sd;
                rc = sell_to_dev(sell, &sd); if (rc) return rc;
                if (BWD && sd.orders) return fail(ESCAT_EINVAL, "escat_sell.orders must be 0 for a backward entry point");
                rc = launch_csr_sell<BWD>(C, max_part_rows, col, val, sd.part_list ? sell->n_list : n_parts, sd, IN, RES, T, tab, F, st);
            } else if (sell && (sell->part_list || sell->orders || sell->val_planes >= 3))
                return fail(ESCAT_EINVAL, LIST_NEEDS_REG);
            else
                rc = launch_chain_csr<BWD>(C, G, rowptr, col, val, part_ptr, n_parts, lds, IN, RES, T, tab, F, st);
            if (rc) return rc;
            CHECK_LAUNCH("k_chain_csr");
            return ESCAT_OK;
        }
    }
    if (sell && (sell->part_list || sell->orders || sell->frames)) return fail(ESCAT_EINVAL, LIST_NEEDS_REG);
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
            rc = launch_csr_step<1>(C, G, rowptr, col, val, xin, C, y, C, n, se, RES, ld_row, F, st, planes); if (rc) return rc;
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
            rc = launch_csr_step<2>(C, G, rowptr, col, val, xin, C, y, C, n, se, Gm, ld_row, F, st, planes); if (rc) return rc;
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
        int early_ = 0;
        const size_t lds_s = sell_lds_bytes(3, 4, max_part_rows, F, BWD, 1, &early_, sell ? sell->orders : 0);
        if (lds <= LDS_BUDGET && lds_s <= LDS_BUDGET && max_part_rows <= 512 && sell && force != 2) {
            SellDev sd;
            rc = sell_to_dev(sell, &sd); if (rc) return rc;
            if (BWD && sd.orders) return fail(ESCAT_EINVAL, "escat_sell.orders must be 0 for a backward entry point");
            rc = launch_bsr3_sell<BWD>(max_part_rows, col, blk, sd.part_list ? sell->n_list : n_parts, sd, IN, RES, T, tab, F, st);
            if (rc) return rc;
            CHECK_LAUNCH("k_chain_bsr3_sell");
            return ESCAT_OK;
        }
        if (sell && (sell->part_list || sell->orders)) return fail(ESCAT_EINVAL, LIST_NEEDS_REG);
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
    if (sell && (sell->part_list || sell->orders)) return fail(ESCAT_EINVAL, LIST_NEEDS_REG);
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

template <bool SECOND>
static bool launch_orders_fwd_t(int F, int grid, hipStream_t st, const float *x0, const float *W1, int64_t rows, float *W)
{
    switch (F) {
#define CASE(f) case f: hipLaunchKernelGGL((k_orders_fwd_t<f, SECOND>), dim3(grid), dim3(256), 0, st, x0, W1, rows, W); return true;
        CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(7) CASE(8) CASE(9)
#undef CASE
    }
    return false;
}
template <bool SECOND>
static bool launch_orders_bwd_t(int F, int grid, hipStream_t st, const float *W1, int64_t ldw1, const float *dW, int64_t rows,
                                float *dx0, float *dW1)
{
    switch (F) {
#define CASE(f) case f: hipLaunchKernelGGL((k_orders_bwd_t<f, SECOND>), dim3(grid), dim3(256), 0, st, W1, ldw1, dW, rows, dx0, dW1); return true;
        CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(7) CASE(8) CASE(9)
#undef CASE
    }
    return false;
}

extern "C" int escat_scatter_orders_f32(const float *x0, const float *W1, int64_t rows, int F, int second_order, float *W,
                                        void *stream)
{
    if (!x0 || !W1 || !W) return fail(ESCAT_EINVAL, "NULL pointer");
    if (F < 1 || F > 64) return fail(ESCAT_ERANGE, "F=%d out of range [1,64]", F);
    if (rows <= 0) return rows == 0 ? ESCAT_OK : fail(ESCAT_EINVAL, "negative rows");
    const int Wtot = 1 + F + (second_order ? F * (F - 1) / 2 : 0);
    const bool aligned = (((uintptr_t)W) & 15) == 0;
    const int grid = grid_for(rows, 256);
    if (!(aligned && (second_order ? launch_orders_fwd_t<true>(F, grid, S(stream), x0, W1, rows, W)
                                   : launch_orders_fwd_t<false>(F, grid, S(stream), x0, W1, rows, W))))
        hipLaunchKernelGGL(k_scatter_orders, dim3(grid_for(rows * Wtot, 256)), dim3(256), 0, S(stream), x0, W1, rows, F, Wtot, W);
    CHECK_LAUNCH("k_scatter_orders");
    return ESCAT_OK;
}

extern "C" int escat_scatter_orders_bwd_f32(const float *W1, int64_t ldw1, const float *dW, int64_t rows, int F, int second_order,
                                            float *dx0, float *dW1, void *stream)
{
    if (!W1 || !dW || !dx0 || !dW1) return fail(ESCAT_EINVAL, "NULL pointer");
    if (F < 1 || F > 64) return fail(ESCAT_ERANGE, "F=%d out of range [1,64]", F);
    if (ldw1 < F) return fail(ESCAT_EINVAL, "ldw1=%lld < F=%d", (long long)ldw1, F);
    if (rows <= 0) return rows == 0 ? ESCAT_OK : fail(ESCAT_EINVAL, "negative rows");
    const int Wtot = 1 + F + (second_order ? F * (F - 1) / 2 : 0);
    const bool aligned = (((uintptr_t)dW) & 15) == 0;
    const int grid = grid_for(rows, 256);
    if (!(aligned && (second_order ? launch_orders_bwd_t<true>(F, grid, S(stream), W1, ldw1, dW, rows, dx0, dW1)
                                   : launch_orders_bwd_t<false>(F, grid, S(stream), W1, ldw1, dW, rows, dx0, dW1))))
        hipLaunchKernelGGL(k_scatter_orders_bwd, dim3(grid_for(rows * F, 256)), dim3(256), 0, S(stream), W1, ldw1, dW, rows, F, Wtot,
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

