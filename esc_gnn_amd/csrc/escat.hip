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
#include "escat_dev.h"

// ------------------------------------------------------------------ errors
static thread_local char g_err[512] = "";
int escat_fail(int code, const char *fmt, ...)      // shared with escat_build.hip (escat_internal.h)
{
    va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof(g_err), fmt, ap); va_end(ap);
    return code;
}
extern "C" const char *escat_last_error(void) { return g_err; }
extern "C" int escat_abi_version(void) { return 5; }

#ifdef ESCAT_STAMPS
extern "C" int escat_debug_set_stamps_bsr3(void *), escat_debug_set_stamps_csr12(void *), escat_debug_set_stamps_csr34(void *),
               escat_debug_set_stamps_csr56(void *), escat_debug_set_stamps_csr78(void *), escat_debug_set_stamps_csrv3(void *),
               escat_debug_set_stamps_bsr3h(void *);
extern "C" int escat_debug_set_stamps(void *p)
{
    return escat_debug_set_stamps_bsr3(p) | escat_debug_set_stamps_csr12(p) | escat_debug_set_stamps_csr34(p) |
           escat_debug_set_stamps_csr56(p) | escat_debug_set_stamps_csr78(p) | escat_debug_set_stamps_csrv3(p) |
           escat_debug_set_stamps_bsr3h(p);
}
#endif

static inline int enc_sign(int sign, bool store, bool ab) { return 1 | (sign < 0 ? 2 : 0) | (store ? EV_STORE : 0) | (ab ? EV_ABS : 0); }

// validates + encodes the host event list; marks the first event of every (slot, chan)
static int build_table(const escat_event *ev, int n_ev, int T, int F, int C, EvTable *tab, bool bwd = false)
{
    if (n_ev < 0 || n_ev > ESCAT_MAX_EVENTS) return fail(ESCAT_ERANGE, "n_events=%d exceeds ESCAT_MAX_EVENTS", n_ev);
    if (n_ev > 0 && !ev) return fail(ESCAT_EINVAL, "events is NULL");
    if (F < 1 || F > ESCAT_MAX_SLOTS) return fail(ESCAT_ERANGE, "F=%d out of range [1,%d]", F, ESCAT_MAX_SLOTS);
    static thread_local unsigned char seen[ESCAT_MAX_SLOTS * ESCAT_MAX_CHANNELS];   // 1 = touched, 2 = closed by an |.| event
    memset(seen, 0, sizeof(seen));
    tab->n = n_ev;
    for (int k = 0; k < n_ev; ++k) {
        int sg = ev[k].sign;
        const bool ab = sg == 1 + ESCAT_EVENT_ABS || sg == -1 + ESCAT_EVENT_ABS;
        if (ab) sg -= ESCAT_EVENT_ABS;
        if (ab && bwd) return fail(ESCAT_EINVAL, "event %d: ESCAT_EVENT_ABS is a forward-only flag", k);
        if (ev[k].step < 0 || ev[k].step > T) return fail(ESCAT_EINVAL, "event %d: step %d outside [0,%d]", k, ev[k].step, T);
        if (k && ev[k].step < ev[k - 1].step) return fail(ESCAT_EINVAL, "events not sorted by step");
        if (ev[k].slot < 0 || ev[k].slot >= F) return fail(ESCAT_EINVAL, "event %d: slot %d outside [0,%d)", k, ev[k].slot, F);
        if (ev[k].chan >= C) return fail(ESCAT_EINVAL, "event %d: chan %d >= C=%d", k, ev[k].chan, C);
        if (sg != 1 && sg != -1) return fail(ESCAT_EINVAL, "event %d: sign must be +1/-1 (+ ESCAT_EVENT_ABS)", k);
        const int c0 = ev[k].chan < 0 ? 0 : ev[k].chan, c1 = ev[k].chan < 0 ? C : ev[k].chan + 1;
        int touched = 0;
        for (int c = c0; c < c1; ++c) {
            if (seen[ev[k].slot * ESCAT_MAX_CHANNELS + c] == 2)
                return fail(ESCAT_EINVAL, "event %d follows the ESCAT_EVENT_ABS event of slot %d (it must be the slot's last)", k, ev[k].slot);
            touched += seen[ev[k].slot * ESCAT_MAX_CHANNELS + c];
        }
        if (touched != 0 && touched != c1 - c0)
            return fail(ESCAT_EINVAL, "event %d mixes touched and untouched channels of slot %d", k, ev[k].slot);
        const bool first = touched == 0;      // first touch of these OUT elements: store instead of accumulate
        for (int c = c0; c < c1; ++c) seen[ev[k].slot * ESCAT_MAX_CHANNELS + c] = ab ? 2 : 1;
        tab->e[k].step = ev[k].step; tab->e[k].slot = ev[k].slot; tab->e[k].chan = ev[k].chan;
        tab->e[k].sign = enc_sign(sg, first, ab);
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

// The same step, staged through LDS (round 2; the default of the multi-launch path).  The lane-group kernel above chains
// rowptr -> col -> x per lane and keeps 12 bytes per lane in flight: measured 3.6 TB/s at cfg3.  Here a workgroup takes
// STG_RPB consecutive block rows per trip: their blocks are ONE contiguous range of the value array, copied to LDS with
// 16-byte non-temporal loads that all go out together (the stream must not evict x from the L2s), the x rows of all
// those blocks are gathered into LDS at the same time, the rows' offsets ride along; then 8 lanes per row compute all
// three block rows out of LDS (odd strides: conflict-free).  tools/lds/step_bench.hip: 0.432 -> 0.316 ms per step on
// a cfg3-shaped operator (5.0 TB/s), the same kernel without any x gather 0.27 ms.
// A trip whose rows hold more than STG_CAP blocks is computed straight from global memory (same lanes, same order).
#ifndef STG_RPB
#define STG_RPB 32
#endif
#ifndef STG_CAP
#define STG_CAP (STG_RPB * 24)
#endif
#define STG_NQ ((STG_CAP * 9 + 3) / 4 + 1)
#define STG_LDS_BYTES ((STG_NQ * 4 + STG_CAP * 3 + STG_RPB + 4) * 4)
typedef float escat_vf4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ void __launch_bounds__(256)
k_bsr3_step_staged(const int *__restrict__ rowptr, const int *__restrict__ col, const float *__restrict__ blk,
                   const float *__restrict__ xin, float *__restrict__ yout, int nb, int64_t nnzb,
                   StepEv ev, float *OUT, int F)
{
    constexpr int LPR = 256 / STG_RPB, QPT = (STG_NQ + 255) / 256, EPT = (STG_CAP + 255) / 256;
    extern __shared__ float stg_lds[];
    float *vL = stg_lds, *xL = stg_lds + STG_NQ * 4;
    int *rpL = reinterpret_cast<int *>(xL + STG_CAP * 3);
    const int trips = (nb + STG_RPB - 1) / STG_RPB;
    const int64_t nfl = nnzb * 9;
    // The columns of a trip are loaded one trip AHEAD (three registers): the chain rowptr -> col -> x of a trip is then one
    // hop shorter on its critical path -- the next trip's row bounds and columns travel under this trip's value stream
    // and compute (round 3; the fully pipelined variant of round 2, values included, lost occupancy and was slower).
    int trip = blockIdx.x;
    int e0 = 0, e1 = 0;
    int c[EPT];
#pragma unroll
    for (int j = 0; j < EPT; ++j) c[j] = 0;
    if (trip < trips) {
        const int r0 = trip * STG_RPB, r1 = r0 + STG_RPB < nb ? r0 + STG_RPB : nb;
        e0 = rowptr[r0]; e1 = rowptr[r1];
        if (e1 - e0 <= STG_CAP) {
#pragma unroll
            for (int j = 0; j < EPT; ++j) { const int i = threadIdx.x + 256 * j; c[j] = i < e1 - e0 ? col[e0 + i] : 0; }
        }
    }
    for (; trip < trips; trip += gridDim.x) {
        const int r0 = trip * STG_RPB, r1 = r0 + STG_RPB < nb ? r0 + STG_RPB : nb;
        const int ne = e1 - e0;
        const bool staged = ne <= STG_CAP;
        const int64_t f0 = (int64_t)e0 * 9, a0 = f0 & ~(int64_t)3;
        // the next trip's bounds first: they are the oldest loads in flight when its columns are asked for below
        const int tripN = trip + gridDim.x;
        int e0N = 0, e1N = 0;
        if (tripN < trips) {
            const int r0N = tripN * STG_RPB, r1N = r0N + STG_RPB < nb ? r0N + STG_RPB : nb;
            e0N = rowptr[r0N]; e1N = rowptr[r1N];
        }
        int cN[EPT];
#pragma unroll
        for (int j = 0; j < EPT; ++j) cN[j] = 0;
        if (threadIdx.x <= r1 - r0) rpL[threadIdx.x] = rowptr[r0 + threadIdx.x] - e0;
        if (staged) {
            const int nq = (int)(((int64_t)e1 * 9 - a0 + 3) >> 2);
            float4 v[QPT];
#pragma unroll
            for (int j = 0; j < QPT; ++j) {
                const int q = threadIdx.x + 256 * j;
                const int64_t gq = a0 + 4 * (int64_t)q;
                if (q < nq) {
                    if (gq + 4 <= nfl) {
                        const escat_vf4 t = __builtin_nontemporal_load(reinterpret_cast<const escat_vf4 *>(blk + gq));
                        v[j] = make_float4(t.x, t.y, t.z, t.w);
                    } else {                                      // the array's last, partial 16 bytes
                        v[j].x = blk[gq]; v[j].y = gq + 1 < nfl ? blk[gq + 1] : 0.f; v[j].z = gq + 2 < nfl ? blk[gq + 2] : 0.f; v[j].w = 0.f;
                    }
                }
            }
            float xr[EPT][3];
#pragma unroll
            for (int j = 0; j < EPT; ++j) {
                const float *xp = xin + (int64_t)c[j] * 3;
                xr[j][0] = xp[0]; xr[j][1] = xp[1]; xr[j][2] = xp[2];
            }
            if (tripN < trips && e1N - e0N <= STG_CAP) {
#pragma unroll
                for (int j = 0; j < EPT; ++j) { const int i = threadIdx.x + 256 * j; cN[j] = i < e1N - e0N ? col[e0N + i] : 0; }
            }
#pragma unroll
            for (int j = 0; j < QPT; ++j) { const int q = threadIdx.x + 256 * j; if (q < nq) *reinterpret_cast<float4 *>(vL + 4 * q) = v[j]; }
#pragma unroll
            for (int j = 0; j < EPT; ++j) {
                const int i = threadIdx.x + 256 * j;
                if (i < ne) { xL[3 * i] = xr[j][0]; xL[3 * i + 1] = xr[j][1]; xL[3 * i + 2] = xr[j][2]; }
            }
        }
        __syncthreads();
        const int lr = threadIdx.x / LPR, row = r0 + lr, k = threadIdx.x % LPR;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f;
        if (row < r1) {
            const int i0 = rpL[lr], i1 = rpL[lr + 1];
            if (staged) {
                const float *vb = vL + (int)(f0 - a0);
                for (int i = i0 + k; i < i1; i += LPR) {
                    const float *b = vb + i * 9; const float *x = xL + 3 * i;
                    const float x0 = x[0], x1 = x[1], x2 = x[2];
                    s0 += b[0] * x0 + b[1] * x1 + b[2] * x2;
                    s1 += b[3] * x0 + b[4] * x1 + b[5] * x2;
                    s2 += b[6] * x0 + b[7] * x1 + b[8] * x2;
                }
            } else {
                for (int i = i0 + k; i < i1; i += LPR) {
                    const float *b = blk + ((int64_t)e0 + i) * 9; const float *x = xin + (int64_t)col[e0 + i] * 3;
                    const float x0 = x[0], x1 = x[1], x2 = x[2];
                    s0 += b[0] * x0 + b[1] * x1 + b[2] * x2;
                    s1 += b[3] * x0 + b[4] * x1 + b[5] * x2;
                    s2 += b[6] * x0 + b[7] * x1 + b[8] * x2;
                }
            }
        }
#pragma unroll
        for (int m = LPR / 2; m >= 1; m >>= 1) { s0 += __shfl_xor(s0, m, 64); s1 += __shfl_xor(s1, m, 64); s2 += __shfl_xor(s2, m, 64); }
        if (row < r1 && k < 3) {
            const int64_t r = 3 * (int64_t)row + k;
            float y = k == 0 ? s0 : k == 1 ? s1 : s2;
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
        if (!staged && tripN < trips && e1N - e0N <= STG_CAP) {       // (the unstaged trip did not reach the prefetch above)
#pragma unroll
            for (int j = 0; j < EPT; ++j) { const int i = threadIdx.x + 256 * j; cN[j] = i < e1N - e0N ? col[e0N + i] : 0; }
        }
        e0 = e0N; e1 = e1N;
#pragma unroll
        for (int j = 0; j < EPT; ++j) c[j] = cN[j];
        __syncthreads();
    }
}

// The scalar step staged the same way (north_star's "CSR SpMM ... with multi-column feature tiles staged in LDS"): a trip
// is RPB consecutive rows, their column / value ranges are contiguous and stream in with every load of the workgroup
// in flight at once, the x rows (C floats each) of all entries are gathered into LDS alongside, then 256 / RPB lanes
// per row reduce out of LDS.  VP as in k_csr_step.
template <int C> struct StgCsr { static constexpr int RPB = C <= 4 ? 64 : 32, CAP = RPB * 24; };
template <int C, int VP> constexpr int stg_csr_lds_bytes()
{
    return (StgCsr<C>::CAP * ((VP == 3 ? 3 : 1) + C + (VP == 4 ? 1 : 0)) + StgCsr<C>::RPB + 4) * 4;
}

template <int C, int MODE, int VP>
__global__ void __launch_bounds__(256)
k_csr_step_staged(const int *__restrict__ rowptr, const int *__restrict__ col, const float *__restrict__ val,
                  const float *__restrict__ xin, int ldx, float *__restrict__ yout, int ldy, int n,
                  StepEv ev, float *OUT, int64_t ld_row, int F)
{
    constexpr int RPB = StgCsr<C>::RPB, CAP = StgCsr<C>::CAP, LPR = 256 / RPB, EPT = CAP / 256, NV = VP == 3 ? 3 : 1;
    extern __shared__ float stg_lds[];
    float *vL = stg_lds, *xL = vL + CAP * NV;
    unsigned *cL = reinterpret_cast<unsigned *>(xL + CAP * C);
    int *rpL = reinterpret_cast<int *>(cL + (VP == 4 ? CAP : 0));
    const int trips = (n + RPB - 1) / RPB;
    for (int trip = blockIdx.x; trip < trips; trip += gridDim.x) {
        const int r0 = trip * RPB, r1 = r0 + RPB < n ? r0 + RPB : n;
        const int e0 = rowptr[r0], e1 = rowptr[r1], ne = e1 - e0;
        const bool staged = ne <= CAP;
        if (threadIdx.x <= r1 - r0) rpL[threadIdx.x] = rowptr[r0 + threadIdx.x] - e0;
        if (staged) {
            unsigned cw[EPT]; float vv[EPT][NV];
#pragma unroll
            for (int j = 0; j < EPT; ++j) {
                const int i = threadIdx.x + 256 * j;
                const bool ok = i < ne;
                cw[j] = ok ? (unsigned)__builtin_nontemporal_load(col + e0 + i) : 0u;
#pragma unroll
                for (int q = 0; q < NV; ++q) vv[j][q] = ok ? __builtin_nontemporal_load(val + ((int64_t)e0 + i) * NV + q) : 0.f;
            }
            float xr[EPT][C];
#pragma unroll
            for (int j = 0; j < EPT; ++j) {
                const float *x = xin + (int64_t)(VP == 4 ? (cw[j] & 0x1fffffffu) : cw[j]) * ldx;
#pragma unroll
                for (int c = 0; c < C; ++c) xr[j][c] = x[c];
            }
#pragma unroll
            for (int j = 0; j < EPT; ++j) {
                const int i = threadIdx.x + 256 * j;
                if (i < ne) {
#pragma unroll
                    for (int q = 0; q < NV; ++q) vL[i * NV + q] = vv[j][q];
#pragma unroll
                    for (int c = 0; c < C; ++c) xL[i * C + c] = xr[j][c];
                    if (VP == 4) cL[i] = cw[j];
                }
            }
        }
        __syncthreads();
        const int lr = threadIdx.x / LPR, row = r0 + lr, k = threadIdx.x % LPR;
        float acc[C];
#pragma unroll
        for (int c = 0; c < C; ++c) acc[c] = 0.f;
        if (row < r1) {
            const int i0 = rpL[lr], i1 = rpL[lr + 1];
            for (int i = i0 + k; i < i1; i += LPR) {
                unsigned w = 0u; float v3[NV]; const float *x;
                if (staged) {
#pragma unroll
                    for (int q = 0; q < NV; ++q) v3[q] = vL[i * NV + q];
                    if (VP == 4) w = cL[i];
                    x = xL + i * C;
                } else {
                    w = (unsigned)col[e0 + i];
#pragma unroll
                    for (int q = 0; q < NV; ++q) v3[q] = val[((int64_t)e0 + i) * NV + q];
                    x = xin + (int64_t)(VP == 4 ? (w & 0x1fffffffu) : w) * ldx;
                }
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    float v;
                    if constexpr (VP == 4) v = __uint_as_float(__float_as_uint(v3[0]) | ((w << (2 - c)) & 0x80000000u));
                    else if constexpr (VP == 3) v = v3[c % 3];
                    else v = v3[0];
                    acc[c] += v * x[c];
                }
            }
        }
#pragma unroll
        for (int c = 0; c < C; ++c) acc[c] = group_sum<LPR>(acc[c]);
        if (row < r1) {
#pragma unroll
            for (int c = 0; c < C; ++c) {
                if (c % LPR != k) continue;         // lane c % LPR of the row's group owns channel c
                float y = acc[c];
                if (MODE == 2) {
                    const float *g = OUT + (int64_t)row * ld_row + (int64_t)c * F;
                    for (int q = 0; q < ev.n; ++q)
                        if (ev.e[q].chan < 0 || ev.e[q].chan == c) y += (ev.e[q].sign & 2) ? -g[ev.e[q].slot] : g[ev.e[q].slot];
                }
                yout[(int64_t)row * ldy + c] = y;
                if (MODE == 1) {
                    float *o = OUT + (int64_t)row * ld_row + (int64_t)c * F;
                    for (int q = 0; q < ev.n; ++q)
                        if (ev.e[q].chan < 0 || ev.e[q].chan == c) emit_one(o, y, ev.e[q].slot, ev.e[q].sign);
                }
            }
        }
        __syncthreads();
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

static bool step_lanes(const char *name)          // ESCAT_CSR_STEP / ESCAT_BSR3_STEP = lanes: the round-1 lane-group kernels
{
    const char *e = getenv(name);                  // read per launch: tests switch it inside one process
    return e && !strcmp(e, "lanes");
}

template <int C, int MODE, int VP>
static void launch_csr_step_staged(const int *rp, const int *ci, const float *v, const float *xin, int ldx, float *y, int ldy,
                                   int64_t n, const StepEv &ev, float *OUT, int64_t ld_row, int F, hipStream_t st)
{
    constexpr int lds = stg_csr_lds_bytes<C, VP>();
    // static trip order: every workgroup of the grid must be resident, so the grid is what the device holds
    static const int per_cu = [] {
        int occ = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_csr_step_staged<C, MODE, VP>, 256, lds) != hipSuccess || occ < 1) occ = 1;
        return occ;
    }();
    const int64_t trips = (n + StgCsr<C>::RPB - 1) / StgCsr<C>::RPB, cap = (int64_t)escat_cu_count() * per_cu;
    hipLaunchKernelGGL((k_csr_step_staged<C, MODE, VP>), dim3((unsigned)(trips < cap ? (trips < 1 ? 1 : trips) : cap)), dim3(256),
                       lds, st, rp, ci, v, xin, ldx, y, ldy, (int)n, ev, OUT, ld_row, F);
}

template <int MODE>
static int launch_csr_step(int C, int G, const int *rp, const int *ci, const float *v, const float *xin, int ldx,
                           float *y, int ldy, int64_t n, const StepEv &ev, float *OUT, int64_t ld_row, int F, hipStream_t st,
                           int planes = 1)
{
    // one value per entry: staged (cfg2-shaped single product 0.123 -> 0.102 ms).  The per-channel formats keep the
    // lane-group kernel, whose four-entries-per-lane batching is faster for them (cfg3 factored chain 8.3 vs 9.3 ms staged).
    if (planes != 3 && planes != 4 && !step_lanes("ESCAT_CSR_STEP")) {
        switch (C) {
#define CASE(c) case c: launch_csr_step_staged<c, MODE, 1>(rp, ci, v, xin, ldx, y, ldy, n, ev, OUT, ld_row, F, st); break;
            CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(7) CASE(8)
#undef CASE
            default: return fail(ESCAT_ERANGE, "C=%d outside [1,%d]", C, ESCAT_MAX_CHANNELS);
        }
        return ESCAT_OK;
    }
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
                             int64_t nb, int64_t nnzb, const StepEv &ev, float *OUT, int F, hipStream_t st)
{
    if (!step_lanes("ESCAT_BSR3_STEP")) {
        // static trip order: every workgroup of the grid must be resident, so the grid is what the device holds
        static const int per_cu = [] {
            int occ = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_bsr3_step_staged<MODE>, 256, STG_LDS_BYTES) != hipSuccess || occ < 1) occ = 1;
            return occ;
        }();
        const int64_t trips = (nb + STG_RPB - 1) / STG_RPB, cap = (int64_t)escat_cu_count() * per_cu;
        hipLaunchKernelGGL((k_bsr3_step_staged<MODE>), dim3((unsigned)(trips < cap ? (trips < 1 ? 1 : trips) : cap)), dim3(256),
                           STG_LDS_BYTES, st, rp, ci, blk, xin, y, (int)nb, nnzb, ev, OUT, F);
        return;
    }
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
    launch_bsr3_step<0>(pick_g(nb, nnzb), rowptr, col, blk, x, y, nb, nnzb, ev, nullptr, 0, S(stream));
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
    if ((!sell->val && !sell->half_val && !sell->ent_of_slot) || !sell->lc || !sell->lane_meta) return fail(ESCAT_EINVAL, "escat_sell has a NULL member");
    sd->ent = (!sell->val && !sell->half_val) ? (const signed char *)sell->ent_of_slot : nullptr;
    sd->rowptr = nullptr;
    if (((uintptr_t)sell->val & 15) != 0) return fail(ESCAT_EINVAL, "escat_sell.val must be 16-byte aligned");
    if ((sell->half_val != nullptr) != (sell->half_aux != nullptr)) return fail(ESCAT_EINVAL, "escat_sell.half_val and half_aux go together");
    sd->half_val = sell->half_val; sd->half_aux = sell->half_aux;
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
    if (sell->layout != 0 && sell->layout != 1) return fail(ESCAT_EINVAL, "escat_sell.layout=%d: 0 or 1", sell->layout);
    sd->layout = sell->layout; sd->pos_of_row = sell->pos_of_row;
    if (sd->frames && sd->planes < 3) return fail(ESCAT_EINVAL, "escat_sell.frames goes with val_planes = 3");
    return ESCAT_OK;
}

int escat_cu_count()
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
size_t escat_sell_lds_bytes(int C, int CP, int max_rows, int F, bool bwd, int want, int *early, int orders)
{
    const size_t nv = (size_t)2 * sell_padR(max_rows) * CP;
    const size_t slab = ((size_t)max_rows * C * (F + (!bwd && orders ? 1 : 0)) + 3) & ~(size_t)3;
    const size_t base = (2 * nv + slab) * sizeof(float);
    if (!bwd) {                                   // + the staged input + the row -> position table of layout-1 packs
        *early = 1;
        return base + ((((size_t)max_rows * C + 3) & ~(size_t)3)) * sizeof(float) + (((size_t)max_rows + 3) & ~(size_t)3) * sizeof(int);
    }
    const size_t two = base + slab * sizeof(float);
    if (two * want <= LDS_BUDGET) { *early = 1; return two; }
    *early = 0;
    return base;
}

template <bool BWD>
static int launch_csr_sell(int C, int max_rows, const int *ci, const float *v, int n_parts, const SellDev &sd,
                           const float *IN, float *RES, int T, const EvTable &tab, int F, hipStream_t st)
{
    if (sd.planes >= 3) {
        if (C != 3) return fail(ESCAT_EINVAL, "a pack with one value per channel (val_planes = 3, 4) needs C = 3, got %d", C);
        return escat_launch_csrv3_sell(BWD, max_rows, ci, v, n_parts, sd, IN, RES, T, tab, F, st);
    }
    switch (C) {
        case 1: case 2: return escat_launch_csr_sell_c12(BWD, C, max_rows, ci, v, n_parts, sd, IN, RES, T, tab, F, st);
        case 3: case 4: return escat_launch_csr_sell_c34(BWD, C, max_rows, ci, v, n_parts, sd, IN, RES, T, tab, F, st);
        case 5: case 6: return escat_launch_csr_sell_c56(BWD, C, max_rows, ci, v, n_parts, sd, IN, RES, T, tab, F, st);
        case 7: case 8: return escat_launch_csr_sell_c78(BWD, C, max_rows, ci, v, n_parts, sd, IN, RES, T, tab, F, st);
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
    rc = build_table(events, n_events, T, F, C, &tab, BWD); if (rc) return rc;
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
            const size_t lds_s = escat_sell_lds_bytes(C, CP, max_part_rows, F, BWD, max_part_rows <= 256 ? 4 : 2, &early_, sell ? sell->orders : 0);
            if (max_part_rows <= 512 && packed && force != 2 && lds_s <= LDS_BUDGET) {
                SellDev sd;
                rc = sell_to_dev(sell, &sd); if (rc) return rc;
                if (BWD && sd.orders) return fail(ESCAT_EINVAL, "escat_sell.orders must be 0 for a backward entry point");
                if (!sd.val) return fail(ESCAT_EINVAL, "escat_sell.val is NULL (CSR packs have no half-valued or direct form)");
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

// ---- both tracks in one launch (escat_sell_both.hip)
// The event table as a program of event steps; refuses (ESCAT_ERANGE: the caller runs the two chains separately) whatever
// is not "every slot is one power or the difference of two, one band open at a time, the same for all channels".
static int build_both_prog(const EvTable &tab, int T, int F, bool bwd, BothProg *pg)
{
    static thread_local signed char first_at[ESCAT_MAX_SLOTS], count[ESCAT_MAX_SLOTS];
    for (int f = 0; f < F; ++f) { first_at[f] = -1; count[f] = 0; }
    if (T > 127) return fail(ESCAT_ERANGE, "both-tracks chain: T=%d > 127", T);
    for (int k = 0; k < tab.n; ++k) {
        const Ev &e = tab.e[k];
        if (e.chan >= 0) return fail(ESCAT_ERANGE, "both-tracks chain: per-channel events are not supported");
        if (e.sign & EV_ABS) return fail(ESCAT_ERANGE, "both-tracks chain: ESCAT_EVENT_ABS is not supported");
        if (++count[e.slot] > 2) return fail(ESCAT_ERANGE, "both-tracks chain: slot %d has more than two events", e.slot);
        if (count[e.slot] == 1) first_at[e.slot] = (signed char)e.step;
        else if (first_at[e.slot] == e.step) return fail(ESCAT_ERANGE, "both-tracks chain: slot %d takes the same power twice", e.slot);
    }
    pg->n = 0;
    int open_slot = -1;
    for (int k = 0; k < tab.n;) {
        int k1 = k;
        while (k1 < tab.n && tab.e[k1].step == tab.e[k].step) ++k1;
        if (pg->n >= BOTH_MAX_STEPS) return fail(ESCAT_ERANGE, "both-tracks chain: more than %d steps with events", BOTH_MAX_STEPS);
        if (k1 - k > (bwd ? 2 : BOTH_MAX_TERMS)) return fail(ESCAT_ERANGE, "both-tracks chain: %d events at step %d", k1 - k, tab.e[k].step);
        BothStep &st = pg->s[pg->n++];
        st.step = tab.e[k].step; st.n = k1 - k;
        int closes = 0, opens = 0, directs = 0, opened = -1;
        for (int j = k; j < k1; ++j) {
            const Ev &e = tab.e[j];
            int role;
            if (count[e.slot] == 1) { role = 2; ++directs; }
            else if (e.sign & EV_STORE) { role = 0; ++opens; opened = e.slot; }
            else {
                role = 1; ++closes;
                if (open_slot != e.slot) return fail(ESCAT_ERANGE, "both-tracks chain: slot %d closes while slot %d is open", e.slot, open_slot);
                open_slot = -1;
            }
            st.t[j - k].slot = e.slot;
            st.t[j - k].kind = role | ((e.sign & 2) ? 4 : 0);
        }
        if (closes > 1 || opens > 1 || directs > 1) return fail(ESCAT_ERANGE, "both-tracks chain: step %d has two events of a kind", st.step);
        if (opens) {
            if (open_slot >= 0) return fail(ESCAT_ERANGE, "both-tracks chain: two bands open at step %d", st.step);
            open_slot = opened;
        }
        k = k1;
    }
    if (open_slot >= 0) return fail(ESCAT_EINVAL, "both-tracks chain: slot %d never closes", open_slot);
    return ESCAT_OK;
}

template <bool BWD>
static int chain_both(const int32_t *rowptr, const int32_t *col_signed, const float *mag, int64_t n, int64_t nnz,
                      const float *INs, const float *INv, int T, const escat_event *events, int n_events, int F,
                      float *RESs, float *RESv, const int32_t *part_ptr, int n_parts, int max_part_rows,
                      const escat_sell *sell, void *stream)
{
    int rc = check_csr(rowptr, col_signed, mag, n, nnz); if (rc) return rc;
    if (!INs || !INv || !RESs || !RESv) return fail(ESCAT_EINVAL, "input or output pointer is NULL");
    if (T < 1) return fail(ESCAT_ERANGE, "both-tracks chain: T=%d < 1", T);
    if (!sell || !sell->val || sell->val_planes != 4 || !sell->frames || sell->layout != 1 || !sell->pos_of_row)
        return fail(ESCAT_EINVAL, "both-tracks chain: needs a sign-bit pack (val_planes = 4, layout 1) with frames and pos_of_row");
    if (sell->orders) return fail(ESCAT_EINVAL, "both-tracks chain: escat_sell.orders must be 0");
    if (!part_ptr || n_parts <= 0 || max_part_rows <= 0 || max_part_rows > 512)
        return fail(ESCAT_ERANGE, "both-tracks chain: parts of 1..512 rows (got max %d)", max_part_rows);
    if ((int64_t)n * 3 * F >= ((int64_t)1 << 31)) return fail(ESCAT_ERANGE, "both-tracks chain: n * 3 * F >= 2^31");
    EvTable tab;
    rc = build_table(events, n_events, T, F, 3, &tab, BWD); if (rc) return rc;
    BothProg pg;
    rc = build_both_prog(tab, T, F, BWD, &pg); if (rc) return rc;
    if (n == 0) return ESCAT_OK;
    SellDev sd;
    rc = sell_to_dev(sell, &sd); if (rc) return rc;
    rc = escat_launch_both_sell(BWD, max_part_rows, col_signed, mag, sd.part_list ? sell->n_list : n_parts, sd, INs, INv, RESs, RESv,
                                T, pg, F, S(stream));
    if (rc) return rc;
    CHECK_LAUNCH("k_chain_both");
    return ESCAT_OK;
}

extern "C" int escat_chain_both_f32(const int32_t *rowptr, const int32_t *col_signed, const float *mag, int64_t n, int64_t nnz,
                                    const float *Xs, const float *Xv, int T, const escat_event *events, int n_events, int F,
                                    float *OUTs, float *OUTv, const int32_t *part_ptr, int n_parts, int max_part_rows,
                                    const escat_sell *sell, void *stream)
{
    return chain_both<false>(rowptr, col_signed, mag, n, nnz, Xs, Xv, T, events, n_events, F, OUTs, OUTv, part_ptr, n_parts,
                             max_part_rows, sell, stream);
}

extern "C" int escat_chain_both_bwd_f32(const int32_t *rowptr_t, const int32_t *col_signed_t, const float *mag_t, int64_t n, int64_t nnz,
                                        const float *Gs, const float *Gv, int T, const escat_event *events, int n_events, int F,
                                        float *dXs, float *dXv, const int32_t *part_ptr, int n_parts, int max_part_rows,
                                        const escat_sell *sell, void *stream)
{
    return chain_both<true>(rowptr_t, col_signed_t, mag_t, n, nnz, Gs, Gv, T, events, n_events, F, dXs, dXv, part_ptr, n_parts,
                            max_part_rows, sell, stream);
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
    rc = build_table(events, n_events, T, F, 1, &tab, BWD); if (rc) return rc;
    if (nb == 0) return ESCAT_OK;
    hipStream_t st = S(stream);
    const int g = pick_g(nb, nnzb);
    const int64_t n = 3 * nb;

    const int force = chain_path_override();
    if (part_ptr && n_parts > 0 && max_part_rows > 0 && force < 3) {
        const size_t lds = (size_t)max_part_rows * 3 * (2 + F) * sizeof(float);
        int early_ = 0;
        const size_t lds_s = escat_sell_lds_bytes(3, 4, max_part_rows, F, BWD, 1, &early_, sell ? sell->orders : 0);
        if (lds <= LDS_BUDGET && lds_s <= LDS_BUDGET && max_part_rows <= 512 && sell && force != 2) {
            SellDev sd;
            rc = sell_to_dev(sell, &sd); if (rc) return rc;
            if (BWD && sd.orders) return fail(ESCAT_EINVAL, "escat_sell.orders must be 0 for a backward entry point");
            sd.rowptr = rowptr;
            // half-valued pack (escat_sell_bsr3h.hip) when given and instantiated for the capacity; ESCAT_Q_HALF=0: ignore it
            static const bool no_half = getenv("ESCAT_Q_HALF") && !strcmp(getenv("ESCAT_Q_HALF"), "0");
            if (sd.half_val && sd.dmax <= 20 && !(no_half && sd.val))
                rc = escat_launch_bsr3h_sell(BWD, max_part_rows, col, blk, sd.part_list ? sell->n_list : n_parts, sd, IN, RES, T, tab, F, st);
            else if (!sd.val && !sd.ent) return fail(ESCAT_EINVAL, "escat_sell.val is NULL and the half-valued pack cannot be used (dmax=%d)", sd.dmax);
            else
            rc = escat_launch_bsr3_sell(BWD, max_part_rows, col, blk, sd.part_list ? sell->n_list : n_parts, sd, IN, RES, T, tab, F, st);
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
            launch_bsr3_step<1>(g, rowptr, col, blk, xin, y, nb, nnzb, se, RES, F, st);
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
            launch_bsr3_step<2>(g, rowptr, col, blk, xin, y, nb, nnzb, se, Gm, F, st);
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


