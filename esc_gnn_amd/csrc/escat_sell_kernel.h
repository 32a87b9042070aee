// escat_sell_kernel.h -- the register-resident chain kernel (k_chain_sell) and its operator flavours; included by the
// escat_sell_*.hip translation units, which instantiate it per operator family (parallel builds).
#ifndef ESCAT_SELL_KERNEL_H
#define ESCAT_SELL_KERNEL_H
#include "escat_dev.h"

// debug build only (-DESCAT_STAMPS, see tools/stamps.py): wall-clock stamps of the phases of every part
#ifdef ESCAT_STAMPS
static __device__ unsigned long long *g_stamps = nullptr;
#define STAMP(part, i) do { if (g_stamps && threadIdx.x == 0) g_stamps[(size_t)(part) * 16 + (i)] = wall_clock64(); } while (0)
#define STAMPW(part, i, tid) do { if (g_stamps && threadIdx.x == (tid)) g_stamps[(size_t)(part) * 16 + (i)] = wall_clock64(); } while (0)
#define ESCAT_STAMP_SETTER(name) extern "C" int name(void *p) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), &p, sizeof(p)); }
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

// ---- LDS gathers with an explicit software pipeline.  The loads are inline asm, so the compiler neither narrows them
// (a float4 gather whose pad lane is dead becomes ds_read_b96: 8 LDS cycles instead of 4) nor places its own waits: with
// compiler-visible loads the s_waitcnt it derived alternated lgkmcnt(1) / lgkmcnt(0) -- at most two gathers in flight,
// which leaves the step loop bound by LDS latency (2 waves per SIMD).  lds_wait<N>() returns once at most N later LDS
// operations of the wave are outstanding; it is tied to the destination registers so that no use moves above it.
template <int CP> struct LdsRow;
template <> struct LdsRow<1> { typedef float T; };
template <> struct LdsRow<2> { typedef vf2 T; };
template <> struct LdsRow<4> { typedef vf4 T; };
__device__ __forceinline__ void lds_gather(float &d, unsigned a) { asm volatile("ds_read_b32 %0, %1" : "=v"(d) : "v"(a)); }
__device__ __forceinline__ void lds_gather(vf2 &d, unsigned a) { asm volatile("ds_read_b64 %0, %1" : "=v"(d) : "v"(a)); }
__device__ __forceinline__ void lds_gather(vf4 &d, unsigned a) { asm volatile("ds_read_b128 %0, %1" : "=v"(d) : "v"(a)); }
template <int N, class T> __device__ __forceinline__ void lds_wait(T &d) { asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(d) : "n"(N)); }
__device__ __forceinline__ unsigned lds_addr(const float *p) { return (unsigned)(uintptr_t)(lds_void_t *)p; }

// ---- the operator of one row, in registers.  Two flavours with one interface:
//   load(sd, doff, poff, w, lane)  issue the (coalesced) loads of this lane's slots
//   gather(xa, w, acc)          acc += sum over slots of value * x[LDS row]   (one step)
//   tail(...)                   entries beyond dmax, streamed
struct __attribute__((packed, aligned(4))) f4u_k { float x, y, z, w; };    // a 16-byte load that is only dword-aligned

template <int DMAX, bool DIRECT_ = false> struct OpBsr3 {
    static constexpr bool DIRECT = DIRECT_;       // no value pack: the blocks come straight from the canonical array (escat_sell.ent_of_slot)
    static constexpr bool FRAMES = false;
    static constexpr bool POS = true;             // takes layout-1 packs (LDS rows = owner positions)
    static constexpr int C = 3, CP = 4, MINW = 1;
#ifndef ESCAT_Q_RING
#define ESCAT_Q_RING 4
#endif
    static constexpr int RING = ESCAT_Q_RING;     // LDS gathers in flight per lane (4 VGPRs each)
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
    // DIRECT: slot d of this lane holds entry e0 + ent[(doff + d) * 64 + lane] of the canonical (nnz, 3, 3) array `blk`
    // (e0 = first entry of the lane's row); a padded slot (ent < 0, or a lane without a row) reads block 0 and is zeroed.
    // 36 contiguous bytes per block and lane: strided over the lanes, every byte of every line used once.
    __device__ __forceinline__ void load_direct(const SellDev &sd, const float *__restrict__ blk, int doff, int poff, int w, int lane, int e0)
    {
        const signed char *eb = sd.ent + (int64_t)doff * 64 + lane;
        const int wl = w > 0 ? w - 1 : 0;
        const int pl = w > 0 ? ((w + SLOT_GROUP - 1) / SLOT_GROUP) * (SLOT_GROUP / 2) - 1 : 0;
        int j[DMAX];
#pragma unroll
        for (int d = 0; d < DMAX; ++d) j[d] = w > 0 ? (int)eb[(d < wl ? d : wl) * 64] : -1;
#pragma unroll
        for (int j2 = 0; j2 < (DMAX + 1) / 2; ++j2)
            lc[j2] = sd.lc[(int64_t)(poff + (j2 < pl ? j2 : pl)) * 64 + lane];
#pragma unroll
        for (int d = 0; d < DMAX; ++d) {
            const float *bb = blk + (int64_t)(j[d] < 0 ? 0 : e0 + j[d]) * 9;
            const f4u_k lo = *reinterpret_cast<const f4u_k *>(bb), hi = *reinterpret_cast<const f4u_k *>(bb + 4);
            const float b8 = bb[8];
            const bool z = j[d] < 0;
            b0[d] = vf4{z ? 0.f : lo.x, z ? 0.f : lo.w, z ? 0.f : lo.y, z ? 0.f : hi.x};     // b00 b10 b01 b11
            b1[d] = vf4{z ? 0.f : lo.z, z ? 0.f : hi.y, z ? 0.f : hi.z, z ? 0.f : hi.w};     // b02 b12 b20 b21
            b2[d] = z ? 0.f : b8;
        }
    }
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
    template <int W, class Mid>
    __device__ __forceinline__ void gather_w(const float *xa, vf2 &a01, vf2 &t2, float &a2, Mid mid) const
    {
        constexpr int D = RING;
        vf4 xv[D];
        const unsigned base = lds_addr(xa);
#pragma unroll
        for (int d = 0; d < D && d < W; ++d) lds_gather(xv[d], base + row_of(d) * 16u);   // a row is 4 floats
        mid();                                    // work that needs no gather result runs under the first gathers' latency
#pragma unroll
        for (int d = 0; d < W; ++d) {
            // gather d has landed once at most min(D-1, W-1-d) younger ones are outstanding (LDS returns in order)
            if (W - 1 - d >= D - 1) lds_wait<D - 1>(xv[d % D]);
            else if (W - 1 - d == 6) lds_wait<6>(xv[d % D]);
            else if (W - 1 - d == 5) lds_wait<5>(xv[d % D]);
            else if (W - 1 - d == 4) lds_wait<4>(xv[d % D]);
            else if (W - 1 - d == 3) lds_wait<3>(xv[d % D]);
            else if (W - 1 - d == 2) lds_wait<2>(xv[d % D]);
            else if (W - 1 - d == 1) lds_wait<1>(xv[d % D]);
            else lds_wait<0>(xv[d % D]);
            const vf4 x = xv[d % D];
            a01 = __builtin_elementwise_fma(b0[d].xy, x.xx, a01);
            a01 = __builtin_elementwise_fma(b0[d].zw, x.yy, a01);
            a01 = __builtin_elementwise_fma(b1[d].xy, x.zz, a01);
            t2 = __builtin_elementwise_fma(b1[d].zw, x.xy, t2);
            a2 = fmaf(b2[d], x.z, a2);
            if (d + D < W) {
                // the slot just consumed takes the next gather; the empty asm statements pin this slot's FMAs before it
                // (volatile asm keeps its order: without them the scheduler sinks the FMAs and renames the ring away)
                asm volatile("" :: "v"(a01), "v"(t2), "v"(a2));
                lds_gather(xv[d % D], base + row_of(d + D) * 16u);
            }
        }
    }
    template <int W, class Mid>
    __device__ __forceinline__ void dispatch(int w2, const float *xa, vf2 &a01, vf2 &t2, float &a2, Mid mid) const
    {
        if constexpr (W <= DMAX) {
            if (w2 == W / 2) gather_w<W>(xa, a01, t2, a2, mid);
            else dispatch<W + 2>(w2, xa, a01, t2, a2, mid);
        }
    }
    template <class Mid>
    __device__ __forceinline__ void gather(const float *xa, int w, float (&a)[3], Mid mid) const
    {
        vf2 a01 = {0.f, 0.f}, t2 = {0.f, 0.f};    // (y0, y1) and the two-column partial sums of y2
        float a2 = 0.f;
        dispatch<2>((w + 1) >> 1, xa, a01, t2, a2, mid);
        a[0] += a01.x; a[1] += a01.y; a[2] += a2 + (t2.x + t2.y);
    }
    static __device__ __forceinline__ void tail(const float *__restrict__ blk, const int *__restrict__ col, int e0, int n,
                                                int gs, const float *xa, float (&a)[3], const int *__restrict__ pos)
    {
        for (int e = e0; e < e0 + n; ++e) {       // rows longer than the pack: from L2, every step
            const float *bb = blk + (int64_t)e * 9;
            const float *x = xa + (pos ? pos[col[e]] : col[e] - gs) * 4;
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
    static constexpr bool POS = true;
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
    static constexpr int RING = CC <= 4 ? 6 : 3;  // gathers in flight per lane (CP / 4 or 2 x 4 VGPRs each)
    template <int W, class Mid>
    __device__ __forceinline__ void gather_w(const float *xa, float (&acc)[CC], Mid mid) const      // see OpBsr3::gather_w
    {
        constexpr int D = RING;
        constexpr int PER = CP == 8 ? 2 : 1;     // LDS operations per gather
        typedef typename LdsRow<(CP == 8 ? 4 : CP)>::T row_t;
        row_t xv[D][PER];
        const unsigned base = lds_addr(xa);
#pragma unroll
        for (int d = 0; d < D && d < W; ++d) {
            lds_gather(xv[d][0], base + row_of(d) * (unsigned)(CP * 4));
            if (PER == 2) lds_gather(xv[d][PER - 1], base + row_of(d) * (unsigned)(CP * 4) + 16u);
        }
        mid();
#pragma unroll
        for (int d = 0; d < W; ++d) {
            const int later = (W - 1 - d < D - 1 ? W - 1 - d : D - 1) * PER;     // younger LDS operations still allowed in flight
            row_t &last = xv[d % D][PER - 1];
            if (later >= 10) lds_wait<10>(last);
            else if (later == 8) lds_wait<8>(last);
            else if (later == 6) lds_wait<6>(last);
            else if (later == 5) lds_wait<5>(last);
            else if (later == 4) lds_wait<4>(last);
            else if (later == 3) lds_wait<3>(last);
            else if (later == 2) lds_wait<2>(last);
            else if (later == 1) lds_wait<1>(last);
            else lds_wait<0>(last);
            if (PER == 2) asm volatile("" : "+v"(xv[d % D][0]));            // the first half landed before the second
            float x[CP];
            if constexpr (CP == 1) x[0] = xv[d % D][0];
            else if constexpr (CP == 2) { x[0] = xv[d % D][0].x; x[1] = xv[d % D][0].y; }
            else {
#pragma unroll
                for (int h = 0; h < PER; ++h) { x[4 * h] = xv[d % D][h].x; x[4 * h + 1] = xv[d % D][h].y; x[4 * h + 2] = xv[d % D][h].z; x[4 * h + 3] = xv[d % D][h].w; }
            }
#pragma unroll
            for (int c = 0; c < CC; ++c) acc[c] = fmaf(v[d], x[c], acc[c]);
            if (d + D < W) {
#pragma unroll
                for (int c = 0; c < CC; ++c) asm volatile("" :: "v"(acc[c]));      // see OpBsr3::gather_w
                lds_gather(xv[d % D][0], base + row_of(d + D) * (unsigned)(CP * 4));
                if (PER == 2) lds_gather(xv[d % D][PER - 1], base + row_of(d + D) * (unsigned)(CP * 4) + 16u);
            }
        }
    }
    template <int W, class Mid>
    __device__ __forceinline__ void dispatch(int w2, const float *xa, float (&acc)[CC], Mid mid) const
    {
        if constexpr (W <= DMAX) {
            if (w2 == W / 2) gather_w<W>(xa, acc, mid);
            else dispatch<W + 2>(w2, xa, acc, mid);
        }
    }
    template <class Mid>
    __device__ __forceinline__ void gather(const float *xa, int w, float (&acc)[CC], Mid mid) const { dispatch<2>((w + 1) >> 1, xa, acc, mid); }
    static __device__ __forceinline__ void tail(const float *__restrict__ val, const int *__restrict__ col, int e0, int n,
                                                int gs, const float *xa, float (&acc)[CC], const int *__restrict__ pos)
    {
        for (int e = e0; e < e0 + n; ++e) {
            const float vv = val[e];
            const float *x = xa + (pos ? pos[col[e]] : col[e] - gs) * CP;
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
    static constexpr bool POS = true;             // layout-1 packs (LDS rows = owner positions), as OpCsr / OpBsr3 (round 3)
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
    template <class Mid>
    __device__ __forceinline__ void gather(const float *xa, int w, float (&acc)[3], Mid mid) const { mid(); dispatch<2>((w + 1) >> 1, xa, acc); }
    static __device__ __forceinline__ void tail(const float *__restrict__ val, const int *__restrict__ col, int e0, int n,
                                                int gs, const float *xa, float (&acc)[3], const int *__restrict__ pos)
    {
        for (int e = e0; e < e0 + n; ++e) {
            if constexpr (SIGNED) {          // global arrays in the packed form of k_csr_step<.., 4>
                const unsigned cw = (unsigned)col[e], vb = __float_as_uint(val[e]);
                const int cj = (int)(cw & 0x1fffffffu);
                const float *x = xa + (pos ? pos[cj] : cj - gs) * 4;
#pragma unroll
                for (int c = 0; c < 3; ++c) acc[c] = fmaf(__uint_as_float(vb | ((cw << (2 - c)) & 0x80000000u)), x[c], acc[c]);
            } else {
                const float *x = xa + (pos ? pos[col[e]] : col[e] - gs) * 4;
#pragma unroll
                for (int c = 0; c < 3; ++c) acc[c] = fmaf(val[(int64_t)e * 3 + c], x[c], acc[c]);
            }
        }
    }
};


typedef const __attribute__((address_space(1))) void glb_void_t;

template <class Op> struct is_direct { static constexpr bool value = false; };
template <int DMAX> struct is_direct<OpBsr3<DMAX, true>> { static constexpr bool value = true; };

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

// the same for layout-1 packs: every owner builds its own row of the first vector (at its position xr, copy B at rB) from
// the staged input, which is in the operator's row order (row r)
// Forward: the slab row of a node is its POSITION too (posL[r] = position of row r, for the epilogue): the lanes of a
// wave then touch the slab at a constant stride instead of at the sorted rows' scattered offsets (the events' LDS bank
// conflicts were 16 % of the P chain's LDS cycles).
template <bool BWD, int NT>
__device__ __forceinline__ int chain_prologue_pos(float *xa, float *xb, float *slab, const float *inb, int rows, bool own, int r,
                                                  int xr, int rB, int C, int CP, int F, int T, const EvTable &tab, int Fs, int so,
                                                  int *posL = nullptr)
{
    const int tid = threadIdx.x;
    int p;
    if (tid < CP) { xa[rows * CP + tid] = 0.f; xb[rows * CP + tid] = 0.f; }
    if (!BWD) {
        int p1 = 0; while (p1 < tab.n && tab.e[p1].step == 0) ++p1;
        if (own) {
            posL[r] = xr;
            for (int c = 0; c < C; ++c) {
                const int i = xr * C + c; const float y = inb[r * C + c];
                xa[xr * CP + c] = y;
                xa[rB * CP + c] = y;
                if (so) slab[i * Fs] = y;
                for (int k = 0; k < p1; ++k)
                    if (tab.e[k].chan < 0 || tab.e[k].chan == c) emit_one(slab + i * Fs + so, y, tab.e[k].slot, tab.e[k].sign);
            }
        }
        p = p1;
    } else {
        int p0 = tab.n; while (p0 > 0 && tab.e[p0 - 1].step == T) --p0;
        if (own)
            for (int c = 0; c < C; ++c) {
                const int i = r * C + c; float y = 0.f;
                for (int k = p0; k < tab.n; ++k)
                    if (tab.e[k].chan < 0 || tab.e[k].chan == c) y += (tab.e[k].sign & 2) ? -slab[i * F + tab.e[k].slot] : slab[i * F + tab.e[k].slot];
                xa[xr * CP + c] = y;
                xa[rB * CP + c] = y;
            }
        p = p0;
    }
    __syncthreads();
    return p;
}

// in-place rotation of a part's 3-vectors in LDS by the rows' local frames (global, L1/L2-resident: 36 B per row):
// buf[r * rs + c * cs + f] <- sum_c' O_r[c][c'] (TR: O_r[c'][c]) buf[r * rs + c' * cs + f]; rows are independent.
// posL (layout 1, forward slab): the buffer's row of node r is its position posL[r]
template <bool TR, int NT>
__device__ __forceinline__ void rotate_rows(float *buf, const float *__restrict__ fr, int rows, int F, int rs, int cs,
                                            const int *posL = nullptr)
{
    for (int r = threadIdx.x; r < rows; r += NT) {             // one thread per row: its frame is loaded once
        const float *o = fr + r * 9;
        const float o0 = o[0], o1 = o[1], o2 = o[2], o3 = o[3], o4 = o[4], o5 = o[5], o6 = o[6], o7 = o[7], o8 = o[8];
        float *x = buf + (posL ? posL[r] : r) * rs;
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

// one 3-vector out of its row's local frame (the arithmetic of rotate_rows<false>), in registers
__device__ __forceinline__ void frame_to_global(float *v, const float *__restrict__ o)
{
    const float x0 = v[0], x1 = v[1], x2 = v[2];
    v[0] = fmaf(o[2], x2, fmaf(o[1], x1, o[0] * x0));
    v[1] = fmaf(o[5], x2, fmaf(o[4], x1, o[3] * x0));
    v[2] = fmaf(o[8], x2, fmaf(o[7], x1, o[6] * x0));
}

// VEC2: copy two floats at a time in the permuted forward case (not for the BSR-3 kernel: at its 256-register limit the
// second code path costs the step loop 4 spilled dwords)
template <bool BWD, int NT, bool VEC2 = false>
__device__ __forceinline__ void chain_epilogue(const float *xa, const float *slab, float *__restrict__ RES, int64_t first_row,
                                               int rows, int C, int CP, int F, const int *posL = nullptr)
{
    const int tid = threadIdx.x;
    const int nc = rows * C;
    if (!BWD) {
        float *og = RES + first_row * C * F;
        if (VEC2 && posL && ((C * F) & 1) == 0) { // the same two floats at a time (rows of an even number of floats)
            const int cf = C * F, h = cf >> 1;
            int row = tid / h, rest = tid % h;
            const int drow = NT / h, drest = NT % h;
            float2 *og2 = reinterpret_cast<float2 *>(og);
            for (int i = tid; i < rows * h; i += NT) {
                og2[i] = *reinterpret_cast<const float2 *>(slab + posL[row] * cf + 2 * rest);
                row += drow; rest += drest;
                if (rest >= h) { rest -= h; ++row; }
            }
        } else if (posL) {                        // slab rows are positions: output element (row, rest) <- slab[posL[row]]
            const int cf = C * F;
            int row = tid / cf, rest = tid % cf;
            const int drow = NT / cf, drest = NT % cf;
            for (int i = tid; i < nc * F; i += NT) {
                og[i] = slab[posL[row] * cf + rest];
                row += drow; rest += drest;
                if (rest >= cf) { rest -= cf; ++row; }
            }
        } else
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
                                                      int64_t first_row, int rows, int C, int F, int second, const int *posL = nullptr)
{
    const int wtot = 1 + F + (second ? F * (F - 1) / 2 : 0), Fs = F + 1;
    const int nrc = rows * C, per = tile_floats / wtot;
    float *og = RES + first_row * C * wtot;
    for (int c0 = 0; c0 < nrc; c0 += per) {
        const int n = nrc - c0 < per ? nrc - c0 : per;
        for (int rc = threadIdx.x; rc < n; rc += NT) {
            const int rcg = c0 + rc;
            const float *srow = slab + (posL ? posL[rcg / C] * C + rcg % C : rcg) * Fs;
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
__global__ void __launch_bounds__(NT, (Op::MINW == 1 && NT == 256) ? 2 : Op::MINW)      // compiled into the plain kernel, its code costs the step loop registers)
k_chain_sell(const int *__restrict__ col, const float *__restrict__ val, SellDev sd, const float *__restrict__ IN,
             float *__restrict__ RES, int T, EvTable tab, int F, int n_parts, int max_rows, int early, int stagger)
{
    constexpr int C = Op::C, CP = Op::CP;
    if (stagger) {                                // experiment: de-phase the workgroups that share a CU
        const bool late = stagger > 0 ? (blockIdx.x >= (gridDim.x >> 1)) : (blockIdx.x & 1);
        if (late) {
            const unsigned long long t0 = wall_clock64(), dt = (unsigned long long)(stagger > 0 ? stagger : -stagger);
            while (wall_clock64() - t0 < dt) __builtin_amdgcn_s_sleep(32);
        }
    }
    constexpr bool POS = Op::POS;
    extern __shared__ __align__(16) float smem[];
    const int nv = 2 * sell_padR(max_rows) * CP;                 // two copies per vector, a multiple of 32 floats
    const int so = ORD ? 1 : 0, Fs = F + so;                    // fused orders: slab rows are [x | bank]
    const int slab_sz = (max_rows * C * Fs + 3) & ~3;
    float *xa = smem, *xb = smem + nv, *slab = smem + 2 * nv;
    float *inb = (BWD && !early) ? slab : slab + slab_sz;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int in_w = BWD ? C * F : C;                            // input floats per row
    // layout 1, forward: slab rows are positions; posL (row -> position, written by the prologue) sits behind the staged input
    int *posL = (POS && !BWD) ? reinterpret_cast<int *>(slab + slab_sz + ((max_rows * C + 3) & ~3)) : nullptr;

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
        if constexpr (is_direct<Op>::value) {
            const int rr = (rt & 0x3ff) == 0x3ff ? -1 : (rt & 0x3ff);
            op.load_direct(sd, val, doff, poff, w, lane, rr >= 0 ? sd.rowptr[gs + rr] : 0);
        } else
            op.load(sd, doff, poff, w, lane);
        STAMP(li, 1);
        // ... and store the previous part's result underneath the stream
        if (prev_rows >= 0) {
            if constexpr (Op::FRAMES) if (sd.frames) {     // back to global coordinates (uniform branch)
                if (!BWD) rotate_rows<false, NT>(slab, sd.frames + (int64_t)prev_gs * 9, prev_rows, Fs, 3 * Fs, Fs, posL);   // ORD: [x | bank]
                else if (!POS) rotate_rows<false, NT>(xa, sd.frames + (int64_t)prev_gs * 9, prev_rows, 1, CP, 1);
                __syncthreads();
            }
            if (ORD) chain_epilogue_orders<NT>(slab, smem, 2 * nv, RES, prev_row0, prev_rows, C, F, sd.orders == 2, posL);
            else if (!(Op::POS && BWD)) chain_epilogue<BWD, NT, Op::MINW != 1>(xa, slab, RES, prev_row0, prev_rows, C, CP, F, posL);
        }
        STAMP(li, 2);

        const int R = sell_padR(rows);
        const int r = (rt & 0x3ff) == 0x3ff ? -1 : (rt & 0x3ff);
        const int tn = (int)((unsigned)rt >> 10);
        const bool own = r >= 0;
        const int xr = POS ? (int)threadIdx.x : r;                       // LDS row this thread writes: its position / its row
        const int rB = own ? R + (POS ? sell_swzB(xr) : sell_copyB(r, R)) : 0;
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
        const int srow = (POS && !BWD) ? xr : r;                          // this thread's slab row
        int p = POS ? chain_prologue_pos<BWD, NT>(xa, xb, slab, BWD ? slab : inb, rows, own, r, xr, rB, C, CP, F, T, tab, Fs, so, posL)
                    : chain_prologue<BWD, NT>(xa, xb, slab, BWD ? slab : inb, rows, R, C, CP, F, T, tab, Fs, so);
        if (POS && BWD && T == 0 && own) {        // no step: the Horner start is the answer
            XVec<CP> y0; y0.load(xa + xr * CP);
            if constexpr (Op::FRAMES) if (sd.frames) frame_to_global(y0.v, sd.frames + ((int64_t)gs + r) * 9);
            for (int c = 0; c < C; ++c) RES[((int64_t)row0 + r) * C + c] = y0.v[c];
        }
        STAMP(li, 4);

        int rtN = 0, te0N = 0, swN = 0;
        if (has_next) {                           // metadata of the next part: lands during the first step
            const int *lm = sd.lane_meta + ((int64_t)(partN * sd.nsl + wave) * 64 + lane) * 3;
            rtN = lm[0]; te0N = lm[1]; swN = lm[2];
        }
        bool staged = !(has_next && (!BWD || early));   // "nothing to stage early"
#ifndef ESCAT_DEFER_EVENTS
#define ESCAT_DEFER_EVENTS 1
#endif
        constexpr bool DEFER = ESCAT_DEFER_EVENTS != 0;
        int pq0 = 0, pq1 = 0;                     // forward: events of the previous step, applied under this step's first gathers
        for (int t = 1; t <= T; ++t) {
            int q0, q1;
            step_range<BWD>(tab, BWD ? T - t : t, p, q0, q1);
            float acc[C];
#pragma unroll
            for (int c = 0; c < C; ++c) acc[c] = 0.f;
#ifdef ESCAT_STAMPS
            if (t == 5) { STAMPW(li, 8, 0); STAMPW(li, 12, NT - 64); }
#endif
            // The slab bookkeeping of step t-1 (this thread's own slab rows, its own y re-read from its row of xa) needs
            // nothing from this step: it runs while the first gathers are in flight instead of ahead of the barrier.
            op.gather(xa, w, acc, [&]() {
                if (!BWD && own && pq1 > pq0) {
                    XVec<CP> yp; yp.load(xa + xr * CP);
#pragma unroll
                    for (int c = 0; c < C; ++c) apply_events<false>(tab, pq0, pq1, c, yp.v[c], slab + (srow * C + c) * Fs + so);
                }
            });
#ifdef ESCAT_STAMPS
            if (t == 5) { STAMPW(li, 9, 0); STAMPW(li, 13, NT - 64); }
#endif
            if (own) {
                if (tn) Op::tail(val, col, te0, tn, gs, xa, acc, POS ? sd.pos_of_row : nullptr);
                XVec<CP> y;
#pragma unroll
                for (int c = 0; c < CP; ++c) y.v[c] = 0.f;
                if (BWD) {
#pragma unroll
                    for (int c = 0; c < C; ++c) y.v[c] = apply_events<true>(tab, q0, q1, c, acc[c], slab + (srow * C + c) * Fs + so);
                } else {
#pragma unroll
                    for (int c = 0; c < C; ++c) y.v[c] = acc[c];
                }
                y.store(xb + xr * CP); y.store(xb + rB * CP);      // whole padded rows: one wide store per copy
                if (POS && BWD && t == T) {       // layout 1: the owner holds the answer of its row -- straight to HBM
                    if constexpr (Op::FRAMES) if (sd.frames) frame_to_global(y.v, sd.frames + ((int64_t)gs + r) * 9);   // (out of its local frame)
#pragma unroll
                    for (int c = 0; c < C; ++c) RES[((int64_t)row0 + r) * C + c] = y.v[c];
                }
                if (!BWD && (t == T || !DEFER)) { // the last step has no successor to hide behind
#pragma unroll
                    for (int c = 0; c < C; ++c) apply_events<false>(tab, q0, q1, c, acc[c], slab + (srow * C + c) * Fs + so);
                }
            }
            pq0 = q0; pq1 = (!BWD && DEFER && t < T) ? q1 : q0;
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
        if (!BWD) rotate_rows<false, NT>(slab, sd.frames + (int64_t)prev_gs * 9, prev_rows, Fs, 3 * Fs, Fs, posL);   // ORD: [x | bank]
        else if (!POS) rotate_rows<false, NT>(xa, sd.frames + (int64_t)prev_gs * 9, prev_rows, 1, CP, 1);
        __syncthreads();
    }
    if (ORD) chain_epilogue_orders<NT>(slab, smem, 2 * nv, RES, prev_row0, prev_rows, C, F, sd.orders == 2, posL);
    else if (!(Op::POS && BWD)) chain_epilogue<BWD, NT, Op::MINW != 1>(xa, slab, RES, prev_row0, prev_rows, C, CP, F, posL);
}

template <class Op, int NT, bool BWD, bool ORD>
static int launch_sell_k2(const int *ci, const float *v, int n_parts, int max_rows, const SellDev &sd, const float *IN,
                          float *RES, int T, const EvTable &tab, int F, hipStream_t st)
{
    int early = 0;
    // workgroups per CU the registers allow: the 256-VGPR BSR-3 kernel holds one of 512 threads or two of 256 (without
    // the bound above the compiler takes up to 512 registers for a 256-thread kernel: DMAX >= 20 then ran ONE 4-wave
    // workgroup per CU, 0.69 instead of 0.41 ms for 8192 graphs x 250 nodes)
    const int want = Op::MINW == 1 ? (NT == 512 ? 1 : 2) : (NT == 512 ? 2 : 4);
    const size_t lds = escat_sell_lds_bytes(Op::C, Op::CP, max_rows, F, BWD, want, &early, sd.orders);
    if (lds > LDS_BUDGET) return fail(ESCAT_ERANGE, "register-resident chain needs %zu B of LDS", lds);
    if (ORD) {
        // the orders epilogue expands whole output rows in a tile made of the two (dead) vector buffers: one row must fit,
        // or its loop over row chunks never advances
        const int wtot = 1 + F + (sd.orders == 2 ? F * (F - 1) / 2 : 0);
        const int tile = 2 * 2 * sell_padR(max_rows) * Op::CP;
        if (wtot > tile)
            return fail(ESCAT_ERANGE, "escat_sell.orders: an output row of %d floats does not fit the %d-float LDS tile of parts "
                                      "with <= %d rows (use the bank + escat_scatter_orders_f32)", wtot, tile, max_rows);
    }
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
    // experiment knobs (tools/sweep_stagger.sh; DESIGN.md 3 "Round 3"), read once: ESCAT_DBG_PER_CU caps the resident
    // workgroups per CU, ESCAT_DBG_STAGGER delays half of the workgroups at start (10-ns ticks; < 0: the odd ones)
    static const int dbg_per_cu = getenv("ESCAT_DBG_PER_CU") ? atoi(getenv("ESCAT_DBG_PER_CU")) : 0;
    static const int dbg_stagger = getenv("ESCAT_DBG_STAGGER") ? atoi(getenv("ESCAT_DBG_STAGGER")) : 0;
    int per_cu = cached_per_cu;
    const int stagger = dbg_stagger;
    if (dbg_per_cu > 0 && dbg_per_cu < per_cu) per_cu = dbg_per_cu;
    const int64_t resident = (int64_t)per_cu * escat_cu_count();
    const int grid = (int)(n_parts < resident ? n_parts : resident);           // persistent: one wave of workgroups
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NT), lds, st, ci, v, sd, IN, RES, T, tab, F, n_parts, max_rows, early, stagger);
    return ESCAT_OK;
}

template <class Op, int NT, bool BWD>
static int launch_sell_k(const int *ci, const float *v, int n_parts, int max_rows, const SellDev &sd, const float *IN,
                         float *RES, int T, const EvTable &tab, int F, hipStream_t st)
{
    if (sd.nsl * 64 != NT) return fail(ESCAT_EINVAL, "escat_sell.slices_per_part=%d does not match parts of <= %d rows", sd.nsl, max_rows);
    if (sd.layout != (Op::POS ? 1 : 0))
        return fail(ESCAT_EINVAL, "escat_sell.layout=%d: this operator family takes layout %d packs", sd.layout, Op::POS ? 1 : 0);
    if (Op::POS && !sd.pos_of_row) return fail(ESCAT_EINVAL, "escat_sell.pos_of_row is NULL (layout 1)");
    if (!BWD && sd.orders) return launch_sell_k2<Op, NT, false, true>(ci, v, n_parts, max_rows, sd, IN, RES, T, tab, F, st);
    return launch_sell_k2<Op, NT, BWD, false>(ci, v, n_parts, max_rows, sd, IN, RES, T, tab, F, st);
}

#endif
