// escat_sell_bsr3h.hip -- the register-resident chain for a BSR-3 operator whose values are exactly representable in IEEE
// half precision (escat_sell.half_val / .half_aux).
//
// Why: the reference's ellipsoid pipeline stores every operator value as fp16 (process_pyg_data.py:676 default
// hdf5_tensor_dtype='float16', not overridden at generate_ellipsoid_dataset.py:1167-1183; config/dataset_config.py:100)
// and widens to fp32 only at load (models/escgnn_data_classes.py:137-142).  For such an operator the halves ARE the
// values: 9 halves + the slot's 16-bit LDS offset = 5 VGPRs per block instead of 9.5, so a row of 20 blocks takes 100
// registers instead of 190 and TWO 512-thread workgroups share a CU -- one streams its next operator / writes its
// result while the other steps -- where k_chain_sell<OpBsr3> (the whole 512 KB register file for one workgroup) has
// to alternate.  From HBM: 20 B per block instead of 38, 2 load instructions per slot instead of 3.5.
// Arithmetic stays fp32: v_fma_mix_f32 widens the half operand exactly inside the FMA, and the FMAs run in the order
// of OpBsr3::gather_w, so the bank is BIT-IDENTICAL to the fp32-register kernel's on the same values (tested).
//
// pack (built by escat_sell_pack_half from the fp32 pack of the same plan):
//   half_val = uint4[(doff_s + d) * 64 + lane] = half pairs (low | high << 16): (b00 b10)(b01 b11)(b02 b12)(b20 b21)
//   half_aux = uint32[(doff_s + d) * 64 + lane] = b22 (half) | (16 * LDS row of the slot) << 16
// The kernel is deliberately lean (128 VGPRs): no cross-part software pipeline inside a workgroup -- the second
// workgroup on the CU is the overlap -- and rows longer than dmax find their tail records again every step (rare:
// <= 5e-5 of the rows by the capacity rule of the pack).
#include "escat_sell_kernel.h"

typedef unsigned vu4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float fma_mix_lo(unsigned h, float x, float a)
{ asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "+v"(a) : "v"(h), "v"(x)); return a; }
__device__ __forceinline__ float fma_mix_hi(unsigned h, float x, float a)
{ asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(a) : "v"(h), "v"(x)); return a; }
// fma(h, x, +0): the first term of an accumulator that starts at zero (no zero register to keep, no v_mov)
__device__ __forceinline__ float mul_mix_lo(unsigned h, float x)
{ float a; asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(a) : "v"(h), "v"(x)); return a; }
__device__ __forceinline__ float mul_mix_hi(unsigned h, float x)
{ float a; asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(a) : "v"(h), "v"(x)); return a; }

#ifndef ESCAT_QH_RING
#define ESCAT_QH_RING 2
#endif

template <int DMAX> struct OpQH {
    static constexpr int RING = ESCAT_QH_RING;    // LDS gathers in flight per lane (4 VGPRs each; 4 waves per SIMD hide the rest)
    static constexpr int NLOADS = 2 * DMAX;       // vector-memory instructions of load(): counted by the caller's s_waitcnt
    vu4 h[DMAX];
    unsigned m[DMAX];
    // one UNCONDITIONAL load pair per slot register (see OpBsr3::load): slots past the width re-read the last real slot
    __device__ __forceinline__ void load(const SellDev &sd, int doff, int w, int lane)
    {
        const vu4 *vb = reinterpret_cast<const vu4 *>(sd.half_val) + (int64_t)doff * 64 + lane;
        const unsigned *mb = sd.half_aux + (int64_t)doff * 64 + lane;
        const int wl = w > 0 ? w - 1 : 0;
#pragma unroll
        for (int d = 0; d < DMAX; ++d) {
            const int dd = d < wl ? d : wl;
            h[d] = vb[dd * 64];
            m[d] = mb[dd * 64];
        }
    }
    // gather_w runs an even number of slots: a slice of odd width w runs slot w -- a re-read of its last real block, see
    // load() -- against the all-zero row (as OpBsr3 does through its padded lc pairs).  Once per part, after the burst.
    __device__ __forceinline__ void point_odd_slot_at_zero_row(int w, int rows)
    {
        if (w & 1) {
            const unsigned z = (unsigned)rows << 20;          // 16 * rows << 16
#pragma unroll
            for (int d = 1; d < DMAX; d += 2) m[d] = d == w ? ((m[d] & 0xffffu) | z) : m[d];
        }
    }
    template <int W>
    __device__ __forceinline__ void gather_w(unsigned base, float (&s)[5]) const
    {
        constexpr int D = RING;
        vf4 xv[D];
        // LDS address of slot d = base + (m[d] >> 16) as ONE VALU instruction that the optimiser cannot take apart: written
        // in C the shift is loop-invariant over the steps and gets hoisted into DMAX more live registers
        auto addr = [&](int d) -> unsigned {
            unsigned a;
            asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD"
                : "=v"(a) : "v"(m[d]), "v"(base));
            return a;
        };
#pragma unroll
        for (int d = 0; d < D && d < W; ++d) lds_gather(xv[d], addr(d));
#pragma unroll
        for (int d = 0; d < W; ++d) {
            if (W - 1 - d >= D - 1) lds_wait<D - 1>(xv[d % D]);
            else if (W - 1 - d == 2) lds_wait<2>(xv[d % D]);
            else if (W - 1 - d == 1) lds_wait<1>(xv[d % D]);
            else lds_wait<0>(xv[d % D]);
            const vf4 x = xv[d % D];
            // OpBsr3's order: (y0, y1) += (b0j, b1j) xj for j = 0, 1, 2; (t0, t1) += (b20 x0, b21 x1); u += b22 x2
            if (d == 0) {
                s[0] = mul_mix_lo(h[d].x, x.x); s[1] = mul_mix_hi(h[d].x, x.x);
                s[2] = mul_mix_lo(h[d].w, x.x); s[3] = mul_mix_hi(h[d].w, x.y);
                s[4] = mul_mix_lo(m[d], x.z);
            } else {
                s[0] = fma_mix_lo(h[d].x, x.x, s[0]); s[1] = fma_mix_hi(h[d].x, x.x, s[1]);
                s[2] = fma_mix_lo(h[d].w, x.x, s[2]); s[3] = fma_mix_hi(h[d].w, x.y, s[3]);
                s[4] = fma_mix_lo(m[d], x.z, s[4]);
            }
            s[0] = fma_mix_lo(h[d].y, x.y, s[0]); s[1] = fma_mix_hi(h[d].y, x.y, s[1]);
            s[0] = fma_mix_lo(h[d].z, x.z, s[0]); s[1] = fma_mix_hi(h[d].z, x.z, s[1]);
            if (d + D < W) {
                asm volatile("" :: "v"(s[0]), "v"(s[1]), "v"(s[2]), "v"(s[3]), "v"(s[4]));     // see OpBsr3::gather_w
                lds_gather(xv[d % D], addr(d + D));
            }
        }
    }
    template <int W>
    __device__ __forceinline__ void dispatch(int w2, unsigned base, float (&s)[5]) const
    {
        if constexpr (W <= DMAX) {
            if (w2 == W / 2) gather_w<W>(base, s);
            else dispatch<W + 2>(w2, base, s);
        }
    }
    __device__ __forceinline__ void gather(const float *xa, int w, float (&a)[3]) const
    {
        float s[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
        dispatch<2>((w + 1) >> 1, lds_addr(xa), s);
        a[0] += s[0]; a[1] += s[1]; a[2] += s[4] + (s[2] + s[3]);
    }
};

template <int DMAX, int NT, bool BWD, bool ORD>
__global__ void __launch_bounds__(NT, 4)          // 4 waves per SIMD = 128 VGPRs: two 512-thread (four 256-thread) workgroups per CU
k_chain_qh(const int *__restrict__ col, const float *__restrict__ blk, SellDev sd, const float *__restrict__ IN,
           float *__restrict__ RES, int T, EvTable tab, int F, int n_parts, int max_rows)
{
    constexpr int C = 3, CP = 4;
    extern __shared__ __align__(16) float smem[];
    const int nv = 2 * sell_padR(max_rows) * CP;
    const int so = ORD ? 1 : 0, Fs = F + so;
    const int slab_sz = (max_rows * C * Fs + 3) & ~3;
    float *xa = smem, *xb = smem + nv, *slab = smem + 2 * nv;
    float *inb = slab + slab_sz;                                  // forward: the staged x, then the row -> position table
    int *posL = reinterpret_cast<int *>(inb + ((max_rows * C + 3) & ~3));
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int in_w = BWD ? C * F : C;
    const int xr = threadIdx.x;
    // lane records of the NEXT part, fetched by LDS-DMA while this part runs (no registers held across the steps; every
    // wave fetches and reads only its own slice's records, so no barrier is involved): metaR[tid] = row | tail_n << 10,
    // metaS[wave * 8 + 0..4] = doff, poff, w, first row, rows
    int *metaR = reinterpret_cast<int *>(BWD ? slab + slab_sz : reinterpret_cast<float *>(posL + ((max_rows + 3) & ~3)));
    int *metaS = metaR + NT;
    auto part_of = [&](int li_) { return sd.part_list ? sd.part_list[li_] : li_; };
    auto dma_meta = [&](int part_) {
        const int *lm = sd.lane_meta + ((int64_t)(part_ * sd.nsl + wave) * 64 + lane) * 3;
        __builtin_amdgcn_global_load_lds((glb_void_t *)lm, (lds_void_t *)(metaR + wave * 64), 4, 0, 0);
        if (lane < 8) __builtin_amdgcn_global_load_lds((glb_void_t *)(lm + 2), (lds_void_t *)(metaS + wave * 8), 4, 0, 0);
    };
    int prev_row0 = 0, prev_rows = -1;
    OpQH<DMAX> op;
    int li = blockIdx.x;
    if (li >= n_parts) return;
    int part = part_of(li);
    dma_meta(part);
    bool x_staged = false;                        // forward: inb already holds this part's x (copied in during the previous part)
    for (;;) {
        STAMP(li, 0);
        // this part's records -- and, forward, its x -- were requested a part ago (first part: just now)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const int rt = metaR[threadIdx.x], sw = metaS[wave * 8 + (lane & 7)];
        const int doff = __builtin_amdgcn_readlane(sw, 0), w = __builtin_amdgcn_readlane(sw, 2);
        const int gs = __builtin_amdgcn_readlane(sw, 3), rows = __builtin_amdgcn_readlane(sw, 4);
        const int row0 = sd.part_row0 ? sd.part_row0[li] : gs;
        const int liN = li + gridDim.x;
        const bool has_next = liN < n_parts;      // uniform
        const int partN = has_next ? part_of(liN) : 0;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // the records are in registers: their LDS words are free
        STAMP(li, 1);
        if (has_next) dma_meta(partN);
        // backward (the slab of the previous part is dead: its last step ended with a barrier) and the first forward part:
        // the input now, ahead of the operator burst, so that the counted wait below leaves the burst in flight
        const bool stage_now = BWD || !x_staged;
        if (stage_now) dma_in<NT>(BWD ? slab : inb, IN + (int64_t)row0 * in_w, rows * in_w);
        asm volatile("" ::: "memory");
        op.load(sd, doff, w, lane);
        STAMP(li, 8);
        if (stage_now) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(OpQH<DMAX>::NLOADS) : "memory");
        STAMP(li, 9);
        // the previous result leaves underneath the burst
        if (!BWD && prev_rows >= 0) {
            if (ORD) chain_epilogue_orders<NT>(slab, smem, 2 * nv, RES, prev_row0, prev_rows, C, F, sd.orders == 2, posL);
            else chain_epilogue<false, NT, true>(xa, slab, RES, prev_row0, prev_rows, C, CP, F, posL);
        }
        STAMP(li, 2);
        const int r = (rt & 0x3ff) == 0x3ff ? -1 : (rt & 0x3ff);
        const bool own = r >= 0;
        const bool has_tail = __any((int)((unsigned)rt >> 10));  // wave-uniform, rare
        const int R = sell_padR(rows);
        const int rB = R + sell_swzB(xr);
        __syncthreads();                          // every wave's input has landed; the epilogue's LDS reads are done
        STAMP(li, 3);
        int p = chain_prologue_pos<BWD, NT>(xa, xb, slab, BWD ? slab : inb, rows, own, r, xr, rB, C, CP, F, T, tab, Fs, so, posL);
        if (BWD && T == 0 && own)
            for (int c = 0; c < C; ++c) RES[((int64_t)row0 + r) * C + c] = xa[xr * CP + c];
        op.point_odd_slot_at_zero_row(w, rows);
        x_staged = false;
        if (!BWD && has_next) {                   // the next part's x, now that its records are here and inb is free
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const int swN = metaS[wave * 8 + (lane & 7)];
            const int gsN = __builtin_amdgcn_readlane(swN, 3), rowsN = __builtin_amdgcn_readlane(swN, 4);
            dma_in<NT>(inb, IN + (int64_t)(sd.part_row0 ? sd.part_row0[liN] : gsN) * in_w, rowsN * in_w);
            x_staged = true;
        }
#ifdef ESCAT_STAMPS
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); STAMP(li, 4);
#endif
        for (int t = 1; t <= T; ++t) {
            int q0, q1;
            step_range<BWD>(tab, BWD ? T - t : t, p, q0, q1);
            float acc[C] = {0.f, 0.f, 0.f};
            op.gather(xa, w, acc);
            if (own) {
                if (has_tail) {                   // the lane's tail record again (L2): no registers held across the steps
                    const int *lm2 = sd.lane_meta + ((int64_t)(part * sd.nsl + wave) * 64 + lane) * 3;
                    const int tn = (int)((unsigned)lm2[0] >> 10);
                    if (tn) OpBsr3<DMAX>::tail(blk, col, lm2[1], tn, gs, xa, acc, sd.pos_of_row);
                }
                const int srow = BWD ? r : xr;
                XVec<CP> y;
                y.v[3] = 0.f;
                if (BWD) {
#pragma unroll
                    for (int c = 0; c < C; ++c) y.v[c] = apply_events<true>(tab, q0, q1, c, acc[c], slab + (srow * C + c) * Fs + so);
                } else {
#pragma unroll
                    for (int c = 0; c < C; ++c) y.v[c] = acc[c];
                }
                y.store(xb + xr * CP); y.store(xb + rB * CP);
                if (BWD && t == T) {
#pragma unroll
                    for (int c = 0; c < C; ++c) RES[((int64_t)row0 + r) * C + c] = y.v[c];
                }
                if (!BWD) {
#pragma unroll
                    for (int c = 0; c < C; ++c) apply_events<false>(tab, q0, q1, c, acc[c], slab + (srow * C + c) * Fs + so);
                }
            }
            __syncthreads();
            float *tmp = xa; xa = xb; xb = tmp;
#ifdef ESCAT_STAMPS
            if (t == 1) STAMP(li, 5);
#endif
        }
        STAMP(li, 6); STAMP(li, 7);
        prev_row0 = row0; prev_rows = rows;
        if (!has_next) break;
        li = liN; part = partN;
    }
    if (!BWD && prev_rows >= 0) {
        if (ORD) chain_epilogue_orders<NT>(slab, smem, 2 * nv, RES, prev_row0, prev_rows, C, F, sd.orders == 2, posL);
        else chain_epilogue<false, NT, true>(xa, slab, RES, prev_row0, prev_rows, C, CP, F, posL);
    }
}

template <int DMAX, int NT, bool BWD, bool ORD>
static int launch_qh2(const int *ci, const float *v, int n_parts, int max_rows, const SellDev &sd, const float *IN,
                      float *RES, int T, const EvTable &tab, int F, hipStream_t st)
{
    int early = 0;
    // single-slab backward (`want` large enough that two slabs never "fit"): the second workgroup on the CU is the overlap
    const size_t lds = escat_sell_lds_bytes(3, 4, max_rows, F, BWD, 1 << 20, &early, sd.orders) + (NT + NT / 8) * sizeof(int);
    if (lds > LDS_BUDGET) return fail(ESCAT_ERANGE, "register-resident chain needs %zu B of LDS", lds);
    if (ORD) {
        const int wtot = 1 + F + (sd.orders == 2 ? F * (F - 1) / 2 : 0);
        const int tile = 2 * 2 * sell_padR(max_rows) * 4;
        if (wtot > tile)
            return fail(ESCAT_ERANGE, "escat_sell.orders: an output row of %d floats does not fit the %d-float LDS tile of parts "
                                      "with <= %d rows (use the bank + escat_scatter_orders_f32)", wtot, tile, max_rows);
    }
    auto kern = k_chain_qh<DMAX, NT, BWD, ORD>;
    int rc = set_lds(kern, lds); if (rc) return rc;
    static thread_local size_t cached_lds = (size_t)-1; static thread_local int cached_per_cu = 0, cached_dev = -1;
    int dev_now = 0;
    (void)hipGetDevice(&dev_now);
    if (cached_lds != lds || cached_dev != dev_now) {
        int per = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per, kern, NT, lds) != hipSuccess || per < 1) per = 1;
        cached_per_cu = per; cached_lds = lds; cached_dev = dev_now;
    }
    const int64_t resident = (int64_t)cached_per_cu * escat_cu_count();
    const int grid = (int)(n_parts < resident ? n_parts : resident);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NT), lds, st, ci, v, sd, IN, RES, T, tab, F, n_parts, max_rows);
    return ESCAT_OK;
}

template <int DMAX, bool BWD>
static int launch_qh_d(int max_rows, const int *ci, const float *v, int n_parts, const SellDev &sd, const float *IN,
                       float *RES, int T, const EvTable &tab, int F, hipStream_t st)
{
    const int NT = max_rows <= 256 ? 256 : 512;
    if (sd.nsl * 64 != NT) return fail(ESCAT_EINVAL, "escat_sell.slices_per_part=%d does not match parts of <= %d rows", sd.nsl, max_rows);
    if (sd.layout != 1 || !sd.pos_of_row) return fail(ESCAT_EINVAL, "half-valued BSR-3 packs are layout 1 (with pos_of_row)");
    if (((uintptr_t)sd.half_val & 15) != 0) return fail(ESCAT_EINVAL, "escat_sell.half_val must be 16-byte aligned");
#define GO(nt, ord) return launch_qh2<DMAX, nt, BWD, ord>(ci, v, n_parts, max_rows, sd, IN, RES, T, tab, F, st)
    if constexpr (!BWD) {
        if (sd.orders) { if (NT == 256) GO(256, true); else GO(512, true); }
    }
    if (NT == 256) GO(256, false); else GO(512, false);
#undef GO
}

int escat_launch_bsr3h_sell(bool bwd, ESCAT_SELL_ARGS)
{
    switch (sd.dmax) {
#define CASE(d) case d: return bwd ? launch_qh_d<d, true>(max_rows, ci, v, n_parts, sd, IN, RES, T, tab, F, st) \
                                   : launch_qh_d<d, false>(max_rows, ci, v, n_parts, sd, IN, RES, T, tab, F, st);
        CASE(8) CASE(12) CASE(16) CASE(20)
#undef CASE
    }
    return fail(ESCAT_EINVAL, "escat_sell.dmax=%d: half-valued BSR-3 packs use 8, 12, 16 or 20", sd.dmax);
}
#ifdef ESCAT_STAMPS
ESCAT_STAMP_SETTER(escat_debug_set_stamps_bsr3h)
#endif
