// escat_sell_both.hip -- BOTH tracks of ESCGNNModular.forward (models/escgnn_modular.py:674 scalar, :685 vector: two
// _scatter calls, each a get_Batch_Wjxs chain, pyg_utilities.py:350-385) in ONE register-resident launch, for a batch
// whose vector operator has the factored form Q = D Pt D^T on P's own pattern with |Pt_ij| = P_ij (what process_pyg_data
// builds, process_pyg_data.py:306-313, :492, :823-857).
//
// Why (round 3, measured): a step of the P kernel and of the factored-Q kernel is ~45 % fixed latency (the chain
// "last gather -> FMA -> vector stores -> their completion -> barrier -> first gather", ~1200 cycles per workgroup and
// step at cfg2, tools/exp_imbalance.py: 2.0 us fixed + 0.2 us per slot) and both kernels run the SAME gathers on the same
// plan.  Here one thread serves its row for both tracks: one address per slot feeds two ds_read_b128 (the second through
// the instruction's immediate offset), one barrier and one latency chain per step carry twice the work, the operator is
// streamed once (|p| + the slot's LDS row with three sign bits = 6 B per nonzero for both tracks).
// What made the fusion lose on paper before ("both tracks in one launch -- priced, not built") was LDS: two
// tracks' vectors AND a six-channel output slab need 152 KB = one workgroup per CU.  This kernel has NO slab: the owner
// of a row keeps the one open band of each track in registers (6 floats); a finished band leaves through a small
// row-ordered stage in LDS (rows * 3 floats per track: the staged x of the part until the prologue has read it), written
// out by all threads in row order after the step's barrier (forward); the upstream gradient of a step's events is fetched
// in row order one step ahead, under that step's gathers, and parked in the same stage for the owners (backward).
// LDS is the four vectors (two tracks x ping-pong, two copies each) + the stage: 80 KB, two workgroups per CU as before.
//
// Registers per slot: v = +-|p| (sign = the vector track's channel 0; the scalar track multiplies by |v|, a free source
// modifier) and u = 16 * LDS row | relative signs of channels 1, 2 as the top bits of bytes 2 and 3, which one SDWA
// byte-xor each moves into the gathered x before the FMA.  The scalar track's FMAs run in OpCsr's order and the vector
// track's in OpCsrV3's, the rotations use rotate_rows' / frame_to_global's formulas: both banks are BIT-IDENTICAL to what
// the two separate launches give on the same values (tested).
//
// Event tables: the shared-scales plans (every slot = one power, or the difference of two powers; at most one band open
// at a time) -- all of the drop-in's dyadic / custom-scale banks.  escat.hip turns the table into a BothProg and refuses
// anything else (per-channel scales, |.| banks): the caller then runs the two chains separately.
#include "escat_sell_kernel.h"

struct __attribute__((packed, aligned(4))) f2u_k { float x, y; };      // an 8-byte access that is only dword-aligned
template <int OFF> __device__ __forceinline__ void lds_gather_at(vf4 &d, unsigned a)
{ asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(a), "n"(OFF)); }
template <int OFF> __device__ __forceinline__ void lds_store_at(unsigned a, const vf4 &v)
{ asm volatile("ds_write_b128 %0, %1 offset:%2" :: "v"(a), "v"(v), "n"(OFF) : "memory"); }
// x <- x with its sign flipped iff the top bit of byte B of u is set (bytes 2 / 3 of OpBoth::u)
template <int B> __device__ __forceinline__ float flip_by(float x, unsigned u)
{
    if (B == 2) asm("v_xor_b32_sdwa %0, %0, %1 dst_sel:BYTE_3 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_3 src1_sel:BYTE_2" : "+v"(x) : "v"(u));
    else        asm("v_xor_b32_sdwa %0, %0, %1 dst_sel:BYTE_3 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_3 src1_sel:BYTE_3" : "+v"(x) : "v"(u));
    return x;
}

// a += |v| * x resp. a += v * x as single instructions: left to the compiler these become v_pk_fma_f32 on materialised
// (|v|, |v|) and (v, v) register pairs -- 80 more live registers, all spilled
__device__ __forceinline__ void fma_abs(float &a, float v, float x) { asm("v_fma_f32 %0, |%1|, %2, %0" : "+v"(a) : "v"(v), "v"(x)); }
__device__ __forceinline__ void fma_one(float &a, float v, float x) { asm("v_fmac_f32_e32 %0, %1, %2" : "+v"(a) : "v"(v), "v"(x)); }

#ifndef ESCAT_BOTH_RING      // slots in flight per lane.  2: as fast as 3 at 20 slots (0.344 vs 0.346 ms forward on a settled device), and
#define ESCAT_BOTH_RING 2    // the eight registers it frees end the spills of the 24-slot instances (k = 14: 0.395 + 0.402 vs 0.404 + 0.422)
#endif
#ifndef ESCAT_BOTH_EXP     // timing experiments only (results wrong; `make both_exp EXP=<bits>`): 1 no sign flips, 2 no vector-track
#define ESCAT_BOTH_EXP 0   // gathers / FMAs, 4 no global band stores, 8 no scalar-track FMAs, 16 no band write-out at all
#endif

template <int DMAX, int VSTRIDE> struct OpBoth {
    static constexpr int RING = ESCAT_BOTH_RING;  // slots in flight per lane (two gathers = 8 VGPRs each)
    float v[DMAX];
    unsigned u[DMAX];
    // see OpBsr3::load for the unconditional burst; sd.val = the magnitudes, sd.lc = P's LDS rows with the sign bits
    __device__ __forceinline__ void load(const SellDev &sd, int doff, int poff, int w, int lane)
    {
        const int wl = w > 0 ? w - 1 : 0;
        const int pl = w > 0 ? ((w + SLOT_GROUP - 1) / SLOT_GROUP) * (SLOT_GROUP / 2) - 1 : 0;
        unsigned lc[DMAX / 2];
#pragma unroll
        for (int j2 = 0; j2 < DMAX / 2; ++j2)
            lc[j2] = sd.lc[(int64_t)(poff + (j2 < pl ? j2 : pl)) * 64 + lane];
#pragma unroll
        for (int d = 0; d < DMAX; ++d) v[d] = sd.val[(int64_t)(doff + (d < wl ? d : wl)) * 64 + lane];
#pragma unroll
        for (int d = 0; d < DMAX; ++d) {
            const unsigned f = (d & 1) ? (lc[d / 2] >> 16) : (lc[d / 2] & 0xffffu);     // row | s0 << 11 | s1 << 12 | s2 << 13
            v[d] = __uint_as_float(__float_as_uint(v[d]) | ((f << 20) & 0x80000000u));
            const unsigned rel = f ^ (f >> 1), rel2 = f ^ (f >> 2);                    // bit 11: s0 ^ s1 resp. s0 ^ s2
            u[d] = ((f & 0x7ffu) << 4) | ((rel << 12) & 0x00800000u) | ((rel2 << 20) & 0x80000000u);
        }
    }
    __device__ __forceinline__ unsigned addr(int d, unsigned base) const
    {
        unsigned a;      // one VALU instruction the optimiser cannot take apart (see OpQH::gather_w)
        asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD" : "=v"(a) : "v"(u[d]), "v"(base));
        return a;
    }
    template <int W>
    __device__ __forceinline__ void gather_w(unsigned base, float (&aP)[3], float (&aQ)[3]) const
    {
        constexpr int D = RING;
        vf4 xp[D], xq[D];
#pragma unroll
        for (int d = 0; d < D && d < W; ++d) {
            const unsigned a = addr(d, base);
            lds_gather_at<0>(xp[d], a); if (!(ESCAT_BOTH_EXP & 2)) lds_gather_at<VSTRIDE>(xq[d], a);
        }
#pragma unroll
        for (int d = 0; d < W; ++d) {
            // both gathers of slot d have landed once at most 2 * min(D-1, W-1-d) younger LDS operations are outstanding
            const int later = ((ESCAT_BOTH_EXP & 2) ? 1 : 2) * (W - 1 - d < D - 1 ? W - 1 - d : D - 1);
            vf4 &last = (ESCAT_BOTH_EXP & 2) ? xp[d % D] : xq[d % D];
            if (later >= 8) lds_wait<8>(last);
            else if (later == 6) lds_wait<6>(last);
            else if (later == 4) lds_wait<4>(last);
            else if (later == 3) lds_wait<3>(last);
            else if (later == 2) lds_wait<2>(last);
            else if (later == 1) lds_wait<1>(last);
            else lds_wait<0>(last);
            asm volatile("" : "+v"(xp[d % D]));               // the first gather landed before the second
            const vf4 p = xp[d % D];
            if (!(ESCAT_BOTH_EXP & 8)) { fma_abs(aP[0], v[d], p.x); fma_abs(aP[1], v[d], p.y); fma_abs(aP[2], v[d], p.z); }
            else asm volatile("" :: "v"(p.x));
            if (!(ESCAT_BOTH_EXP & 2)) {
                const vf4 q = xq[d % D];
                const float q1 = (ESCAT_BOTH_EXP & 1) ? q.y : flip_by<2>(q.y, u[d]), q2 = (ESCAT_BOTH_EXP & 1) ? q.z : flip_by<3>(q.z, u[d]);
                fma_one(aQ[0], v[d], q.x); fma_one(aQ[1], v[d], q1); fma_one(aQ[2], v[d], q2);
                asm volatile("" :: "v"(q.w));
            }
            asm volatile("" :: "v"(p.w));                     // keep the gathers ds_read_b128
            if (d + D < W) {
                asm volatile("" :: "v"(aP[0]), "v"(aP[1]), "v"(aP[2]), "v"(aQ[0]), "v"(aQ[1]), "v"(aQ[2]));     // see OpBsr3::gather_w
                const unsigned a = addr(d + D, base);
                lds_gather_at<0>(xp[d % D], a); if (!(ESCAT_BOTH_EXP & 2)) lds_gather_at<VSTRIDE>(xq[d % D], a);
            }
        }
    }
    template <int W>
    __device__ __forceinline__ void dispatch(int w2, unsigned base, float (&aP)[3], float (&aQ)[3]) const
    {
        if constexpr (W <= DMAX) {
            if (w2 == W / 2) gather_w<W>(base, aP, aQ);
            else dispatch<W + 2>(w2, base, aP, aQ);
        }
    }
    __device__ __forceinline__ void gather(unsigned base, int w, float (&aP)[3], float (&aQ)[3]) const { dispatch<2>((w + 1) >> 1, base, aP, aQ); }
};

// entries beyond dmax (rare): from the global arrays in the packed form of k_csr_step<.., 4> (col | signs << 29, |value|)
template <int VSTRIDE>
__device__ __forceinline__ void both_tail(const float *__restrict__ mag, const int *__restrict__ colw, int e0, int n, const float *xa,
                                          float (&aP)[3], float (&aQ)[3], const int *__restrict__ pos)
{
    for (int e = e0; e < e0 + n; ++e) {
        const unsigned cw = (unsigned)colw[e], vb = __float_as_uint(mag[e]);
        const float *xp = xa + pos[(int)(cw & 0x1fffffffu)] * 4, *xq = xp + VSTRIDE / 4;
        const float m = __uint_as_float(vb);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            aP[c] = fmaf(m, xp[c], aP[c]);
            aQ[c] = fmaf(__uint_as_float(vb | ((cw << (2 - c)) & 0x80000000u)), xq[c], aQ[c]);
        }
    }
}

// LDS (bytes from the dynamic base): xaP | xaQ | xbP | xbQ, VSTRIDE each (two copies of <= NT + 16 rows of 16 B);
// forward: + the stage (rows * 3 floats per track: the part's x until the prologue, then one finished band at a time on
// its way out); + the slice words of the next part (8 per wave)
template <int DMAX, int NT, bool BWD>
__global__ void __launch_bounds__(NT, 4)          // 128 VGPRs: two 512-thread (four 256-thread) workgroups per CU
k_chain_both(const int *__restrict__ colw, const float *__restrict__ mag, SellDev sd, const float *__restrict__ INs,
             const float *__restrict__ INv, float *__restrict__ RESs, float *__restrict__ RESv, int T, BothProg prog, int F,
             int n_parts, int max_rows)
{
    constexpr int VSTRIDE = 2 * (NT + 16) * 16;
    extern __shared__ __align__(16) float smem[];
    const unsigned lds0 = lds_addr(smem);
    const int stage = (max_rows * 3 + 3) & ~3;
    float *inbS = smem + 4 * (VSTRIDE / 4), *inbV = inbS + stage;            // the stage: rows * 3 floats per track
    int *metaS = reinterpret_cast<int *>(smem + 4 * (VSTRIDE / 4) + 2 * stage);
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int xr = threadIdx.x;
    auto part_of = [&](int li_) { return sd.part_list ? sd.part_list[li_] : li_; };
    auto dma_meta = [&](int part_) {                  // the slice words of a part's slices: every wave fetches and reads its own
        if (lane < 8) {
            const int *lm = sd.lane_meta + ((int64_t)(part_ * sd.nsl + wave) * 64 + lane) * 3 + 2;
            __builtin_amdgcn_global_load_lds((glb_void_t *)lm, (lds_void_t *)(metaS + wave * 8), 4, 0, 0);
        }
    };
    OpBoth<DMAX, VSTRIDE> op;
    int li = blockIdx.x;
    if (li >= n_parts) return;
    int part = part_of(li);
    dma_meta(part);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    int sw = metaS[wave * 8 + (lane & 7)];
    // a part's lane record is requested while the part before it runs (first part: here): the frames' address depends on it
    int rt, te0;
    {
        const int *lm0 = sd.lane_meta + ((int64_t)(part * sd.nsl + wave) * 64 + lane) * 3;
        rt = lm0[0]; te0 = lm0[1];
    }
    for (;;) {
        const int doff = __builtin_amdgcn_readlane(sw, 0), poff = __builtin_amdgcn_readlane(sw, 1);
        const int w = __builtin_amdgcn_readlane(sw, 2);
        const int gs = __builtin_amdgcn_readlane(sw, 3), rows = __builtin_amdgcn_readlane(sw, 4);
        const int row0 = sd.part_row0 ? sd.part_row0[li] : gs;
        const int liN = li + gridDim.x;
        const bool has_next = liN < n_parts;      // uniform
        const int partN = has_next ? part_of(liN) : 0;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // sw is in registers: its LDS words are free
        if (has_next) dma_meta(partN);
        if (!BWD) {                               // (the stage is free: the previous part's last band left behind a barrier)
            dma_in<NT>(inbS, INs + (int64_t)row0 * 3, rows * 3);
            dma_in<NT>(inbV, INv + (int64_t)row0 * 3, rows * 3);
        }
        asm volatile("" ::: "memory");
        op.load(sd, doff, poff, w, lane);
        const int r = (rt & 0x3ff) == 0x3ff ? -1 : (rt & 0x3ff);
        const bool own = r >= 0;
        const int tn = (int)((unsigned)rt >> 10);
        const bool has_tail = __any(tn);          // wave-uniform, rare
        // the row's local frame.  Backward: kept for the whole part (gradients enter and the answer leaves through it).
        // Forward: only the prologue's input rotation uses it here -- finished bands are rotated by the threads that
        // write them out (flush below), which frees nine registers of the step loop.
        float o[9];
        {
            const float *fp = sd.frames + ((int64_t)gs + (own ? r : 0)) * 9;
#pragma unroll
            for (int k = 0; k < 9; ++k) o[k] = fp[k];
        }
        const int64_t orow = ((int64_t)row0 + (own ? r : 0)) * 3;              // first of the row's three (row, channel) lines in IN / RES
        const unsigned offA = (unsigned)xr * 16u, offB = (unsigned)(sell_padR(rows) + sell_swzB(xr)) * 16u;
        // ---- first vector of both tracks + the events of step 0 (forward) / of step T (backward)
        float hP[3] = {0.f, 0.f, 0.f}, hQ[3] = {0.f, 0.f, 0.f};               // forward: the open band of each track
        int p = 0;                                // cursor into prog.s (forward: next step to serve; backward: one past it)
        float yP[3], yQ[3];
        // Backward input.  The upstream gradient of a step's events is one float per (row, channel) at stride F in G: read by
        // the owners it is 64 scattered lines per wave-load, twelve of them per event step (measured: +10 us per step).  Instead
        // every thread loads elements i, i + NT, ... (consecutive lanes 24 B apart; both slots of a step in one 8-byte load when
        // they are adjacent) one step ahead, under that step's gathers, adds the step's terms and parks the sum in the stage in
        // row order; the owners pick their row's six floats up from there.  (sum of the terms first, then onto the vector:
        // fp32 rounding apart from adding the terms one by one, as the separate kernels do)
        constexpr int EPL = 3;                    // elements per thread and track (rows * 3 <= 3 * NT)
        float LA[EPL][2], LB[EPL][2];             // backward: the (<= 2) terms of the step being fetched, per element and track
        auto gload = [&](int ps) {
            const int nt = prog.s[ps].n, s0 = prog.s[ps].t[0].slot, s1 = nt > 1 ? prog.s[ps].t[1].slot : s0;
            int base = xr;
            asm volatile("" : "+v"(base));
            const int n3 = rows * 3;
            const bool adj = nt > 1 && (s1 == s0 + 1 || s0 == s1 + 1);
            const int lo = s0 < s1 ? s0 : s1;
#pragma unroll
            for (int k = 0; k < EPL; ++k) {
                const int i = base + k * NT, ii = i < n3 ? i : 0;
                const float *gs_ = INs + ((int64_t)row0 * 3 + ii) * F, *gv_ = INv + ((int64_t)row0 * 3 + ii) * F;
                if (adj) {
                    const f2u_k a = *reinterpret_cast<const f2u_k *>(gs_ + lo), b = *reinterpret_cast<const f2u_k *>(gv_ + lo);
                    LA[k][0] = s0 == lo ? a.x : a.y; LB[k][0] = s0 == lo ? a.y : a.x;
                    LA[k][1] = s0 == lo ? b.x : b.y; LB[k][1] = s0 == lo ? b.y : b.x;
                } else {
                    LA[k][0] = gs_[s0]; LA[k][1] = gv_[s0];
                    LB[k][0] = nt > 1 ? gs_[s1] : 0.f; LB[k][1] = nt > 1 ? gv_[s1] : 0.f;
                }
            }
        };
        auto gpark = [&](int ps) {
            const int nt = prog.s[ps].n;
            const bool n0 = prog.s[ps].t[0].kind & 4, n1 = nt > 1 && (prog.s[ps].t[1].kind & 4);
            int base = xr;
            asm volatile("" : "+v"(base));
#pragma unroll
            for (int k = 0; k < EPL; ++k) {
                const int i = base + k * NT;
                float sP = n0 ? -LA[k][0] : LA[k][0], sQ = n0 ? -LA[k][1] : LA[k][1];
                if (nt > 1) { sP += n1 ? -LB[k][0] : LB[k][0]; sQ += n1 ? -LB[k][1] : LB[k][1]; }
                if (i < rows * 3) { inbS[i] = sP; inbV[i] = sQ; }
            }
        };
        auto gtake = [&](float (&aP_)[3], float (&aQ_)[3]) {    // the owner adds its row's parked sums (vector track: into the local frame)
            if (own) {
                const float g0 = inbV[r * 3], g1 = inbV[r * 3 + 1], g2 = inbV[r * 3 + 2];
#pragma unroll
                for (int c = 0; c < 3; ++c) aP_[c] += inbS[r * 3 + c];
                aQ_[0] += fmaf(o[6], g2, fmaf(o[3], g1, o[0] * g0));              // rotate_rows<true>
                aQ_[1] += fmaf(o[7], g2, fmaf(o[4], g1, o[1] * g0));
                aQ_[2] += fmaf(o[8], g2, fmaf(o[5], g1, o[2] * g0));
            }
        };
        if (BWD) {
            p = prog.n;
#pragma unroll
            for (int c = 0; c < 3; ++c) { yP[c] = 0.f; yQ[c] = 0.f; }
            const bool evT = p > 0 && prog.s[p - 1].step == T;
            if (evT) gload(p - 1);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();                      // the previous part's last gathers are done: its vectors and the stage are free
            if (evT) {
                gpark(p - 1);
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                gtake(yP, yQ);
                --p;
            }
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();                      // every wave's x has landed
            if (own) {
#pragma unroll
                for (int c = 0; c < 3; ++c) yP[c] = inbS[r * 3 + c];
                const float x0 = inbV[r * 3], x1 = inbV[r * 3 + 1], x2 = inbV[r * 3 + 2];
                yQ[0] = fmaf(o[6], x2, fmaf(o[3], x1, o[0] * x0));                              // rotate_rows<true>
                yQ[1] = fmaf(o[7], x2, fmaf(o[4], x1, o[1] * x0));
                yQ[2] = fmaf(o[8], x2, fmaf(o[5], x1, o[2] * x0));
            } else {
#pragma unroll
                for (int c = 0; c < 3; ++c) { yP[c] = 0.f; yQ[c] = 0.f; }
            }
        }
        // Forward output.  A finished band is one float per (row, channel) at stride F in OUT: written by the owners it is 64
        // scattered lines per wave-store (measured: 0.22 of the launch's 0.53 ms at cfg2).  Instead the owners park the band
        // in the stage buffer in ROW order (the staged x is dead after the prologue), and after the step's barrier every
        // thread writes (row, channel) elements i, i + NT, ...: consecutive lanes 24 B apart, 12 lines per wave-store.
        int pending = -1;                         // slot whose band waits in the stage (uniform)
        int waiting = -1;                         // slot of a second band finished at the same step: in the owners' registers
        float wP[3] = {0.f, 0.f, 0.f}, wQ[3] = {0.f, 0.f, 0.f};                 // until the stage is free (rare: last band + low-pass)
        // Every thread is a writer, at the top of the step after the one that parked the band.  (Measured and dropped, kept as
        // compile-time knobs of the ablation builds: only the waves of the part's short slices write, between their gathers
        // and the barrier where they wait for the long slices anyway -- ESCAT_BOTH_WRITERS = 2, ESCAT_BOTH_FLUSH_TOP = 0:
        // 0.48 vs 0.38 ms per launch at cfg2.)
        // The vector track's band is parked in the local frames; the writer of element i = (row, channel c) forms row c of
        // O_row x band (frame_to_global's arithmetic).
#ifndef ESCAT_BOTH_WRITERS
#define ESCAT_BOTH_WRITERS 1              // 1: every wave writes; 2: the upper half of the waves
#endif
#ifndef ESCAT_BOTH_FLUSH_TOP
#define ESCAT_BOTH_FLUSH_TOP 1            // 1: at the top of the step after the one that parked the band; 0: after that step's gathers
#endif
        constexpr int NTW = NT / ESCAT_BOTH_WRITERS, EPW = 3 * ESCAT_BOTH_WRITERS;      // writer threads; elements per writer and track
        // Two bands at a time where the slots are adjacent: the writer keeps band f of its elements in registers until band
        // f + 1 arrives and stores 8 B per element (half the store instructions and L2 write requests).
        float keepS[EPW], keepV[EPW];
        int kept = -1;                            // slot of the kept band (uniform)
        auto put1 = [&](int slot) {
            if (xr < NT - NTW) return;
            int base = xr - (NT - NTW);
            asm volatile("" : "+v"(base));
#pragma unroll
            for (int k = 0; k < EPW; ++k) {
                const int i = base + k * NTW;
                if (i < rows * 3 && !(ESCAT_BOTH_EXP & 4)) {
                    RESs[((int64_t)row0 * 3 + i) * F + slot] = keepS[k];
                    RESv[((int64_t)row0 * 3 + i) * F + slot] = keepV[k];
                }
            }
        };
        auto flush = [&]() {
            if ((ESCAT_BOTH_EXP & 16) || xr < NT - NTW) return;
            int base = xr - (NT - NTW);
            asm volatile("" : "+v"(base));        // (not loop-invariant for the optimiser: no addresses held across the steps)
            const int n3 = rows * 3;
            float vs[EPW], vv[EPW];
#pragma unroll
            for (int k = 0; k < EPW; ++k) {
                const int i = base + k * NTW, ii = i < n3 ? i : 0, row = ii / 3;
                const float *f = sd.frames + (int64_t)gs * 9 + ii * 3;          // row c of the frame: o[3c], o[3c+1], o[3c+2]
                const float x0 = inbV[row * 3], x1 = inbV[row * 3 + 1], x2 = inbV[row * 3 + 2];
                vs[k] = inbS[ii];
                vv[k] = fmaf(f[2], x2, fmaf(f[1], x1, f[0] * x0));               // frame_to_global
            }
            const bool pair = kept >= 0 && pending == kept + 1;
            if (pair) {
#pragma unroll
                for (int k = 0; k < EPW; ++k) {
                    const int i = base + k * NTW;
                    if (i < n3 && !(ESCAT_BOTH_EXP & 4)) {
                        *reinterpret_cast<f2u_k *>(RESs + ((int64_t)row0 * 3 + i) * F + kept) = f2u_k{keepS[k], vs[k]};
                        *reinterpret_cast<f2u_k *>(RESv + ((int64_t)row0 * 3 + i) * F + kept) = f2u_k{keepV[k], vv[k]};
                    }
                }
            } else {
                if (kept >= 0) put1(kept);
#pragma unroll
                for (int k = 0; k < EPW; ++k) { keepS[k] = vs[k]; keepV[k] = vv[k]; }
            }
            kept = pair ? -1 : pending;
        };
        auto park = [&](int slot, const float (&a)[3], const float (&b)[3]) {   // a finished band, row order, into the stage
            if (own) {
#pragma unroll
                for (int c = 0; c < 3; ++c) { inbS[r * 3 + c] = a[c]; inbV[r * 3 + c] = b[c]; }
            }
            pending = slot;
        };
        // the event handler of the forward: closes and single-power slots first, then the band that opens.
        // stage_read: the stage was read out after this step's opening barrier -- a slow wave may still be reading
        auto emit = [&](int ps, bool stage_read) {
            const int nt = prog.s[ps].n;
            for (int k = 0; k < nt; ++k) {
                const int kind = prog.s[ps].t[k].kind, slot = prog.s[ps].t[k].slot;
                if ((kind & 3) == 0) continue;
                float a[3], b[3];
#pragma unroll
                for (int c = 0; c < 3; ++c) { a[c] = (kind & 4) ? -yP[c] : yP[c]; b[c] = (kind & 4) ? -yQ[c] : yQ[c]; }
                if ((kind & 3) == 1) {
#pragma unroll
                    for (int c = 0; c < 3; ++c) { a[c] += hP[c]; b[c] += hQ[c]; }
                }
                // (written with selects: as "if (pending >= 0) waiting = slot; else pending = slot;" the optimiser turns the two
                // cursors into a two-element stack array -- scratch stores whose s_waitcnt vmcnt(0) then waits, every step, for the
                // band stores issued at the step's top: measured +0.13 ms per launch)
                const bool second = pending >= 0;     // a second band at one step: waits in registers
#pragma unroll
                for (int c = 0; c < 3; ++c) { wP[c] = second ? a[c] : wP[c]; wQ[c] = second ? b[c] : wQ[c]; }
                waiting = second ? slot : waiting;
                if (!second) {
                    if (stage_read) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                    if (own) {
#pragma unroll
                        for (int c = 0; c < 3; ++c) { inbS[r * 3 + c] = a[c]; inbV[r * 3 + c] = b[c]; }
                    }
                }
                pending = second ? pending : slot;
            }
            for (int k = 0; k < nt; ++k) {
                const int kind = prog.s[ps].t[k].kind;
                if ((kind & 3) != 0) continue;
#pragma unroll
                for (int c = 0; c < 3; ++c) { hP[c] = (kind & 4) ? -yP[c] : yP[c]; hQ[c] = (kind & 4) ? -yQ[c] : yQ[c]; }
            }
        };
        unsigned baseR = lds0, baseW = lds0 + 2 * VSTRIDE;
        {
            const vf4 vp = {yP[0], yP[1], yP[2], 0.f}, vq = {yQ[0], yQ[1], yQ[2], 0.f};
            if (own) {
                lds_store_at<0>(baseR + offA, vp); lds_store_at<0>(baseR + offB, vp);
                lds_store_at<VSTRIDE>(baseR + offA, vq); lds_store_at<VSTRIDE>(baseR + offB, vq);
            }
            if (xr < 4) {                         // the all-zero row (target of padded slots), copy A of all four vectors
                float *z = smem + rows * 4 + xr;
                z[0] = 0.f; z[VSTRIDE / 4] = 0.f; z[2 * VSTRIDE / 4] = 0.f; z[3 * VSTRIDE / 4] = 0.f;
            }
            if (!BWD && prog.n > 0 && prog.s[0].step == 0) {
                emit(0, true);                    // (x was read out of the stage just now: a band of step 0 waits for everybody)
                p = 1;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        // the next part: its slice words are here (the wait above covered them)
        int swN = 0, rtN = 0, te0N = 0;
        if (has_next) {
            swN = metaS[wave * 8 + (lane & 7)];
            const int *lmN = sd.lane_meta + ((int64_t)(partN * sd.nsl + wave) * 64 + lane) * 3;
            rtN = lmN[0]; te0N = lmN[1];          // (lands during the steps: two registers held, no dependent hop at the part's start)
        }
        if (BWD && p > 0 && prog.s[p - 1].step == T - 1) {    // events at step T - 1: no earlier step could fetch them (rare)
            gload(p - 1);
            gpark(p - 1);
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
        for (int t = 1; t <= T; ++t) {
            const int st = BWD ? T - t : t;
            // backward: this step's events are in the stage already; the NEXT step's (if it has any) are fetched now
            const bool ev_now = BWD && p > 0 && prog.s[p - 1].step == st;
            const int pn = ev_now ? p - 2 : p - 1;
            const bool ev_next = BWD && st > 0 && pn >= 0 && prog.s[pn].step == st - 1;
            if (ev_next) gload(pn);
            bool stage_read = false;              // forward: the band parked during the previous step leaves now
            auto drain = [&]() {
                if (pending >= 0) {
                    flush(); pending = -1; stage_read = true;
                    if (waiting >= 0) {
                        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                        park(waiting, wP, wQ);
                        waiting = -1; stage_read = false;
                    }
                }
            };
            if (!BWD && ESCAT_BOTH_FLUSH_TOP) drain();
            float aP[3] = {0.f, 0.f, 0.f}, aQ[3] = {0.f, 0.f, 0.f};
            op.gather(baseR, w, aP, aQ);
            if (has_tail && tn)
                both_tail<VSTRIDE>(mag, colw, te0, tn, smem + (baseR - lds0) / 4, aP, aQ, sd.pos_of_row);
            if (ev_now) { gtake(aP, aQ); --p; }
            {
                const vf4 vp = {aP[0], aP[1], aP[2], 0.f}, vq = {aQ[0], aQ[1], aQ[2], 0.f};
                if (own) {
                    lds_store_at<0>(baseW + offA, vp); lds_store_at<0>(baseW + offB, vp);
                    lds_store_at<VSTRIDE>(baseW + offA, vq); lds_store_at<VSTRIDE>(baseW + offB, vq);
                }
            }
            if (!BWD) {
                if (!ESCAT_BOTH_FLUSH_TOP) drain();
                if (p < prog.n && prog.s[p].step == t) {
#pragma unroll
                    for (int c = 0; c < 3; ++c) { yP[c] = aP[c]; yQ[c] = aQ[c]; }
                    emit(p, stage_read); ++p;
                }
            } else {
                if (ev_next) {                    // (consecutive event steps: the owners have to be done with the stage first)
                    if (ev_now) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                    gpark(pn);
                }
                if (t == T && own) {              // the owner holds the answer of its row: straight to HBM
                    frame_to_global(aQ, o);
#pragma unroll
                    for (int c = 0; c < 3; ++c) { RESs[orow + c] = aP[c]; RESv[orow + c] = aQ[c]; }
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            const unsigned tmp = baseR; baseR = baseW; baseW = tmp;
        }
        if (!BWD) {                               // what is left after the last step
            if (pending < 0 && kept >= 0) { put1(kept); kept = -1; }
            while (pending >= 0) {
                flush(); pending = -1;
                if (waiting >= 0) {
                    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                    park(waiting, wP, wQ);
                    waiting = -1;
                    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                }
                if (pending < 0 && kept >= 0) { put1(kept); kept = -1; }
            }
        }
        if (!has_next) break;
        if (!BWD) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");     // the stage has been read: the next part's x may land
        li = liN; part = partN; sw = swN; rt = rtN; te0 = te0N;
    }
}

template <int DMAX, int NT, bool BWD>
static int launch_both2(const int *colw, const float *mag, int n_parts, int max_rows, const SellDev &sd, const float *INs,
                        const float *INv, float *RESs, float *RESv, int T, const BothProg &prog, int F, hipStream_t st)
{
    const size_t lds = (size_t)4 * 2 * (NT + 16) * 16 + 2 * (size_t)((max_rows * 3 + 3) & ~3) * 4 + (NT / 64) * 8 * sizeof(int);
    if (lds > LDS_BUDGET) return fail(ESCAT_ERANGE, "both-tracks chain needs %zu B of LDS", lds);
    auto kern = k_chain_both<DMAX, NT, BWD>;
    int rc = set_lds(kern, lds); if (rc) return rc;
    static thread_local size_t cached_lds = (size_t)-1; static thread_local int cached_per_cu = 0, cached_dev = -1;
    int dev_now = 0;
    (void)hipGetDevice(&dev_now);
    if (cached_lds != lds || cached_dev != dev_now) {
        int per = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per, kern, NT, lds) != hipSuccess || per < 1) per = 1;
        cached_per_cu = per; cached_lds = lds; cached_dev = dev_now;
    }
    static const int dbg_per_cu = getenv("ESCAT_DBG_PER_CU") ? atoi(getenv("ESCAT_DBG_PER_CU")) : 0;
    int per_cu = cached_per_cu;
    if (dbg_per_cu > 0 && dbg_per_cu < per_cu) per_cu = dbg_per_cu;
    const int64_t resident = (int64_t)per_cu * escat_cu_count();
    const int grid = (int)(n_parts < resident ? n_parts : resident);
    static const bool verbose = getenv("ESCAT_DBG_VERBOSE") != nullptr;
    if (verbose) fprintf(stderr, "k_chain_both<%d,%d,%d>: lds %zu B, %d workgroups per CU, grid %d\n", DMAX, NT, (int)BWD, lds, cached_per_cu, grid);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NT), lds, st, colw, mag, sd, INs, INv, RESs, RESv, T, prog, F, n_parts, max_rows);
    return ESCAT_OK;
}

template <int DMAX, bool BWD>
static int launch_both_d(int max_rows, const int *colw, const float *mag, int n_parts, const SellDev &sd, const float *INs,
                         const float *INv, float *RESs, float *RESv, int T, const BothProg &prog, int F, hipStream_t st)
{
    const int NT = max_rows <= 256 ? 256 : 512;
    if (sd.nsl * 64 != NT) return fail(ESCAT_EINVAL, "escat_sell.slices_per_part=%d does not match parts of <= %d rows", sd.nsl, max_rows);
    if (NT == 256) return launch_both2<DMAX, 256, BWD>(colw, mag, n_parts, max_rows, sd, INs, INv, RESs, RESv, T, prog, F, st);
    return launch_both2<DMAX, 512, BWD>(colw, mag, n_parts, max_rows, sd, INs, INv, RESs, RESv, T, prog, F, st);
}

int escat_launch_both_sell(bool bwd, int max_rows, const int *colw, const float *mag, int n_parts, const SellDev &sd,
                           const float *INs, const float *INv, float *RESs, float *RESv, int T, const BothProg &prog, int F,
                           hipStream_t st)
{
    switch (sd.dmax) {
#define CASE(d) case d: return bwd ? launch_both_d<d, true>(max_rows, colw, mag, n_parts, sd, INs, INv, RESs, RESv, T, prog, F, st) \
                                   : launch_both_d<d, false>(max_rows, colw, mag, n_parts, sd, INs, INv, RESs, RESv, T, prog, F, st);
        CASE(8) CASE(16) CASE(20) CASE(24)
#undef CASE
    }
    return fail(ESCAT_EINVAL, "escat_sell.dmax=%d: per-channel CSR packs must use 8, 16, 20 or 24", sd.dmax);
}
