// escat_dev.h -- device-side declarations shared by the translation units of libescat.so (not part of the ABI)
#ifndef ESCAT_DEV_H
#define ESCAT_DEV_H
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/escat.h"
#include "escat_internal.h"

#define fail escat_fail
#define CHECK_LAUNCH(what)                                                              \
    do { hipError_t e_ = hipGetLastError();                                             \
         if (e_ != hipSuccess) return fail(ESCAT_ELAUNCH, "%s: %s", what, hipGetErrorString(e_)); } while (0)

// ------------------------------------------------------------------ event tables (kernel arguments)
struct Ev { int step, slot, sign, chan; };          // sign: +1/-1; bit 8 set (|0x100) = first touch of (slot,chan): store
struct EvTable { int n; Ev e[ESCAT_MAX_EVENTS]; };  // whole chain (fused kernels)
#define STEP_MAX_EV 24
struct StepEv { int n; Ev e[STEP_MAX_EV]; };        // one step (multi-launch path)

#define EV_STORE 0x100
#define EV_ABS   0x200
// encoding: bit0 = valid, bit1 = negative, bit8 = store, bit9 = the slot is final after this event: keep |value|

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
    float v = (s & 2) ? -y : y;
    if (!(s & EV_STORE)) v += o[slot];
    if (s & EV_ABS) v = fabsf(v);             // the slot's last event of a |.| bank (ESCAT_EVENT_ABS; off in the drop-in path)
    o[slot] = v;
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

// Layout 1 packs (escat_sell.layout, built by escat_pack.hip): an LDS row is a POSITION -- the index, inside the part's
// workgroup, of the thread that owns the row -- so the owner stores its new vector at its own thread index, and copy B
// permutes rows inside aligned groups of 16 only: the 8 consecutive lanes of a ds_write_b128 lane group then hit 8
// distinct 16-byte slots in BOTH copies (conflict-free stores), while a row's bank slot in copy B, (p + 1 + (p >> 5) % 15)
// & 15, always differs from its slot in copy A (p & 15): the planner keeps its two choices per gather.
__host__ __device__ __forceinline__ int sell_swzB(int p) { return (p & ~15) | ((p + 1 + ((p >> 5) % 15)) & 15); }

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
                                                  // global coordinates (the chain runs D Pt^t D^T instead of Pt^t)
    int layout;                                   // 0: LDS rows = local rows; 1: LDS rows = owner positions (see sell_swzB)
    const int *pos_of_row;                        // layout 1: global row -> position (for the streamed tails)
    const void *half_val; const unsigned *half_aux;   // half-valued BSR-3 pack (escat_sell_bsr3h.hip), NULL otherwise
    const signed char *ent; const int *rowptr;        // "direct" BSR-3 (no value pack): the plan's entry of every slot, the
};                                                    // canonical row pointers (escat_sell.ent_of_slot)

#define SLOT_GROUP 4      // granule of the packed LDS-row pairs (slots are padded to it)
#define GATHER_AHEAD 3     // LDS gathers in flight ahead of the FMAs


#define LDS_BUDGET (160 * 1024)

template <typename K> static int set_lds(K kern, size_t bytes)
{
    if (bytes > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e != hipSuccess) return fail(ESCAT_ELAUNCH, "hipFuncSetAttribute(%zu B LDS): %s", bytes, hipGetErrorString(e));
    }
    return ESCAT_OK;
}

// defined in escat.hip
__attribute__((visibility("hidden"))) int escat_cu_count();
__attribute__((visibility("hidden"))) size_t escat_sell_lds_bytes(int C, int CP, int max_rows, int F, bool bwd, int want, int *early, int orders);
// register-resident launchers, one translation unit per operator family (escat_sell_*.hip)
#define ESCAT_SELL_ARGS int max_rows, const int *ci, const float *v, int n_parts, const SellDev &sd, const float *IN, float *RES, \
                        int T, const EvTable &tab, int F, hipStream_t st
__attribute__((visibility("hidden"))) int escat_launch_bsr3_sell(bool bwd, ESCAT_SELL_ARGS);
__attribute__((visibility("hidden"))) int escat_launch_csr_sell_c12(bool bwd, int C, ESCAT_SELL_ARGS);
__attribute__((visibility("hidden"))) int escat_launch_csr_sell_c34(bool bwd, int C, ESCAT_SELL_ARGS);
__attribute__((visibility("hidden"))) int escat_launch_csr_sell_c56(bool bwd, int C, ESCAT_SELL_ARGS);
__attribute__((visibility("hidden"))) int escat_launch_csr_sell_c78(bool bwd, int C, ESCAT_SELL_ARGS);
__attribute__((visibility("hidden"))) int escat_launch_csrv3_sell(bool bwd, ESCAT_SELL_ARGS);
__attribute__((visibility("hidden"))) int escat_launch_bsr3h_sell(bool bwd, ESCAT_SELL_ARGS);

// both tracks in one launch (escat_sell_both.hip): the event table as a program of event steps.  A term's kind: role
// (0 = the band that opens at this step: hold <- +-y; 1 = the open band closes: OUT[slot] = hold +- y; 2 = a slot that is a
// single power: OUT[slot] = +-y) | 4 if the power enters negatively.  Terms are in the caller's event order (the backward
// adds the gradients in that order, as apply_events does).
struct BothTerm { int slot, kind; };       // (ints: scalar loads; shorts come through vector memory and its counter)
#define BOTH_MAX_STEPS 24
#define BOTH_MAX_TERMS 3
struct BothStep { int step, n; BothTerm t[BOTH_MAX_TERMS]; };
struct BothProg { int n; BothStep s[BOTH_MAX_STEPS]; };
__attribute__((visibility("hidden"))) int escat_launch_both_sell(bool bwd, int max_rows, const int *colw, const float *mag, int n_parts,
                                                                 const SellDev &sd, const float *INs, const float *INv, float *RESs,
                                                                 float *RESv, int T, const BothProg &prog, int F, hipStream_t st);
#endif
