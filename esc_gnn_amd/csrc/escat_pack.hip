// escat_pack.hip -- operator ingest on the GPU: the SELL-64 pack of the register-resident chain kernels (escat_sell,
// include/escat.h) and the transposed operator for the backward walk, built by HIP kernels instead of torch index ops
// (round 1: ~120 lines of sort / scatter calls in esc_gnn_amd/operators.py, 80 ms per 4096-graph batch).
//
//   escat_sell_pack_plan   row-length histogram -> capacity dmax; per part: rows ordered by length (stable, descending),
//                          slice widths; prefix sums of the slices' slot / pair offsets.  Two small read-backs.
//   escat_sell_pack_fill   one wave per slice: bank-conflict-aware slot + copy planning (as escat_sell_plan_slots),
//                          then the slice's values, packed LDS rows and lane records written with coalesced stores.
//   escat_transpose_sym    A^T on A's own pattern when the pattern is symmetric (undirected graphs: P and Q always):
//                          every entry (i, j) finds (j, i) by binary search in row j.
//
// Layouts (escat_sell_dims.layout)
//   0  LDS rows are the part's local row indices; copy B rotates bits 0-3 to the top of a 256-row block (sell_copyB)
//   1  LDS rows are POSITIONS = the lane index, inside the part, of the thread that owns the row (rows sorted by
//      length), copy B swizzles inside 16-row groups (sell_swzB).  The owning thread then stores its new vector row at
//      its own thread index: both copies' ds_write_b128 are conflict-free (8 consecutive lanes -> 8 distinct 16-B slots),
//      where layout 0 stores to random rows (~2.6-way conflicts on every store of every step).
#include <vector>
#include "escat_dev.h"

#define PACK_MAX_ROWS 512

// ------------------------------------------------------------------ histogram of row lengths (clamped to 33)
// (nearly every row of a kNN graph has one of three or four lengths: LDS atomics on those bins serialise; the lanes that
// share a length are counted with one ballot instead)
__global__ void __launch_bounds__(256)
k_pack_hist(const int *__restrict__ rowptr, int64_t n, unsigned *__restrict__ hist)
{
    __shared__ unsigned h[34];
    if (threadIdx.x < 34) h[threadIdx.x] = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    for (int64_t r0 = (int64_t)blockIdx.x * 256 + (threadIdx.x & ~63); r0 < n; r0 += (int64_t)gridDim.x * 256) {
        const int64_t r = r0 + lane;
        int d = -1;
        if (r < n) { d = rowptr[r + 1] - rowptr[r]; d = d < 33 ? d : 33; }
        unsigned long long todo = __ballot(d >= 0);
        while (todo) {                                           // one round per distinct length in the wave
            const int dl = __builtin_amdgcn_readlane(d, __builtin_ctzll(todo));
            const unsigned long long same = __ballot(d == dl);
            if (lane == __builtin_ctzll(todo)) atomicAdd(&h[dl], (unsigned)__builtin_popcountll(same));
            todo &= ~same;
        }
    }
    __syncthreads();
    if (threadIdx.x < 34 && h[threadIdx.x]) atomicAdd(&hist[threadIdx.x], h[threadIdx.x]);
}

// capacity: the smallest instantiated one that leaves at most 5e-5 of the rows a streamed tail; on the
// device, so that the plan needs one host round trip instead of two.  hist[40] = dmax.
__global__ void k_pack_pick_dmax(unsigned *__restrict__ hist, int64_t n, int values_per_entry)
{
    if (threadIdx.x) return;
    const int caps1[] = {8, 16, 20, 24, 32}, caps3[] = {8, 16, 20, 24, 24}, caps9[] = {8, 12, 16, 20, 22};
    const int *caps = values_per_entry == 1 ? caps1 : values_per_entry == 3 ? caps3 : caps9;
    const long long allowed = (long long)(5e-5 * (double)n);
    int dmax = caps[4];
    for (int i = 4; i >= 0; --i) {
        long long longer = 0;
        for (int d = caps[i] + 1; d < 34; ++d) longer += hist[d];
        if (longer <= allowed) dmax = caps[i];
    }
    hist[40] = (unsigned)dmax;
}

// ------------------------------------------------------------------ per part: order rows by length, slice widths
// perm[(part*nsl)*64 + k] = local row at lane position k (-1: none); pos_of_row[gs + r] = k; w[part*nsl + s] = width.
// Stable counting sort on the (few) distinct lengths: a ballot per key value gives every row its rank among the rows of
// its wave with the same key, per-wave counts in LDS give the rest.
__global__ void __launch_bounds__(PACK_MAX_ROWS)
k_pack_order(const int *__restrict__ rowptr, const int *__restrict__ part_ptr, const unsigned *__restrict__ dmax_p, int nsl,
             int *__restrict__ perm, int *__restrict__ pos_of_row, int *__restrict__ w)
{
    __shared__ int cnt[PACK_MAX_ROWS / 64][34], base[34];
    const int dmax = (int)*dmax_p;
    const int part = blockIdx.x, gs = part_ptr[part], rows = part_ptr[part + 1] - gs;
    const int r = threadIdx.x, lane = r & 63, wv = r >> 6;
    int d = 0, key = -1;
    if (r < rows) { d = rowptr[gs + r + 1] - rowptr[gs + r]; d = d < dmax ? d : dmax; key = dmax - d; }
    int mine = 0;
    for (int k = 0; k <= dmax; ++k) {
        const unsigned long long m = __ballot(key == k);
        if (lane == 0) cnt[wv][k] = __builtin_popcountll(m);
        if (key == k) mine = __builtin_popcountll(m & ((1ull << lane) - 1ull));
    }
    __syncthreads();
    if (r <= dmax) {                               // rows of all waves with a smaller key
        int b = 0;
        for (int k = 0; k < r; ++k)
            for (int q = 0; q < PACK_MAX_ROWS / 64; ++q) b += cnt[q][k];
        base[r] = b;
    }
    __syncthreads();
    int *pp = perm + (int64_t)part * nsl * 64;
    if (r < rows) {
        int rank = base[key] + mine;
        for (int q = 0; q < wv; ++q) rank += cnt[q][key];
        pp[rank] = r;
        pos_of_row[gs + r] = rank;
        if ((rank & 63) == 0) w[part * nsl + (rank >> 6)] = d;      // descending: a slice's first lane has its longest row
    }
    for (int k = rows + r; k < nsl * 64; k += PACK_MAX_ROWS) pp[k] = -1;
    if (r < nsl && r * 64 >= rows) w[part * nsl + r] = 0;
}

// ------------------------------------------------------------------ exclusive scans over the slices (one workgroup)
// tiles of 4096 slices: four consecutive slices per thread (coalesced), a block scan per tile, the carry between tiles
__global__ void __launch_bounds__(1024)
k_pack_scan(const int *__restrict__ w, int n_slices, int *__restrict__ doff, int *__restrict__ poff)
{
    __shared__ int wd[16], wp[16];
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    int carry_d = 0, carry_p = 0;
    for (int s0 = 0; s0 < n_slices; s0 += 4096) {
        const int a = s0 + 4 * t;
        int v[4], q[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) { v[i] = a + i < n_slices ? w[a + i] : 0; q[i] = ((v[i] + 3) / 4) * 2; }
        const int sd_ = v[0] + v[1] + v[2] + v[3], sp_ = q[0] + q[1] + q[2] + q[3];
        int xd = sd_, xp = sp_;
#pragma unroll
        for (int m = 1; m < 64; m <<= 1) {
            const int od = __shfl_up(xd, m, 64), op = __shfl_up(xp, m, 64);
            if (lane >= m) { xd += od; xp += op; }
        }
        if (lane == 63) { wd[wv] = xd; wp[wv] = xp; }
        __syncthreads();
        int bd = carry_d, bp = carry_p, td = 0, tp = 0;
        for (int i = 0; i < 16; ++i) { if (i < wv) { bd += wd[i]; bp += wp[i]; } td += wd[i]; tp += wp[i]; }
        __syncthreads();
        int d = bd + xd - sd_, p = bp + xp - sp_;
#pragma unroll
        for (int i = 0; i < 4; ++i) if (a + i < n_slices) { doff[a + i] = d; poff[a + i] = p; d += v[i]; p += q[i]; }
        carry_d += td; carry_p += tp;
    }
    if (t == 0) { doff[n_slices] = carry_d; poff[n_slices] = carry_p; }
}

struct __attribute__((packed, aligned(4))) f4u { float x, y, z, w; };     // a 16-byte load that is only dword-aligned

// one entry's V values (e < 0: a padded slot, zeros)
template <int V>
__device__ __forceinline__ void get_values(float (&b)[V], const float *__restrict__ val, int64_t e)
{
    if (e < 0) {
#pragma unroll
        for (int q = 0; q < V; ++q) b[q] = 0.f;
    } else if constexpr (V == 9) {                 // 36 contiguous bytes, dword-aligned: two 16-byte loads and one dword
        const f4u lo = *reinterpret_cast<const f4u *>(val + e * 9), hi = *reinterpret_cast<const f4u *>(val + e * 9 + 4);
        b[0] = lo.x; b[1] = lo.y; b[2] = lo.z; b[3] = lo.w; b[4] = hi.x; b[5] = hi.y; b[6] = hi.z; b[7] = hi.w;
        b[8] = val[e * 9 + 8];
    } else {
#pragma unroll
        for (int q = 0; q < V; ++q) b[q] = val[e * V + q];
    }
}

// ... into slot `slot` (absolute) of the packed array, lane `lane`.  SIGNED: b[0] is the magnitude (one plane)
template <int V, bool SIGNED>
__device__ __forceinline__ void store_values(float *__restrict__ val_out, int64_t slot, int lane, const float (&b)[V])
{
    constexpr int PER = V == 9 ? 576 : (V == 3 && !SIGNED ? 192 : 64);
    float *o = val_out + slot * PER;
    if constexpr (V == 9) {
        // pieces b00 b10 b01 b11 | b02 b12 b20 b21 | b22   (bij = row i, column j: b[3*i + j])
        *reinterpret_cast<float4 *>(o + lane * 4) = make_float4(b[0], b[3], b[1], b[4]);
        *reinterpret_cast<float4 *>(o + 256 + lane * 4) = make_float4(b[2], b[5], b[6], b[7]);
        o[512 + lane] = b[8];
    } else if constexpr (V == 3 && !SIGNED) {
#pragma unroll
        for (int c = 0; c < 3; ++c) o[c * 64 + lane] = b[c];
    } else {
        o[lane] = b[0];
    }
}

template <int V, bool SIGNED>
__device__ __forceinline__ void put_values(float *__restrict__ val_out, int64_t slot, int lane, const float *__restrict__ val, int64_t e)
{
    if constexpr (V == 3 && SIGNED) {              // val is (nnz,) magnitudes here
        float b[3] = {e >= 0 ? val[e] : 0.f, 0.f, 0.f};
        store_values<3, true>(val_out, slot, lane, b);
    } else {
        float b[V];
        get_values<V>(b, val, e);
        store_values<V, SIGNED>(val_out, slot, lane, b);
    }
}

// values only, for an operator that shares pattern, capacity and layout with one already packed (its transpose on a
// symmetric pattern; P and Q of one batch): the plan (ent_of_slot: which entry of the row sits in each slot) is reused,
// lc / lane_meta / pos_of_row are shared.
template <int V>
__global__ void __launch_bounds__(256)
k_pack_refill(const int *__restrict__ rowptr, const float *__restrict__ val, const int *__restrict__ part_ptr,
              const int *__restrict__ perm, const int *__restrict__ doff_, const signed char *__restrict__ ent, int nsl,
              int n_slices, float *__restrict__ val_out)
{
    const int s = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (s >= n_slices) return;
    const int gs = part_ptr[s / nsl];
    const int r = perm[(int64_t)s * 64 + lane];
    const int doff = doff_[s], w = doff_[s + 1] - doff;
    const int64_t e0 = r >= 0 ? rowptr[gs + r] : 0;
    // four slots per trip: their plan bytes, then their values, are loaded together (independent loads in flight), then
    // stored -- one slot at a time is two dependent global round trips per slot
    for (int k0 = 0; k0 < w; k0 += 4) {
        int j[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) j[u] = k0 + u < w ? (int)ent[(int64_t)(doff + k0 + u) * 64 + lane] : -1;
        float b[4][V];
#pragma unroll
        for (int u = 0; u < 4; ++u) get_values<V>(b[u], val, j[u] >= 0 ? e0 + j[u] : -1);
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (k0 + u < w) store_values<V, false>(val_out, doff + k0 + u, lane, b[u]);
    }
}

// the same for a BSR-3 operator whose values are all exactly representable as IEEE halves (escat_sell_bsr3h.hip): the
// nine halves of a block as four pairs + b22 beside the slot's LDS offset.  *not_exact is raised by any value v with
// float(half(v)) != v (NaNs included): the pack is then unusable and the caller keeps the fp32 pack.
__device__ __forceinline__ unsigned half_bits_checked(float v, int &bad)
{
    const _Float16 h = (_Float16)v;
    if (__float_as_uint((float)h) != __float_as_uint(v) || v != v) bad = 1;      // (a quiet NaN survives the round trip: refuse it by name)
    unsigned short u;
    __builtin_memcpy(&u, &h, 2);
    return u;
}
__global__ void __launch_bounds__(256)
k_pack_refill_half(const int *__restrict__ rowptr, const float *__restrict__ val, const int *__restrict__ part_ptr,
                   const int *__restrict__ perm, const int *__restrict__ doff_, const int *__restrict__ poff_,
                   const signed char *__restrict__ ent, const unsigned *__restrict__ lc, int nsl, int n_slices,
                   uint4 *__restrict__ hv_out, unsigned *__restrict__ aux_out, int *__restrict__ not_exact)
{
    const int s = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (s >= n_slices) return;
    const int gs = part_ptr[s / nsl];
    const int r = perm[(int64_t)s * 64 + lane];
    const int doff = doff_[s], w = doff_[s + 1] - doff, poff = poff_[s];
    const int64_t e0 = r >= 0 ? rowptr[gs + r] : 0;
    int bad = 0;
    for (int k0 = 0; k0 < w; k0 += 4) {
        int j[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) j[u] = k0 + u < w ? (int)ent[(int64_t)(doff + k0 + u) * 64 + lane] : -1;
        float b[4][9];
#pragma unroll
        for (int u = 0; u < 4; ++u) get_values<9>(b[u], val, j[u] >= 0 ? e0 + j[u] : -1);
        const unsigned l0 = lc[(int64_t)(poff + k0 / 2) * 64 + lane], l1 = lc[(int64_t)(poff + k0 / 2 + 1) * 64 + lane];
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (k0 + u < w) {
                unsigned hb[9];
#pragma unroll
                for (int q = 0; q < 9; ++q) hb[q] = half_bits_checked(b[u][q], bad);
                const unsigned row = ((u < 2 ? l0 : l1) >> ((u & 1) * 16)) & 0xffffu;
                const int64_t o = (int64_t)(doff + k0 + u) * 64 + lane;
                // pairs (b00 b10)(b01 b11)(b02 b12)(b20 b21), low | high << 16   (bij = b[3*i + j])
                hv_out[o] = make_uint4(hb[0] | hb[3] << 16, hb[1] | hb[4] << 16, hb[2] | hb[5] << 16, hb[6] | hb[7] << 16);
                aux_out[o] = hb[8] | (row * 16u) << 16;
            }
    }
    if (bad) *not_exact = 1;
}

// the same for a sign-bit pack (values_per_entry 3 given as |value| + sign bits, escat_sell.val_planes = 4) on the plan of a
// pack with the same pattern, capacity and layout (P's own pack, typically): the magnitudes, and the shared LDS rows with
// the entries' sign bits in bits 11..13 of every 16-bit field.
__global__ void __launch_bounds__(256)
k_pack_refill_signed(const int *__restrict__ rowptr, const float *__restrict__ mag, const int *__restrict__ sign_bits,
                     const int *__restrict__ part_ptr, const int *__restrict__ perm, const int *__restrict__ doff_,
                     const int *__restrict__ poff_, const signed char *__restrict__ ent, const unsigned *__restrict__ lc,
                     int nsl, int n_slices, float *__restrict__ val_out, unsigned *__restrict__ lc_out)
{
    const int s = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (s >= n_slices) return;
    const int gs = part_ptr[s / nsl];
    const int r = perm[(int64_t)s * 64 + lane];
    const int doff = doff_[s], w = doff_[s + 1] - doff, poff = poff_[s];
    const int64_t e0 = r >= 0 ? rowptr[gs + r] : 0;
    for (int k0 = 0; k0 < w; k0 += 4) {           // four slots = two LDS-row pairs per trip
        int j[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) j[u] = k0 + u < w ? (int)ent[(int64_t)(doff + k0 + u) * 64 + lane] : -1;
        float m[4]; unsigned sb[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            m[u] = j[u] >= 0 ? mag[e0 + j[u]] : 0.f;
            sb[u] = j[u] >= 0 ? ((unsigned)sign_bits[e0 + j[u]] & 7u) << 11 : 0u;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (k0 + u < w) val_out[(int64_t)(doff + k0 + u) * 64 + lane] = m[u];
        const int64_t o = (int64_t)(poff + k0 / 2) * 64 + lane;
        lc_out[o] = lc[o] | sb[0] | (sb[1] << 16);
        lc_out[o + 64] = lc[o + 64] | sb[2] | (sb[3] << 16);
    }
}

// ------------------------------------------------------------------ plan + fill, one wave per slice
// V: floats per entry in `val` (1 scalar, 3 one per channel, 9 a 3x3 block); SIGNED (V == 3 input given as |value| +
// sign bits): one plane of magnitudes, the three signs in bits 11..13 of the entry's 16-bit LDS-row field.
template <int V, bool SIGNED, int LAYOUT>
__global__ void __launch_bounds__(64)
k_pack_fill(const int *__restrict__ rowptr, const int *__restrict__ col, const float *__restrict__ val,
            const int *__restrict__ sign_bits, const int *__restrict__ part_ptr, const int *__restrict__ perm,
            const int *__restrict__ pos_of_row, const int *__restrict__ doff_, const int *__restrict__ poff_,
            int dmax, int nsl, float *__restrict__ val_out, unsigned *__restrict__ lc_out, int *__restrict__ lane_meta,
            int *__restrict__ slot_of_e, int *__restrict__ lc_of_e, signed char *__restrict__ ent_out, int plan)
{
    __shared__ short cols[64][33];                 // LDS row candidates (copy A) of each lane's kept entries
    __shared__ signed char ent_of_slot[64][32];    // after planning: which entry sits in slot k (-1: none)
    __shared__ unsigned short row_of_slot[64][32]; // and the LDS row (copy A or B) it reads
    __shared__ volatile unsigned long long load[4];
    const int s = blockIdx.x, lane = threadIdx.x;
    const int part = s / nsl;
    const int gs = part_ptr[part], rows = part_ptr[part + 1] - gs;
    const int R = sell_padR(rows);
    const int r = perm[(int64_t)s * 64 + lane];
    const int doff = doff_[s], poff = poff_[s], w = doff_[s + 1] - doff;
    int e0 = 0, deg = 0, cnt = 0;
    if (r >= 0) { e0 = rowptr[gs + r]; deg = rowptr[gs + r + 1] - e0; cnt = deg < dmax ? deg : dmax; }
    for (int j = 0; j < cnt; ++j) {
        const int c = col[e0 + j];
        cols[lane][j] = (short)(LAYOUT == 1 ? pos_of_row[c] : c - gs);
    }
    for (int k = 0; k < 32; ++k) { ent_of_slot[lane][k] = -1; row_of_slot[lane][k] = (unsigned short)rows; }
    unsigned rem = cnt >= 32 ? 0xffffffffu : ((1u << cnt) - 1u);
    const int l32 = lane & 31;
    const bool g0 = l32 < 4 || (l32 >= 12 && l32 < 16) || (l32 >= 20 && l32 < 28);
    const int g = (lane >> 5) * 2 + (g0 ? 0 : 1);
    int turn = 0;                                  // rank of this lane inside its ds_read_b128 lane group
    for (int q = 0; q < l32; ++q) {
        const bool q0 = q < 4 || (q >= 12 && q < 16) || (q >= 20 && q < 28);
        turn += (q0 == g0);
    }
    const unsigned long long gmask32 = g0 ? 0x0FF0F00FULL : 0xF00F0FF0ULL;
    const unsigned long long gmask = (lane >> 5) ? (gmask32 << 32) : gmask32;
    const int zslot = rows & 15;
    for (int k = 0; k < w; ++k) {
        const bool padded = k >= cnt;
        if (!plan) {                               // unplanned pack: entries in row order, copy A (an operator used once
            if (!padded) {                         // is not worth the milliseconds of the matching)
                ent_of_slot[lane][k] = (signed char)k;
                row_of_slot[lane][k] = (unsigned short)cols[lane][k];
                if (slot_of_e) { slot_of_e[e0 + k] = k; lc_of_e[e0 + k] = cols[lane][k]; }
            }
            continue;
        }
        const unsigned long long pm = __ballot(padded);
        if constexpr (LAYOUT == 1) {
            // Layout-1 packs: a MAXIMUM MATCHING of the lane group's 16 lanes onto the 16 bank slots instead of the
            // greedy turn-taking below.  Every lane offers the slots its remaining entries can reach (two per entry: copy
            // A and copy B); augmenting paths (Kuhn) give every lane a slot of its own whenever that is possible at all
            // -- simulated multiplicity 1.02 against the greedy's 1.5-1.6.  All 16 lanes of a group run the same
            // search redundantly on nibble-packed registers (no LDS, no divergence between the groups of a wave).
            unsigned mycand = 0;
            if (!padded) {
                unsigned m = rem;
                while (m) {
                    const int j = __ffs(m) - 1;
                    m &= m - 1;
                    const int a = cols[lane][j];
                    mycand |= 1u << (a & 15);
                    mycand |= 1u << (sell_swzB(a) & 15);       // (R is a multiple of 16: the slot of R + swzB(a))
                }
            }
            constexpr int idx0[16] = {0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27};
            constexpr int idx1[16] = {4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31};
            unsigned long long cw[4] = {0ULL, 0ULL, 0ULL, 0ULL};
            unsigned active = 0;
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const unsigned ct = (unsigned)__shfl((int)mycand, (lane & 32) + (g0 ? idx0[t] : idx1[t]), 64);
                cw[t >> 2] |= (unsigned long long)ct << (16 * (t & 3));
                if (ct) active |= 1u << t;
            }
            auto cand_of = [&](int x) -> unsigned {
                const unsigned long long v = x < 8 ? (x < 4 ? cw[0] : cw[1]) : (x < 12 ? cw[2] : cw[3]);
                return (unsigned)(v >> (16 * (x & 3))) & 0xffffu;
            };
            auto nget = [](unsigned long long v, int i) -> int { return (int)((v >> (4 * i)) & 15ULL); };
            auto nset = [](unsigned long long &v, int i, int x) { v = (v & ~(15ULL << (4 * i))) | ((unsigned long long)x << (4 * i)); };
            const unsigned dummy = (pm & gmask) ? (1u << zslot) : 0u;     // padded lanes read the zero row (copy A)
            unsigned occupied = dummy, mmatched = 0;
            unsigned long long bin_member = 0ULL, member_bin = 0ULL;
            for (int u = 0; u < 16; ++u) {
                if (!((active >> u) & 1u)) continue;
                unsigned visited = dummy;
                unsigned long long prev = 0ULL, queue = (unsigned long long)u;
                int qh = 0, qt = 1, found = -1;
                while (qh < qt && found < 0) {
                    const int x = nget(queue, qh); ++qh;
                    unsigned m = cand_of(x) & ~visited;
                    const unsigned f = m & ~occupied;
                    if (f) { found = __ffs(f) - 1; nset(prev, found, x); break; }
                    while (m) {
                        const int b = __ffs(m) - 1;
                        m &= m - 1;
                        visited |= 1u << b;
                        nset(prev, b, x);
                        if (qt < 16) { nset(queue, qt, nget(bin_member, b)); ++qt; }
                    }
                }
                if (found >= 0) {
                    int b = found;
                    for (int guard = 0; guard < 17; ++guard) {
                        const int x = nget(prev, b);
                        const bool had = (mmatched >> x) & 1u;
                        const int nb = nget(member_bin, x);
                        nset(bin_member, b, x); occupied |= 1u << b;
                        nset(member_bin, x, b); mmatched |= 1u << x;
                        if (x == u || !had) break;
                        b = nb;
                    }
                }
            }
            // lanes the matching could not seat (fewer free slots than lanes): the least-loaded slot they can reach
            unsigned long long loads = 0ULL;
            for (int b = 0; b < 16; ++b) if ((occupied >> b) & 1u) loads |= 1ULL << (4 * b);
            for (int u = 0; u < 16; ++u) {
                if (!((active >> u) & 1u) || ((mmatched >> u) & 1u)) continue;
                unsigned m = cand_of(u);
                int best = 99, bb = 0;
                while (m) { const int b = __ffs(m) - 1; m &= m - 1; const int c = nget(loads, b); if (c < best) { best = c; bb = b; } }
                if (nget(loads, bb) < 15) loads += 1ULL << (4 * bb);
                nset(member_bin, u, bb);
            }
            if (!padded) {
                const int mybin = nget(member_bin, turn);
                unsigned m = rem;
                int bj = -1, bpos = 0;
                while (m) {
                    const int j = __ffs(m) - 1;
                    m &= m - 1;
                    const int a = cols[lane][j];
                    if ((a & 15) == mybin) { bj = j; bpos = a; break; }
                    const int b = R + sell_swzB(a);
                    if ((b & 15) == mybin) { bj = j; bpos = b; break; }
                }
                if (bj < 0) { bj = __ffs(rem) - 1; bpos = cols[lane][bj]; }      // (cannot happen: mybin came from mycand)
                ent_of_slot[lane][k] = (signed char)bj;
                row_of_slot[lane][k] = (unsigned short)bpos;
                if (slot_of_e) { slot_of_e[e0 + bj] = k; lc_of_e[e0 + bj] = bpos; }
                rem &= ~(1u << bj);
            }
            continue;
        }
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
                    const int b = R + (LAYOUT == 1 ? sell_swzB(a) : sell_copyB(a, R));
                    const int cA = (int)((L >> (4 * (a & 15))) & 15ULL), cB = (int)((L >> (4 * (b & 15))) & 15ULL);
                    if (cA < best) { best = cA; bj = j; bpos = a; }
                    if (cB < best) { best = cB; bj = j; bpos = b; }
                    if (best == 0) break;
                }
                if (((L >> (4 * (bpos & 15))) & 15ULL) < 15ULL) L += 1ULL << (4 * (bpos & 15));
                load[g] = L;
                ent_of_slot[lane][k] = (signed char)bj;
                row_of_slot[lane][k] = (unsigned short)bpos;
                if (slot_of_e) { slot_of_e[e0 + bj] = k; lc_of_e[e0 + bj] = bpos; }
                rem &= ~(1u << bj);
            }
        }
    }
    // ---- values: slot-major, every store of the wave one contiguous run
    for (int k = 0; k < w; ++k) {
        const int j = ent_of_slot[lane][k];
        put_values<V, SIGNED>(val_out, doff + k, lane, val, j >= 0 ? e0 + j : -1);
        if (ent_out) ent_out[(int64_t)(doff + k) * 64 + lane] = (signed char)j;
    }
    // ---- packed LDS rows: pairs cover w rounded up to 4 slots; slots past the row's end read the zero row
    const int npair = ((w + 3) / 4) * 2;
    for (int q = 0; q < npair; ++q) {
        unsigned f[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int k = 2 * q + h;
            unsigned v = k < 32 ? row_of_slot[lane][k] : (unsigned)rows;
            if (k >= w) v = (unsigned)rows;
            if constexpr (SIGNED) {
                const int j = k < w ? ent_of_slot[lane][k] : -1;
                if (j >= 0) v |= (unsigned)sign_bits[e0 + j] << 11;
            }
            f[h] = v;
        }
        lc_out[(int64_t)(poff + q) * 64 + lane] = f[0] | (f[1] << 16);
    }
    // ---- the lane's start-up record
    int *lm = lane_meta + ((int64_t)s * 64 + lane) * 3;
    const int tail_n = deg > dmax ? deg - dmax : 0;
    lm[0] = r >= 0 ? (r | (tail_n << 10)) : 0x3ff;
    lm[1] = r >= 0 ? e0 + dmax : 0;
    lm[2] = lane == 0 ? doff : lane == 1 ? poff : lane == 2 ? w : lane == 3 ? gs : lane == 4 ? rows : 0;
}

// ------------------------------------------------------------------ host side
static inline hipStream_t S(void *s) { return (hipStream_t)s; }

// workspace (ints): hist[64] | w[ns] | doff[ns+1] | poff[ns+1] | perm[ns*64] | pos_of_row[n]
struct PackWs { unsigned *hist; int *w, *doff, *poff, *perm, *pos; };
static size_t pack_ws_ints(int64_t n, int64_t ns) { return 64 + (size_t)ns + 2 * ((size_t)ns + 1) + (size_t)ns * 64 + (size_t)n + 16; }
static PackWs pack_ws(void *ws, int64_t n, int64_t ns)
{
    int *p = (int *)ws;
    PackWs o;
    o.hist = (unsigned *)p; p += 64;
    o.w = p; p += ns;
    o.doff = p; p += ns + 1;
    o.poff = p; p += ns + 1;
    o.perm = p; p += ns * 64;
    o.pos = p;
    return o;
}

extern "C" size_t escat_sell_pack_workspace_bytes(int64_t n, int n_parts)
{
    if (n < 0 || n_parts < 0) return 0;
    return pack_ws_ints(n, (int64_t)n_parts * 8) * sizeof(int);
}

extern "C" int escat_sell_pack_plan(const int32_t *rowptr, int64_t n, const int32_t *part_ptr, int n_parts, int max_part_rows,
                                    int values_per_entry, int layout, void *workspace, size_t workspace_bytes,
                                    escat_sell_dims *dims, void *stream)
{
    if (!rowptr || !part_ptr || !workspace || !dims) return fail(ESCAT_EINVAL, "NULL pointer");
    if (n <= 0 || n_parts <= 0) return fail(ESCAT_EINVAL, "empty operator");
    if (max_part_rows < 1 || max_part_rows > PACK_MAX_ROWS)
        return fail(ESCAT_ERANGE, "parts of up to %d rows: the register-resident pack takes at most %d", max_part_rows, PACK_MAX_ROWS);
    if (values_per_entry != 1 && values_per_entry != 3 && values_per_entry != 9)
        return fail(ESCAT_EINVAL, "values_per_entry=%d: 1, 3 or 9", values_per_entry);
    if (layout != 0 && layout != 1) return fail(ESCAT_EINVAL, "layout=%d: 0 or 1", layout);
    const int nsl = max_part_rows <= 256 ? 4 : 8;
    const int64_t ns = (int64_t)n_parts * nsl;
    if (ns >= (1 << 24)) return fail(ESCAT_ERANGE, "%lld slices", (long long)ns);
    if (workspace_bytes < pack_ws_ints(n, (int64_t)n_parts * 8) * sizeof(int))
        return fail(ESCAT_EWORKSPACE, "workspace %zu B < %zu B needed", workspace_bytes, pack_ws_ints(n, (int64_t)n_parts * 8) * sizeof(int));
    PackWs ws = pack_ws(workspace, n, ns);
    hipStream_t st = S(stream);
    hipError_t e = hipMemsetAsync(ws.hist, 0, 64 * sizeof(int), st);
    if (e != hipSuccess) return fail(ESCAT_ELAUNCH, "hipMemsetAsync: %s", hipGetErrorString(e));
    int64_t blocks = (n + 255) / 256; if (blocks > 512) blocks = 512;     // (every block ends with 34 atomics on the same words)
    hipLaunchKernelGGL(k_pack_hist, dim3((unsigned)blocks), dim3(256), 0, st, rowptr, n, ws.hist);
    hipLaunchKernelGGL(k_pack_pick_dmax, dim3(1), dim3(64), 0, st, ws.hist, n, values_per_entry);
    hipLaunchKernelGGL(k_pack_order, dim3(n_parts), dim3(PACK_MAX_ROWS), 0, st, rowptr, part_ptr, ws.hist + 40, nsl, ws.perm, ws.pos, ws.w);
    hipLaunchKernelGGL(k_pack_scan, dim3(1), dim3(1024), 0, st, ws.w, (int)ns, ws.doff, ws.poff);
    // one round trip: the histogram + the capacity picked on the device, and the two totals
    unsigned hist[41];
    int tot[2];
    e = hipMemcpyAsync(hist, ws.hist, sizeof(hist), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipMemcpyAsync(&tot[0], ws.doff + ns, sizeof(int), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipMemcpyAsync(&tot[1], ws.poff + ns, sizeof(int), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) return fail(ESCAT_ELAUNCH, "slice offsets: %s", hipGetErrorString(e));
    const int dmax = (int)hist[40];
    for (int i = 0; i < 34; ++i) dims->row_len_hist[i] = (int32_t)hist[i];
    CHECK_LAUNCH("escat_sell_pack_plan");
    dims->dmax = dmax; dims->n_slices = (int32_t)ns; dims->slices_per_part = nsl; dims->layout = layout;
    dims->total_d = tot[0]; dims->total_p = tot[1];
    const int per = values_per_entry == 9 ? 576 : values_per_entry == 3 ? 192 : 64;
    dims->val_floats = ((int64_t)tot[0] + 1) * per;           // + one padding slot (empty slices read it)
    dims->lc_words = ((int64_t)tot[1] + 1) * 64;              // + one padding pair
    return ESCAT_OK;
}

static int pack_fill(const int32_t *rowptr, const int32_t *col, const float *val, const int32_t *sign_bits,
                     int64_t n, int64_t nnz, const int32_t *part_ptr, int n_parts, int values_per_entry,
                     const escat_sell_dims *dims, void *workspace, size_t workspace_bytes,
                     float *val_out, uint32_t *lc_out, int32_t *lane_meta_out,
                     int32_t *slot_of_e, int32_t *lc_of_e, int8_t *ent_of_slot, void *stream, int plan);

extern "C" int escat_sell_pack_fill(const int32_t *rowptr, const int32_t *col, const float *val, const int32_t *sign_bits,
                                    int64_t n, int64_t nnz, const int32_t *part_ptr, int n_parts, int values_per_entry,
                                    const escat_sell_dims *dims, void *workspace, size_t workspace_bytes,
                                    float *val_out, uint32_t *lc_out, int32_t *lane_meta_out,
                                    int32_t *slot_of_e, int32_t *lc_of_e, int8_t *ent_of_slot, void *stream)
{
    return pack_fill(rowptr, col, val, sign_bits, n, nnz, part_ptr, n_parts, values_per_entry, dims, workspace, workspace_bytes,
                     val_out, lc_out, lane_meta_out, slot_of_e, lc_of_e, ent_of_slot, stream, 1);
}

extern "C" int escat_sell_pack_fill_unplanned(const int32_t *rowptr, const int32_t *col, const float *val, const int32_t *sign_bits,
                                              int64_t n, int64_t nnz, const int32_t *part_ptr, int n_parts, int values_per_entry,
                                              const escat_sell_dims *dims, void *workspace, size_t workspace_bytes,
                                              float *val_out, uint32_t *lc_out, int32_t *lane_meta_out,
                                              int32_t *slot_of_e, int32_t *lc_of_e, int8_t *ent_of_slot, void *stream)
{
    return pack_fill(rowptr, col, val, sign_bits, n, nnz, part_ptr, n_parts, values_per_entry, dims, workspace, workspace_bytes,
                     val_out, lc_out, lane_meta_out, slot_of_e, lc_of_e, ent_of_slot, stream, 0);
}

static int pack_fill(const int32_t *rowptr, const int32_t *col, const float *val, const int32_t *sign_bits,
                     int64_t n, int64_t nnz, const int32_t *part_ptr, int n_parts, int values_per_entry,
                     const escat_sell_dims *dims, void *workspace, size_t workspace_bytes,
                     float *val_out, uint32_t *lc_out, int32_t *lane_meta_out,
                     int32_t *slot_of_e, int32_t *lc_of_e, int8_t *ent_of_slot, void *stream, int plan)
{
    if (!rowptr || !col || !val || !part_ptr || !dims || !workspace || !val_out || !lc_out || !lane_meta_out)
        return fail(ESCAT_EINVAL, "NULL pointer");
    if ((slot_of_e != nullptr) != (lc_of_e != nullptr)) return fail(ESCAT_EINVAL, "slot_of_e and lc_of_e go together");
    if (((uintptr_t)val_out & 15) != 0) return fail(ESCAT_EINVAL, "val_out must be 16-byte aligned");
    const int64_t ns = dims->n_slices;
    if (ns != (int64_t)n_parts * dims->slices_per_part) return fail(ESCAT_EINVAL, "dims do not belong to this operator");
    if (workspace_bytes < pack_ws_ints(n, (int64_t)n_parts * 8) * sizeof(int)) return fail(ESCAT_EWORKSPACE, "workspace too small");
    if (sign_bits && values_per_entry != 3) return fail(ESCAT_EINVAL, "sign_bits go with values_per_entry = 3 (val = |value|, one per entry)");
    PackWs ws = pack_ws(workspace, n, ns);
    hipStream_t st = S(stream);
    const int per = values_per_entry == 9 ? 576 : (values_per_entry == 3 && !sign_bits) ? 192 : 64;
    // the padding slot / pair behind the last slice; slot_of_e starts at -1 (entries beyond dmax keep it)
    hipError_t e = hipMemsetAsync(val_out + dims->total_d * per, 0, (size_t)per * sizeof(float), st);
    if (e == hipSuccess) e = hipMemsetAsync(lc_out + dims->total_p * 64, 0, 64 * sizeof(uint32_t), st);
    if (e == hipSuccess && slot_of_e) e = hipMemsetAsync(slot_of_e, 0xff, (size_t)(nnz > 0 ? nnz : 1) * sizeof(int32_t), st);
    if (e != hipSuccess) return fail(ESCAT_ELAUNCH, "hipMemsetAsync: %s", hipGetErrorString(e));
#define GO(VV, SG, LY) hipLaunchKernelGGL((k_pack_fill<VV, SG, LY>), dim3((unsigned)ns), dim3(64), 0, st, rowptr, col, val, sign_bits, \
        part_ptr, ws.perm, ws.pos, ws.doff, ws.poff, dims->dmax, dims->slices_per_part, val_out, lc_out, lane_meta_out, slot_of_e, lc_of_e, \
        (signed char *)ent_of_slot, plan)
    const int ly = dims->layout;
    if (values_per_entry == 9) { if (ly) GO(9, false, 1); else GO(9, false, 0); }
    else if (values_per_entry == 1) { if (ly) GO(1, false, 1); else GO(1, false, 0); }
    else if (sign_bits) { if (ly) GO(3, true, 1); else GO(3, true, 0); }
    else { if (ly) GO(3, false, 1); else GO(3, false, 0); }
#undef GO
    CHECK_LAUNCH("escat_sell_pack_fill");
    return ESCAT_OK;
}

extern "C" int escat_sell_pack_refill(const int32_t *rowptr, const float *val, int64_t n, const int32_t *part_ptr, int n_parts,
                                      int values_per_entry, const escat_sell_dims *dims, const void *workspace,
                                      size_t workspace_bytes, const int8_t *ent_of_slot, float *val_out, void *stream)
{
    if (!rowptr || !val || !part_ptr || !dims || !workspace || !ent_of_slot || !val_out) return fail(ESCAT_EINVAL, "NULL pointer");
    if (((uintptr_t)val_out & 15) != 0) return fail(ESCAT_EINVAL, "val_out must be 16-byte aligned");
    const int64_t ns = dims->n_slices;
    if (ns != (int64_t)n_parts * dims->slices_per_part) return fail(ESCAT_EINVAL, "dims do not belong to this operator");
    if (workspace_bytes < pack_ws_ints(n, (int64_t)n_parts * 8) * sizeof(int)) return fail(ESCAT_EWORKSPACE, "workspace too small");
    PackWs ws = pack_ws(const_cast<void *>(workspace), n, ns);
    hipStream_t st = S(stream);
    const int per = values_per_entry == 9 ? 576 : values_per_entry == 3 ? 192 : 64;
    hipError_t e = hipMemsetAsync(val_out + dims->total_d * per, 0, (size_t)per * sizeof(float), st);
    if (e != hipSuccess) return fail(ESCAT_ELAUNCH, "hipMemsetAsync: %s", hipGetErrorString(e));
    const unsigned grid = (unsigned)((ns + 3) / 4);
#define GO(VV) hipLaunchKernelGGL((k_pack_refill<VV>), dim3(grid), dim3(256), 0, st, rowptr, val, part_ptr, ws.perm, ws.doff, \
                                  (const signed char *)ent_of_slot, dims->slices_per_part, (int)ns, val_out)
    if (values_per_entry == 9) GO(9); else if (values_per_entry == 3) GO(3); else if (values_per_entry == 1) GO(1);
    else return fail(ESCAT_EINVAL, "values_per_entry=%d: 1, 3 or 9", values_per_entry);
#undef GO
    CHECK_LAUNCH("escat_sell_pack_refill");
    return ESCAT_OK;
}

extern "C" int escat_sell_pack_refill_half(const int32_t *rowptr, const float *blk, int64_t n, const int32_t *part_ptr, int n_parts,
                                           const escat_sell_dims *dims, const void *workspace, size_t workspace_bytes,
                                           const int8_t *ent_of_slot, const uint32_t *lc, void *half_val_out,
                                           uint32_t *half_aux_out, int32_t *not_exact, void *stream)
{
    if (!rowptr || !blk || !part_ptr || !dims || !workspace || !ent_of_slot || !lc || !half_val_out || !half_aux_out || !not_exact)
        return fail(ESCAT_EINVAL, "NULL pointer");
    if (((uintptr_t)half_val_out & 15) != 0) return fail(ESCAT_EINVAL, "half_val_out must be 16-byte aligned");
    if (dims->layout != 1) return fail(ESCAT_EINVAL, "half-valued packs are layout 1");
    const int64_t ns = dims->n_slices;
    if (ns != (int64_t)n_parts * dims->slices_per_part) return fail(ESCAT_EINVAL, "dims do not belong to this operator");
    if (workspace_bytes < pack_ws_ints(n, (int64_t)n_parts * 8) * sizeof(int)) return fail(ESCAT_EWORKSPACE, "workspace too small");
    PackWs ws = pack_ws(const_cast<void *>(workspace), n, ns);
    hipStream_t st = S(stream);
    // the padding slot behind the last slice (empty slices read it)
    hipError_t e = hipMemsetAsync((char *)half_val_out + dims->total_d * 64 * 16, 0, 64 * 16, st);
    if (e == hipSuccess) e = hipMemsetAsync(half_aux_out + dims->total_d * 64, 0, 64 * sizeof(uint32_t), st);
    if (e != hipSuccess) return fail(ESCAT_ELAUNCH, "hipMemsetAsync: %s", hipGetErrorString(e));
    hipLaunchKernelGGL(k_pack_refill_half, dim3((unsigned)((ns + 3) / 4)), dim3(256), 0, st, rowptr, blk, part_ptr, ws.perm, ws.doff,
                       ws.poff, (const signed char *)ent_of_slot, lc, dims->slices_per_part, (int)ns, (uint4 *)half_val_out,
                       half_aux_out, not_exact);
    CHECK_LAUNCH("escat_sell_pack_refill_half");
    return ESCAT_OK;
}

extern "C" int escat_sell_pack_refill_signed(const int32_t *rowptr, const float *mag, const int32_t *sign_bits, int64_t n,
                                             const int32_t *part_ptr, int n_parts, const escat_sell_dims *dims,
                                             const void *workspace, size_t workspace_bytes, const int8_t *ent_of_slot,
                                             const uint32_t *lc, float *val_out, uint32_t *lc_out, void *stream)
{
    if (!rowptr || !mag || !sign_bits || !part_ptr || !dims || !workspace || !ent_of_slot || !lc || !val_out || !lc_out)
        return fail(ESCAT_EINVAL, "NULL pointer");
    const int64_t ns = dims->n_slices;
    if (ns != (int64_t)n_parts * dims->slices_per_part) return fail(ESCAT_EINVAL, "dims do not belong to this operator");
    if (workspace_bytes < pack_ws_ints(n, (int64_t)n_parts * 8) * sizeof(int)) return fail(ESCAT_EWORKSPACE, "workspace too small");
    PackWs ws = pack_ws(const_cast<void *>(workspace), n, ns);
    hipStream_t st = S(stream);
    hipError_t e = hipMemsetAsync(val_out + dims->total_d * 64, 0, 64 * sizeof(float), st);
    if (e == hipSuccess) e = hipMemsetAsync(lc_out + dims->total_p * 64, 0, 64 * sizeof(uint32_t), st);
    if (e != hipSuccess) return fail(ESCAT_ELAUNCH, "hipMemsetAsync: %s", hipGetErrorString(e));
    hipLaunchKernelGGL(k_pack_refill_signed, dim3((unsigned)((ns + 3) / 4)), dim3(256), 0, st, rowptr, mag, sign_bits, part_ptr, ws.perm,
                       ws.doff, ws.poff, (const signed char *)ent_of_slot, lc, dims->slices_per_part, (int)ns, val_out, lc_out);
    CHECK_LAUNCH("escat_sell_pack_refill_signed");
    return ESCAT_OK;
}

// ------------------------------------------------------------------ transpose on a symmetric pattern
template <int V>
__global__ void __launch_bounds__(256)
k_transpose_sym(const int *__restrict__ rowptr, const int *__restrict__ col, const float *__restrict__ val, int64_t n,
                float *__restrict__ val_t, int *__restrict__ not_symmetric)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    for (int e = rowptr[i]; e < rowptr[i + 1]; ++e) {
        const int j = col[e];
        if (j < 0 || j >= n) { atomicOr(not_symmetric, 1); continue; }      // not a valid operator: let the caller find out
        int lo = rowptr[j], hi = rowptr[j + 1] - 1, at = -1;
        while (lo <= hi) {                         // rows are sorted by column
            const int mid = (lo + hi) >> 1, c = col[mid];
            if (c == (int)i) { at = mid; break; }
            if (c < (int)i) lo = mid + 1; else hi = mid - 1;
        }
        if (at < 0) { atomicOr(not_symmetric, 1); continue; }
        if constexpr (V == 9) {
            float b[9];
            get_values<9>(b, val, e);
            f4u lo = {b[0], b[3], b[6], b[1]}, hi = {b[4], b[7], b[2], b[5]};       // the block, transposed, row-major
            *reinterpret_cast<f4u *>(val_t + (int64_t)at * 9) = lo;
            *reinterpret_cast<f4u *>(val_t + (int64_t)at * 9 + 4) = hi;
            val_t[(int64_t)at * 9 + 8] = b[8];
        } else {
#pragma unroll
            for (int q = 0; q < V; ++q) val_t[(int64_t)at * V + q] = val[(int64_t)e * V + q];
        }
    }
}

extern "C" int escat_transpose_sym(const int32_t *rowptr, const int32_t *col, const float *val, int64_t n, int64_t nnz,
                                   int values_per_entry, float *val_t, int32_t *not_symmetric, void *stream)
{
    if (!rowptr || !col || !val || !val_t || !not_symmetric) return fail(ESCAT_EINVAL, "NULL pointer");
    if (n <= 0 || nnz <= 0) return ESCAT_OK;
    const unsigned grid = (unsigned)((n + 255) / 256);
    if (values_per_entry == 1) hipLaunchKernelGGL((k_transpose_sym<1>), dim3(grid), dim3(256), 0, S(stream), rowptr, col, val, n, val_t, not_symmetric);
    else if (values_per_entry == 3) hipLaunchKernelGGL((k_transpose_sym<3>), dim3(grid), dim3(256), 0, S(stream), rowptr, col, val, n, val_t, not_symmetric);
    else if (values_per_entry == 9) hipLaunchKernelGGL((k_transpose_sym<9>), dim3(grid), dim3(256), 0, S(stream), rowptr, col, val, n, val_t, not_symmetric);
    else return fail(ESCAT_EINVAL, "values_per_entry=%d: 1, 3 or 9", values_per_entry);
    CHECK_LAUNCH("escat_transpose_sym");
    return ESCAT_OK;
}

// ------------------------------------------------------------------ diagonal blocks of a block-diagonal operator
// A cut between rows c and c+1 is legal iff no row "straddles" it, i.e. no row r has both c and c+1 inside
// [min(r, first column), max(r, last column)].  Every row adds +1 at its interval's start and -1 at its end; after a
// prefix sum (the caller's) the legal cuts are the zeros.  Rows are sorted by column (canonical CSR).
__global__ void __launch_bounds__(256)
k_row_cover(const int *__restrict__ rowptr, const int *__restrict__ col, int64_t n, int *__restrict__ delta)
{
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= n) return;
    const int e0 = rowptr[r], e1 = rowptr[r + 1];
    int lo = (int)r, hi = (int)r;
    if (e1 > e0) { const int a = col[e0], b = col[e1 - 1]; lo = a < lo ? a : lo; hi = b > hi ? b : hi; }
    if (lo < 0) lo = 0;                            // (columns outside [0, n): clamped here, rejected by the operator's validation)
    if (hi >= n) hi = (int)n - 1;
    if (hi > lo) { atomicAdd(&delta[lo], 1); atomicAdd(&delta[hi], -1); }
}

extern "C" int escat_row_cover(const int32_t *rowptr, const int32_t *col, int64_t n, int32_t *delta, void *stream)
{
    if (!rowptr || !delta) return fail(ESCAT_EINVAL, "NULL pointer");
    if (n <= 0) return ESCAT_OK;
    hipLaunchKernelGGL(k_row_cover, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, S(stream), rowptr, col, n, delta);
    CHECK_LAUNCH("escat_row_cover");
    return ESCAT_OK;
}

// host utility: greedy merge of consecutive diagonal blocks (bounds[0] = 0 < ... < bounds[nb-1] = n, HOST pointers) into
// parts of at most `target` rows (a block larger than the target stays alone); returns the number of part boundaries
// written to out (out[0] = 0, last = n), or a negative error code.
extern "C" int64_t escat_merge_row_ranges(const int64_t *bounds, int64_t nb, int64_t target, int32_t *out, int32_t *max_rows)
{
    if (!bounds || !out || !max_rows || nb < 2) return fail(ESCAT_EINVAL, "bad bounds");
    int64_t k = 0, start = bounds[0], mx = 0;
    out[k++] = (int32_t)start;
    for (int64_t i = 1; i < nb; ++i)
        if (bounds[i] - start > target && bounds[i - 1] > start) {
            if (bounds[i - 1] - start > mx) mx = bounds[i - 1] - start;
            start = bounds[i - 1];
            out[k++] = (int32_t)start;
        }
    if (bounds[nb - 1] - start > mx) mx = bounds[nb - 1] - start;
    out[k++] = (int32_t)bounds[nb - 1];
    *max_rows = (int32_t)mx;
    return k;
}

// ------------------------------------------------------------------ COO ingest of the reference's two layouts
// What the reference's pipeline hands over every step (models/escgnn_data_classes.py:137-142 + PyG collate): int64 COO
// flagged un-coalesced, P already row-major sorted (process_pyg_data.py:775), Q in block-emission order (:839-857: the
// nine entries of one 3x3 block in a row, blocks in P's order).  One pass checks the layout and writes the canonical
// int32 structure; *bad is set when the tensor is anything else (the caller then coalesces the general way).
__device__ __forceinline__ void fill_rowptr(int *__restrict__ rowptr, long long prev_row, long long row, long long e)
{
    for (long long q = prev_row + 1; q <= row; ++q) rowptr[q] = (int)e;
}

// delta (optional, n + 1 ints, zeroed): the row-cover marks of k_row_cover, made in the same pass -- a row's first entry
// adds +1 at min(row, its column), its last entry -1 at max(row, its column) (a span of one row cancels itself)
__global__ void __launch_bounds__(256)
k_coo_sorted_to_csr(const long long *__restrict__ rows, const long long *__restrict__ cols, int64_t nnz, int64_t n,
                    int *__restrict__ rowptr, int *__restrict__ col32, int *__restrict__ bad, int *__restrict__ delta)
{
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= nnz) return;
    const long long r = rows[e], c = cols[e];
    if (r < 0 || r >= n || c < 0 || c >= n) { atomicOr(bad, 1); return; }
    const long long pr = e ? rows[e - 1] : -1, pc = e ? cols[e - 1] : -1;
    if (e && !(r > pr || (r == pr && c > pc))) { atomicOr(bad, 1); return; }
    if (e && (pr < 0 || pr >= n || pc < 0 || pc >= n)) return;     // entry e-1 flags itself; never index from its row
    col32[e] = (int)c;
    if (r != pr) {
        fill_rowptr(rowptr, pr, r, e);
        if (delta) {
            atomicAdd(&delta[r < c ? r : c], 1);
            if (e) atomicAdd(&delta[pr > pc ? pr : pc], -1);
        }
    }
    if (e == nnz - 1) {
        fill_rowptr(rowptr, r, n, nnz);
        if (delta) atomicAdd(&delta[r > c ? r : c], -1);
    }
}

// 63 lanes of a wave cover 7 blocks (one entry per lane: every load is one coalesced 504-byte access); the block's
// first entry and the previous block's reach the other lanes by shuffles.
__global__ void __launch_bounds__(256)
k_coo_blocks_to_bsr3(const long long *__restrict__ rows, const long long *__restrict__ cols, int64_t nnzb, int64_t nb,
                     int *__restrict__ rowptr, int *__restrict__ col32, int *__restrict__ bad)
{
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6;
    const int q = lane % 9, k = lane / 9;
    const int64_t b = wave * 7 + k;
    const bool live = lane < 63 && b < nnzb;
    long long r = -1, c = -1;
    if (live) { r = rows[9 * b + q]; c = cols[9 * b + q]; }
    const long long r0 = __shfl(r, lane - q, 64), c0 = __shfl(c, lane - q, 64);
    long long pr0 = __shfl(r, lane >= 9 ? lane - 9 : 0, 64), pc0 = __shfl(c, lane >= 9 ? lane - 9 : 0, 64);
    if (!live) return;
    if (r != r0 + q / 3 || c != c0 + q % 3) { atomicOr(bad, 1); return; }
    if (q) return;
    if (r0 < 0 || c0 < 0 || r0 % 3 || c0 % 3 || r0 / 3 >= nb || c0 / 3 >= nb) { atomicOr(bad, 1); return; }
    const long long br = r0 / 3, bc = c0 / 3;
    long long pbr = -1, pbc = -1;
    if (b) {
        if (!k) { pr0 = rows[9 * (b - 1)]; pc0 = cols[9 * (b - 1)]; }
        pbr = pr0 / 3; pbc = pc0 / 3;
        if (pr0 < 0 || !(br > pbr || (br == pbr && bc > pbc))) { atomicOr(bad, 1); return; }
    }
    col32[b] = (int)bc;
    if (br != pbr) fill_rowptr(rowptr, pbr, br, b);
    if (b == nnzb - 1) fill_rowptr(rowptr, br, nb, nnzb);
}

extern "C" int escat_ingest_coo(const int64_t *rows, const int64_t *cols, int64_t nnz, int64_t n, int blocks3,
                                int32_t *rowptr, int32_t *col32, int32_t *bad, void *stream)
{
    if (!rows || !cols || !rowptr || !col32 || !bad) return fail(ESCAT_EINVAL, "NULL pointer");
    if (n <= 0 || nnz <= 0) return fail(ESCAT_EINVAL, "empty operator");
    if (n >= ((int64_t)1 << 31) || nnz >= ((int64_t)1 << 31)) return fail(ESCAT_ERANGE, "n or nnz >= 2^31 (int32 indices)");
    if (blocks3) {
        if (n % 3 || nnz % 9) return fail(ESCAT_EINVAL, "a BSR-3 ingest needs n %% 3 == 0 and nnz %% 9 == 0");
        const int64_t nnzb = nnz / 9;
        hipLaunchKernelGGL(k_coo_blocks_to_bsr3, dim3((unsigned)((nnzb + 27) / 28)), dim3(256), 0, S(stream),
                           (const long long *)rows, (const long long *)cols, nnzb, n / 3, rowptr, col32, bad);
    } else {
        hipLaunchKernelGGL(k_coo_sorted_to_csr, dim3((unsigned)((nnz + 255) / 256)), dim3(256), 0, S(stream),
                           (const long long *)rows, (const long long *)cols, nnz, n, rowptr, col32, bad, (int *)nullptr);
    }
    CHECK_LAUNCH("escat_ingest_coo");
    return ESCAT_OK;
}

extern "C" int escat_ingest_coo_cover(const int64_t *rows, const int64_t *cols, int64_t nnz, int64_t n,
                                      int32_t *rowptr, int32_t *col32, int32_t *bad, int32_t *delta, void *stream)
{
    if (!rows || !cols || !rowptr || !col32 || !bad || !delta) return fail(ESCAT_EINVAL, "NULL pointer");
    if (n <= 0 || nnz <= 0) return fail(ESCAT_EINVAL, "empty operator");
    if (n >= ((int64_t)1 << 31) || nnz >= ((int64_t)1 << 31)) return fail(ESCAT_ERANGE, "n or nnz >= 2^31 (int32 indices)");
    hipError_t e = hipMemsetAsync(delta, 0, (size_t)(n + 1) * sizeof(int32_t), S(stream));
    if (e != hipSuccess) return fail(ESCAT_ELAUNCH, "hipMemsetAsync(delta): %s", hipGetErrorString(e));
    hipLaunchKernelGGL(k_coo_sorted_to_csr, dim3((unsigned)((nnz + 255) / 256)), dim3(256), 0, S(stream),
                       (const long long *)rows, (const long long *)cols, nnz, n, rowptr, col32, bad, delta);
    CHECK_LAUNCH("escat_ingest_coo_cover");
    return ESCAT_OK;
}

// Q of a batch whose P went through the ingest a moment ago: the blocks must be in block-emission order AND sit exactly on
// the reference pattern (block b = entry b of the reference CSR).  Nothing is written but the flags: bit 0 = not in
// block-emission order (as *bad above), bit 1 = in order but on another pattern.  No block depends on its neighbour, so
// a wave keeps U trips of 7 blocks (two 8-byte loads per lane each) in flight.
template <int U>
__global__ void __launch_bounds__(256)
k_coo_blocks_like(const long long *__restrict__ rows, const long long *__restrict__ cols, int64_t nnzb, int64_t nb,
                  const int *__restrict__ ref_rowptr, const int *__restrict__ ref_col, int *__restrict__ flags)
{
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6;
    const int q = lane % 9, k = lane / 9;
    long long r[U], c[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int64_t b = (wave * U + u) * 7 + k;
        const bool live = lane < 63 && b < nnzb;
        r[u] = -1; c[u] = -1;
        if (live) { r[u] = __builtin_nontemporal_load(rows + 9 * b + q); c[u] = __builtin_nontemporal_load(cols + 9 * b + q); }
    }
    int f = 0;
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int64_t b = (wave * U + u) * 7 + k;
        const bool live = lane < 63 && b < nnzb;
        const long long r0 = __shfl(r[u], lane - q, 64), c0 = __shfl(c[u], lane - q, 64);
        if (!live) continue;
        if (r[u] != r0 + q / 3 || c[u] != c0 + q % 3) { f |= 1; continue; }
        if (q) continue;
        if (r0 < 0 || c0 < 0 || r0 % 3 || c0 % 3 || r0 / 3 >= nb || c0 / 3 >= nb) { f |= 1; continue; }
        const long long br = r0 / 3, bc = c0 / 3;
        if (ref_col[b] != (int)bc || !((int64_t)ref_rowptr[br] <= b && b < (int64_t)ref_rowptr[br + 1])) f |= 2;
    }
    if (f) atomicOr(flags, f);
}

extern "C" int escat_ingest_coo_like(const int64_t *rows, const int64_t *cols, int64_t nnz, int64_t n,
                                     const int32_t *ref_rowptr, const int32_t *ref_col, int64_t ref_nnz, int32_t *flags, void *stream)
{
    if (!rows || !cols || !ref_rowptr || !ref_col || !flags) return fail(ESCAT_EINVAL, "NULL pointer");
    if (n <= 0 || nnz <= 0 || n % 3 || nnz % 9) return fail(ESCAT_EINVAL, "a BSR-3 ingest needs n %% 3 == 0 and nnz %% 9 == 0 (both > 0)");
    if (n >= ((int64_t)1 << 31) || nnz >= ((int64_t)1 << 31)) return fail(ESCAT_ERANGE, "n or nnz >= 2^31 (int32 indices)");
    if (ref_nnz != nnz / 9) return fail(ESCAT_EINVAL, "the reference pattern has %lld entries, the operator %lld blocks", (long long)ref_nnz, (long long)(nnz / 9));
    constexpr int U = 8;
    const int64_t nnzb = nnz / 9, waves = (nnzb + 7 * U - 1) / (7 * U);
    hipLaunchKernelGGL(k_coo_blocks_like<U>, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, S(stream),
                       (const long long *)rows, (const long long *)cols, nnzb, n / 3, ref_rowptr, ref_col, flags);
    CHECK_LAUNCH("escat_ingest_coo_like");
    return ESCAT_OK;
}

// ------------------------------------------------------------------ the closed row ranges from the cover marks, in one call
// delta: the +1 / -1 marks (k_row_cover, or k_coo_sorted_to_csr's delta).  Three small kernels -- block sums, the scan with
// the cuts of every block compacted in place, the cuts of all blocks compacted -- then the (few thousand) cuts come to
// the host, where the greedy merge runs (escat_merge_row_ranges).  delta is overwritten.
#define COVER_ROWS 2048           // rows per workgroup: 256 threads x 8

__device__ __forceinline__ int block_exclusive_scan_256(int v, int *sh, int *total)
{
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    int x = v;
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) { const int o = __shfl_up(x, m, 64); if (lane >= m) x += o; }
    if (lane == 63) sh[w] = x;
    __syncthreads();
    int base = 0;
    for (int i = 0; i < w; ++i) base += sh[i];
    *total = sh[0] + sh[1] + sh[2] + sh[3];
    __syncthreads();
    return base + x - v;
}

__global__ void __launch_bounds__(256)
k_cover_sums(const int *__restrict__ delta, int64_t n, int *__restrict__ bsum)
{
    __shared__ int sh[4];
    const int64_t r0 = (int64_t)blockIdx.x * COVER_ROWS + threadIdx.x * 8;
    int s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) if (r0 + i < n) s += delta[r0 + i];
    int tot;
    (void)block_exclusive_scan_256(s, sh, &tot);
    if (threadIdx.x == 0) bsum[blockIdx.x] = tot;
}

__global__ void __launch_bounds__(256)
k_cover_cuts(int *__restrict__ delta, int64_t n, const int *__restrict__ bsum, int *__restrict__ bcnt)
{
    __shared__ int sh[4];
    int carry = 0;
    for (int b = threadIdx.x; b < (int)blockIdx.x; b += 256) carry += bsum[b];
    int tot;
    (void)block_exclusive_scan_256(carry, sh, &tot);
    carry = tot;                                   // marks of all rows before this block
    const int64_t r0 = (int64_t)blockIdx.x * COVER_ROWS + threadIdx.x * 8;
    int d[8], s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) { d[i] = r0 + i < n ? delta[r0 + i] : 0; s += d[i]; }
    int run = carry + block_exclusive_scan_256(s, sh, &tot);
    int cut[8], nc = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) { run += d[i]; cut[i] = (r0 + i < n - 1 && run == 0) ? 1 : 0; nc += cut[i]; }
    int pos = block_exclusive_scan_256(nc, sh, &tot);      // (its barriers also order the loads above before the stores below)
    int *list = delta + (int64_t)blockIdx.x * COVER_ROWS;
#pragma unroll
    for (int i = 0; i < 8; ++i) if (cut[i]) list[pos++] = (int)(r0 + i + 1);
    if (threadIdx.x == 0) bcnt[blockIdx.x] = tot;
}

__global__ void __launch_bounds__(256)
k_cover_compact(const int *__restrict__ delta, const int *__restrict__ bcnt, int nblk, int *__restrict__ cuts, int *__restrict__ total,
                const int *__restrict__ flag_in)
{
    __shared__ int sh[4];
    int off = 0;
    for (int b = threadIdx.x; b < (int)blockIdx.x; b += 256) off += bcnt[b];
    int tot;
    (void)block_exclusive_scan_256(off, sh, &tot);
    const int cnt = bcnt[blockIdx.x];
    const int *list = delta + (int64_t)blockIdx.x * COVER_ROWS;
    for (int i = threadIdx.x; i < cnt; i += 256) cuts[tot + i] = list[i];
    if ((int)blockIdx.x == nblk - 1 && threadIdx.x == 0) { total[0] = tot + cnt; total[1] = flag_in ? *flag_in : 0; }
}

extern "C" size_t escat_row_blocks_workspace_bytes(int64_t n)
{
    const int64_t nblk = (n + COVER_ROWS - 1) / COVER_ROWS;
    return (size_t)(2 * nblk + 4 + n) * sizeof(int32_t);
}

extern "C" int64_t escat_row_blocks(int32_t *delta, int64_t n, int64_t target, void *workspace, size_t workspace_bytes,
                                    int32_t *out_host, int64_t out_cap, int32_t *max_rows, const int32_t *flag_dev, int32_t *flag_host,
                                    void *stream)
{
    if (!delta || !workspace || !out_host || !max_rows) return fail(ESCAT_EINVAL, "NULL pointer");
    if (n <= 0 || n >= ((int64_t)1 << 31)) return fail(ESCAT_EINVAL, "n = %lld", (long long)n);
    if (workspace_bytes < escat_row_blocks_workspace_bytes(n)) return fail(ESCAT_EWORKSPACE, "escat_row_blocks: workspace too small");
    const int nblk = (int)((n + COVER_ROWS - 1) / COVER_ROWS);
    int *bsum = (int *)workspace, *bcnt = bsum + nblk, *total = bcnt + nblk, *cuts = total + 4;
    hipLaunchKernelGGL(k_cover_sums, dim3(nblk), dim3(256), 0, S(stream), delta, n, bsum);
    hipLaunchKernelGGL(k_cover_cuts, dim3(nblk), dim3(256), 0, S(stream), delta, n, bsum, bcnt);
    hipLaunchKernelGGL(k_cover_compact, dim3(nblk), dim3(256), 0, S(stream), delta, bcnt, nblk, cuts, total, flag_dev);
    CHECK_LAUNCH("escat_row_blocks");
    // count and (optimistically) the first cuts in one copy; the rest only if there are more
    static thread_local std::vector<int64_t> bounds;
    static thread_local std::vector<int32_t> stage;
    const int64_t first = 1 << 14;
    stage.resize((size_t)(4 + first));
    const int64_t have = n < first ? n : first;
    hipError_t e = hipMemcpyAsync(stage.data(), total, (size_t)(4 + have) * sizeof(int32_t), hipMemcpyDeviceToHost, S(stream));
    if (e == hipSuccess) e = hipStreamSynchronize(S(stream));
    if (e != hipSuccess) return fail(ESCAT_ELAUNCH, "escat_row_blocks: %s", hipGetErrorString(e));
    const int64_t nc = stage[0];
    if (flag_host) *flag_host = stage[1];
    if (flag_dev && stage[1]) return 0;           // the marks are those of a tensor the ingest refused: nothing to merge
    if (nc < 0 || nc >= n) return fail(ESCAT_ELAUNCH, "escat_row_blocks: %lld cuts for %lld rows", (long long)nc, (long long)n);
    if (nc > have) {
        stage.resize((size_t)(4 + nc));
        e = hipMemcpyAsync(stage.data() + 4 + have, cuts + have, (size_t)(nc - have) * sizeof(int32_t), hipMemcpyDeviceToHost, S(stream));
        if (e == hipSuccess) e = hipStreamSynchronize(S(stream));
        if (e != hipSuccess) return fail(ESCAT_ELAUNCH, "escat_row_blocks: %s", hipGetErrorString(e));
    }
    if (nc + 2 > out_cap) return fail(ESCAT_ERANGE, "escat_row_blocks: %lld boundaries, room for %lld", (long long)(nc + 2), (long long)out_cap);
    bounds.resize((size_t)(nc + 2));
    bounds[0] = 0;
    for (int64_t i = 0; i < nc; ++i) bounds[(size_t)i + 1] = stage[(size_t)(4 + i)];
    bounds[(size_t)nc + 1] = n;
    return escat_merge_row_ranges(bounds.data(), nc + 2, target, out_host, max_rows);
}
