/*
 * escat.h -- C ABI of libescat.so: MI355X (gfx950) kernels for ESc-GNN's
 * diffusion-wavelet scattering hot path.
 *
 * The reference (dj408/esc-gnn) has no FFI layer: its boundary for this path is
 * the Python function pyg_utilities.get_Batch_Wjxs (pyg_utilities.py:283-560)
 * and, one level up, ESCGNNModular._scatter (models/escgnn_modular.py:309-362).
 * Each entry point below names the reference lines whose work it replaces; the
 * ctypes binding a maintainer would add is shown in INTEGRATION.md and is what
 * esc_gnn_amd/_lib.py does.
 *
 * Conventions (all entry points):
 *   - plain pointers and sizes only; every array pointer is a DEVICE pointer
 *     unless the parameter comment says "host";
 *   - returns 0 on success, a negative ESCAT_E* code otherwise; the message is
 *     available from escat_last_error() (thread-local); no C++ exception
 *     crosses the boundary;
 *   - asynchronous on `stream` (a hipStream_t passed as void*; NULL = the
 *     default stream), no internal synchronisation, no hidden allocation:
 *     scratch memory comes in through the workspace arguments;
 *   - int32 indices (nnz < 2^31 is checked), fp32 values, row-major dense
 *     operands.
 *
 * Operator formats
 *   CSR   rowptr[n+1], col[nnz], val[nnz]; rows sorted by column (the order a
 *         coalesced torch COO has).
 *   BSR-3 rowptr[nb+1], col[nnzb], blk[nnzb][3][3]: block (i,j) covers scalar
 *         rows 3i..3i+2 / cols 3j..3j+2; this is the vector-track operator Q
 *         (process_pyg_data.py:823-868 emits exactly one 3x3 block per P-nnz).
 *
 * Event table (how every mode of get_Batch_Wjxs is expressed)
 *   The chain visits y_t = A^t x for t = 0..T.  For every event
 *   {step, slot, sign, chan}: OUT[:, c, slot] (+|-)= y_step[:, c] for c = chan
 *   (or all channels when chan < 0).  A wavelet band Ptxs[f] - Ptxs[f+1]
 *   (pyg_utilities.py:381) is a + event at the first power and a - event at
 *   the second; the low-pass (:383) is a single + event.  Events must be
 *   sorted by step; every (slot, channel) of OUT must receive at least one
 *   event (OUT is not cleared).
 */
#ifndef ESCAT_H
#define ESCAT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ESCAT_OK            0
#define ESCAT_EINVAL       -1   /* bad argument (null pointer, size, unsorted events, ...) */
#define ESCAT_ERANGE       -2   /* size beyond what the kernels support */
#define ESCAT_EWORKSPACE   -3   /* workspace too small */
#define ESCAT_ELAUNCH      -4   /* HIP reported a launch error */

#define ESCAT_MAX_EVENTS   160
#define ESCAT_MAX_CHANNELS 8    /* per launch; the host side chunks wider X */
#define ESCAT_MAX_SLOTS    160  /* F: output slots of a chain (the all-powers plan of infogain uses T+1) */

typedef struct { int32_t step, slot, sign, chan; } escat_event;
/* Optional |.| epilogue (forward entry points only; OFF in the drop-in path: ESCGNNModular._scatter keeps the signed
 * coefficients, models/escgnn_modular.py:326-341 -- the modulus exists only in LEGS-style callers,
 * models/comparisons/legs.py:299-302).  Adding ESCAT_EVENT_ABS to the `sign` of the LAST event of a slot
 * (sign = +1 + ESCAT_EVENT_ABS or -1 + ESCAT_EVENT_ABS) makes the chain keep |value| in that slot once the event
 * has been applied, in the same pass; the backward entry points reject it. */
#define ESCAT_EVENT_ABS    256

/* Packed "SELL-64" copy of a CSR / BSR-3 operator for the register-resident
 * chain kernels (all members are device pointers; the struct itself lives on
 * the host).  Every part owns slices_per_part slices (4 if no part has more than
 * 256 rows, else 8; trailing slices of a part may be empty).  Inside each part
 * rows are sorted by length, descending, and cut into slices of 64 rows (one
 * wave); slice s keeps the first w_s = min(longest row of the slice, dmax)
 * entries of each row, slot-major:
 *   CSR   val[(doff_s + d)*64 + lane]
 *   BSR-3 val[(doff_s + d)*576 + {lane*4+0..3 | 256+lane*4+0..3 | 512+lane}]
 *         = block entries b00 b10 b01 b11 | b02 b12 b20 b21 | b22  (bij = row i, column j; pairs for packed fp32 FMAs)
 *   lc[(poff_s + d/2)*64 + lane] = LDS row of slot d | LDS row of slot d+1 << 16, where the LDS row of an entry
 *         with local column a (= column - first row of the part) is a (copy A of the vector) or R + copyB(a)
 *         (copy B; R = rows of the part + 1 rounded up to 16, copyB rotates bits 0-3 of a to the top of its
 *         256-row block), as chosen by escat_sell_plan_slots; slots past a row's end hold `rows` (the zero row)
 *         (the pairs of a slice cover w_s rounded up to a multiple of 4 slots)
 * with doff_s, poff_s = prefix sums of w_s and 2*ceil(w_s/4) over all slices, and
 *   lane_meta[(s*64 + lane)*3 + 0] = local row served by the lane (0x3ff: none) | tail_n << 10
 *   lane_meta[(s*64 + lane)*3 + 1] = tail_e0: the row's entries [tail_e0, tail_e0 + tail_n) of the CSR / BSR-3
 *                                    arrays lie beyond dmax and are streamed every step
 *   lane_meta[(s*64 + lane)*3 + 2] = slice word: lanes 0..4 hold doff_s, poff_s, w_s, first row of the part,
 *                                    rows of the part (one 12-byte load per lane starts a part)
 * val and lc end with one padding slot / pair (the kernels read it for empty slices).
 * dmax: 8/16/20/24/32 (CSR), 8/12/16/20/22 (BSR-3).
 * esc_gnn_amd/operators.py builds it (torch index ops + escat_sell_plan_slots, once per operator). */
typedef struct {
    const float    *val;
    const uint32_t *lc;
    const int32_t  *lane_meta;
    int32_t         dmax;
    int32_t         n_slices;
    int32_t         slices_per_part;
    int32_t         val_planes;     /* CSR packs: 0/1 = one value per entry; 3 = one value per channel (C = 3):
                                     * val[((doff_s + d)*3 + c)*64 + lane], and the entry points' `val` is (nnz, 3);
                                     * 4 = the same operator when the three values differ in sign only: one plane of
                                     * |value| as for 1, the signs of channels 0..2 in bits 11..13 of each 16-bit LDS-row
                                     * field of lc, and -- in the entry points' global arrays -- `val` (nnz,) = |value|,
                                     * `col` = column | signs << 29.  A struct with val = NULL and val_planes = 3 or 4
                                     * only describes the global arrays (multi-launch path, no pack). */
    /* whole-operator variant for the multi-launch path of graphs that do not fit LDS (BSR-3 only), selected by
     * col32 != NULL: slices of 64 rows sorted by length inside windows, no dmax cap; val as above,
     * col32[(doff_s + d)*64 + lane] = GLOBAL block column (-1 = padding), perm[s*64 + lane] = global row (-1: none),
     * slice_doff[s] = doff_s (n_slices + 1 entries); lc / lane_meta unused. */
    const int32_t  *col32;
    const int32_t  *perm;
    const int32_t  *slice_doff;
    /* optional batch view (register-resident path only): the launch covers the n_list parts part_list[i] of the packed
     * operator, in that order; the rows of part part_list[i] in the dense operands (X / OUT / G / dX) start at
     * part_row0[i] (the dense operands then have sum-of-listed-rows rows, not the operator's n).  This is how a
     * shuffled mini-batch is drawn from a whole dataset's operators kept packed in HBM, with no copy and no re-ingest.
     * NULL / NULL / 0: all parts, rows as in the operator. */
    const int32_t  *part_list;
    const int32_t  *part_row0;
    int32_t         n_list;
    /* optional, forward entry points on the register-resident path only: fuse the orders 0/1/2 of
     * ESCGNNModular._scatter (models/escgnn_modular.py:333-341) into the chain's epilogue.  0: OUT = the bank (n, C, F);
     * 1: OUT (n, C, 1+F) = [x | bank]; 2: OUT (n, C, 1+F+F(F-1)/2) = [x | bank | bank_i*bank_j, i<j in triu order]
     * -- what escat_scatter_orders_f32 would make of the bank, without the bank's round trip through HBM.
     * Must be 0 for the backward entry points. */
    int32_t         orders;
    /* val_planes = 3 only, optional: local frames O (n, 3, 3) of the operator's rows.  The chain then computes
     * D Pt^t D^T (the vector operator in factored form, D = blockdiag(O)): inputs are rotated into the frames and results
     * back out inside the kernel, in LDS.  NULL: the chain on Pt itself. */
    const float    *frames;
    /* how LDS rows are numbered (ABI version 2).  0: an LDS row is the part's local row index, copy B rotates bits 0-3 of
     * it to the top of its 256-row block (the description above; per-channel packs, val_planes = 3 / 4, use it).
     * 1: an LDS row is a POSITION = the lane index, inside the part, of the thread that serves the row (rows sorted by
     * length), copy B = R + ((p & ~15) | ((p + 1 + (p >> 5) % 15) & 15)): the owner stores its new vector row at its own
     * thread index, so both copies' 16-byte stores are bank-conflict-free.  Scalar (val_planes 0/1) and BSR-3 packs
     * must be layout 1 and bring pos_of_row[global row] = position (the streamed tails of rows longer than dmax need
     * it).  escat_sell_pack_plan / _fill build either layout. */
    int32_t         layout;
    const int32_t  *pos_of_row;
    /* ABI version 3, BSR-3 packs only, optional (NULL / NULL: none): the same pack with the block values held as IEEE
     * halves, for an operator whose every value v satisfies float(half(v)) == v -- what the reference's ellipsoid pipeline
     * stores (process_pyg_data.py:676 hdf5_tensor_dtype='float16'; widened at load, models/escgnn_data_classes.py:137-142).
     *   half_val  uint32[(doff_s + d)*64 + lane][4] = half pairs, low | high << 16: (b00 b10) (b01 b11) (b02 b12) (b20 b21)
     *   half_aux  uint32[(doff_s + d)*64 + lane]    = b22 (half) | (16 * LDS row of the slot) << 16
     * (one padding slot behind the last slice, as for val).  escat_sell_pack_refill_half builds both and reports whether
     * the values qualify.  The chain entry points then run a kernel that keeps a block in 5 registers instead of 9.5 (two
     * workgroups per CU instead of one) and computes, in fp32, bit for bit what the fp32 pack gives.  dmax <= 20; val may be
     * NULL when half_val is given (lc is not read either, but must be non-NULL). */
    const void     *half_val;
    const uint32_t *half_aux;
    /* ABI version 4, BSR-3 only, optional: val == NULL, half_val == NULL and ent_of_slot != NULL = "direct" -- no value pack at
     * all: the chain reads every row's blocks from the canonical blk array it is given anyway (entry rowptr[row] +
     * ent_of_slot[(doff_s + d) * 64 + lane] sits in slot d; -1 = a padded slot), following the plan of a pack with the
     * same pattern (P of the batch: lc / lane_meta / pos_of_row are that pack's).  For an operator that is used once:
     * packing its 36 B per block costs more than the strided reads of one chain. */
    const int8_t *ent_of_slot;
} escat_sell;

/* ---- the pack itself, built on the GPU (replaces round 1's torch index ops; esc_gnn_amd/operators.py::Sell calls
 * these).  Two phases because the pack's size depends on the data:
 *   escat_sell_pack_plan   picks dmax from the row-length histogram, orders every part's rows by length (stable,
 *                          descending), sizes the slices and prefix-sums their offsets into `workspace`; reads two
 *                          small results back (it synchronises `stream`) and fills *dims (host);
 *   escat_sell_pack_fill   with val_out (dims->val_floats floats, 16-byte aligned), lc_out (dims->lc_words words) and
 *                          lane_meta_out (n_slices*64*3 ints) allocated by the caller: one wave per slice plans slots and
 *                          vector copies (layout 0: greedily, as escat_sell_plan_slots; layout 1: by a maximum matching
 *                          of each ds_read_b128 lane group's 16 lanes onto the 16 bank slots) and writes the slice.  The same workspace must be
 *                          passed to both; afterwards it holds (int32) hist[64] | w[ns] | doff[ns+1] | poff[ns+1] |
 *                          perm[ns*64] | pos_of_row[n] with ns = n_slices, of which pos_of_row is what escat_sell wants.
 * values_per_entry: 1 (CSR), 9 (BSR-3 blocks, row-major), 3 (one value per channel); with sign_bits != NULL
 * (values_per_entry 3) `val` is (nnz,) magnitudes and sign_bits[e] bits 0..2 the channels' signs: the val_planes = 4
 * format.  slot_of_e / lc_of_e (nnz ints each, optional, both or neither): the plan per entry, for diagnostics. */
typedef struct {
    int32_t dmax, n_slices, slices_per_part, layout;
    int64_t total_d, total_p;           /* slots / LDS-row pairs over all slices */
    int64_t val_floats, lc_words;       /* elements of val / lc to allocate (padding slot / pair included) */
    int32_t row_len_hist[34];           /* rows by length (33 = longer): what dmax was chosen from, for callers that decide
                                         * whether a pack of another values_per_entry can share this plan (round 3) */
} escat_sell_dims;
size_t escat_sell_pack_workspace_bytes(int64_t n, int n_parts);
int escat_sell_pack_plan(const int32_t *rowptr, int64_t n, const int32_t *part_ptr, int n_parts, int max_part_rows,
                         int values_per_entry, int layout, void *workspace, size_t workspace_bytes,
                         escat_sell_dims *dims, void *stream);
int escat_sell_pack_fill(const int32_t *rowptr, const int32_t *col, const float *val, const int32_t *sign_bits,
                         int64_t n, int64_t nnz, const int32_t *part_ptr, int n_parts, int values_per_entry,
                         const escat_sell_dims *dims, void *workspace, size_t workspace_bytes,
                         float *val_out, uint32_t *lc_out, int32_t *lane_meta_out,
                         int32_t *slot_of_e, int32_t *lc_of_e, int8_t *ent_of_slot, void *stream);
/* The same without the slot / copy planning: entries in row order, copy A of the vector.  The chain then pays the LDS bank
 * conflicts the plan would have avoided (measured at cfg2: P forward 0.25 instead of 0.20 ms) but the pack costs a
 * tenth: for an operator that is used once (a DataLoader that yields fresh COO tensors every step,
 * models/escgnn_data_classes.py:137-142), where the matching's milliseconds never pay back. */
int escat_sell_pack_fill_unplanned(const int32_t *rowptr, const int32_t *col, const float *val, const int32_t *sign_bits,
                                   int64_t n, int64_t nnz, const int32_t *part_ptr, int n_parts, int values_per_entry,
                                   const escat_sell_dims *dims, void *workspace, size_t workspace_bytes,
                                   float *val_out, uint32_t *lc_out, int32_t *lane_meta_out,
                                   int32_t *slot_of_e, int32_t *lc_of_e, int8_t *ent_of_slot, void *stream);
/* Values only, for an operator that shares pattern (rowptr / col), parts, dmax and layout with one packed before -- its
 * transpose on a symmetric pattern, or P and Q of one batch: pass the first pack's workspace and its ent_of_slot
 * (total_d * 64 bytes, optional output of escat_sell_pack_fill: which entry of its row each slot holds, -1 = padding);
 * lc, lane_meta and pos_of_row of the first pack are then shared as they are.  Not for sign-bit packs. */
int escat_sell_pack_refill(const int32_t *rowptr, const float *val, int64_t n, const int32_t *part_ptr, int n_parts,
                           int values_per_entry, const escat_sell_dims *dims, const void *workspace,
                           size_t workspace_bytes, const int8_t *ent_of_slot, float *val_out, void *stream);

/* The same for a sign-bit pack (values_per_entry 3 as |value| + sign bits: escat_sell.val_planes = 4) on the plan of a pack
 * with the same pattern, parts, capacity and layout -- P's own pack, for the signed operator Pt of the factored vector
 * operator: val_out ((total_d + 1) * 64 floats) = the magnitudes, lc_out (dims->lc_words) = the first pack's lc with the
 * entries' sign bits in bits 11..13 of every 16-bit field.  lane_meta and pos_of_row are shared as they are. */
int escat_sell_pack_refill_signed(const int32_t *rowptr, const float *mag, const int32_t *sign_bits, int64_t n,
                                  const int32_t *part_ptr, int n_parts, const escat_sell_dims *dims, const void *workspace,
                                  size_t workspace_bytes, const int8_t *ent_of_slot, const uint32_t *lc, float *val_out,
                                  uint32_t *lc_out, void *stream);

/* The half-valued form of a BSR-3 pack (escat_sell.half_val / .half_aux) from the canonical blocks and the plan of a pack
 * built before on the same pattern (workspace, ent_of_slot and lc of escat_sell_pack_fill, layout 1).  half_val_out:
 * (dims->total_d + 1) * 64 * 16 bytes, 16-byte aligned; half_aux_out: (dims->total_d + 1) * 64 words.  *not_exact (device,
 * caller zeroes) is set to 1 if some value is not exactly representable as an IEEE half (NaN included): the outputs are
 * then meaningless and the caller keeps the fp32 pack. */
int escat_sell_pack_refill_half(const int32_t *rowptr, const float *blk, int64_t n, const int32_t *part_ptr, int n_parts,
                                const escat_sell_dims *dims, const void *workspace, size_t workspace_bytes,
                                const int8_t *ent_of_slot, const uint32_t *lc, void *half_val_out, uint32_t *half_aux_out,
                                int32_t *not_exact, void *stream);

/* COO ingest of the two layouts the reference's pipeline produces (models/escgnn_data_classes.py:137-142: int64 COO,
 * flagged un-coalesced): blocks3 = 0: entries already row-major sorted and unique (P, process_pyg_data.py:775) ->
 * rowptr (n + 1) / col32 (nnz); blocks3 = 1: block-emission order (Q, :839-857: nine entries per 3x3 block, row-major
 * inside, blocks in P's order) -> rowptr (n/3 + 1) / col32 (nnz/9) of the BSR-3 form, whose blocks are the COO values
 * as they lie.  *bad (device, caller zeroes) is set if the tensor is not in that layout: the outputs are then
 * meaningless and the caller must coalesce the general way.  One elementwise pass, no sort. */
int escat_ingest_coo(const int64_t *rows, const int64_t *cols, int64_t nnz, int64_t n, int blocks3,
                     int32_t *rowptr, int32_t *col32, int32_t *bad, void *stream);

/* Finding the diagonal blocks of a block-diagonal operator (the graphs of a PyG batch,
 * models/escgnn_data_classes.py:20-27).  escat_row_cover: delta (n + 1 ints, caller zeroes) receives +1 / -1 at the ends
 * of every row's column span; after a prefix sum, a cut after row c is legal iff the sum at c is 0.
 * escat_merge_row_ranges (HOST pointers, no GPU): greedy merge of consecutive blocks bounds[0..nb) into parts of at
 * most `target` rows; returns the number of boundaries written to out, *max_rows = the largest part. */
int escat_row_cover(const int32_t *rowptr, const int32_t *col, int64_t n, int32_t *delta, void *stream);
int64_t escat_merge_row_ranges(const int64_t *bounds, int64_t nb, int64_t target, int32_t *out, int32_t *max_rows);

/* The same ingest for a tensor handed over ONCE (a DataLoader's fresh batch, models/escgnn_data_classes.py:137-142), where
 * every launch and host round trip shows (round 3):
 * escat_ingest_coo_cover = escat_ingest_coo(blocks3 = 0) that also leaves escat_row_cover's marks in delta (n + 1 ints,
 *   zeroed here) -- the pattern is read once for both;
 * escat_row_blocks: from those marks to the parts, in one call -- prefix sum, cuts and their compaction as three small
 *   kernels, the cuts copied to the host, escat_merge_row_ranges there.  Synchronises the stream.  delta is overwritten;
 *   workspace (device) of escat_row_blocks_workspace_bytes(n); out_host: room for out_cap boundaries (HOST).  Returns
 *   the number of boundaries or a negative error code.  flag_dev / flag_host (optional): a device word -- the *bad of
 *   the ingest that made the marks -- comes to the host in the same copy; when it is non-zero the call returns 0;
 * escat_ingest_coo_like: Q in block-emission order whose blocks sit exactly on a pattern already ingested (P of the same
 *   batch: ref_rowptr (n/3 + 1), ref_col (ref_nnz) = its canonical CSR).  Writes nothing but *flags (device, caller
 *   zeroes): 0 = same pattern, the caller reuses P's rowptr / col / parts / pack plan; bit 0 = not in block-emission
 *   order; bit 1 = another pattern (then escat_ingest_coo decides). */
int escat_ingest_coo_cover(const int64_t *rows, const int64_t *cols, int64_t nnz, int64_t n,
                           int32_t *rowptr, int32_t *col32, int32_t *bad, int32_t *delta, void *stream);
size_t escat_row_blocks_workspace_bytes(int64_t n);
int64_t escat_row_blocks(int32_t *delta, int64_t n, int64_t target, void *workspace, size_t workspace_bytes,
                         int32_t *out_host, int64_t out_cap, int32_t *max_rows, const int32_t *flag_dev, int32_t *flag_host,
                         void *stream);
int escat_ingest_coo_like(const int64_t *rows, const int64_t *cols, int64_t nnz, int64_t n,
                          const int32_t *ref_rowptr, const int32_t *ref_col, int64_t ref_nnz, int32_t *flags, void *stream);

/* The transposed operator for the backward entry points when the PATTERN is symmetric (undirected graphs: P and Q of
 * process_pyg_data always are): A^T then lives on A's own rowptr / col, and val_t[(j,i)] = val[(i,j)] (3x3 blocks
 * transposed for values_per_entry = 9).  *not_symmetric (device, caller zeroes) is set to 1 if some (i,j) has no (j,i):
 * val_t is then incomplete and the caller must transpose the general way. */
int escat_transpose_sym(const int32_t *rowptr, const int32_t *col, const float *val, int64_t n, int64_t nnz,
                        int values_per_entry, float *val_t, int32_t *not_symmetric, void *stream);

/* Pack-time planning for escat_sell: for every entry e among the first dmax of its row, the slot (column of its
 * slice) and the LDS row (vector copy A or B) it will use, chosen so that the 16 lanes of each ds_read_b128 lane
 * group land on distinct 16-byte bank slots where possible.  slice_part[s] = part of slice s, perm[s*64 + lane] = local row of the lane (-1: none); slot_of_e must be
 * pre-filled with -1 (entries beyond dmax keep it). */
int escat_sell_plan_slots(const int32_t *rowptr, const int32_t *col, const int32_t *part_ptr,
                          const int32_t *slice_part, const int32_t *perm, int n_slices, int dmax,
                          int32_t *slot_of_e, int32_t *lc_of_e, void *stream);

/* last error message of the calling thread ("" if none). */
const char *escat_last_error(void);
/* ABI version, bumped when a signature or a struct changes (4: escat_sell.ent_of_slot, escat_sell_dims.row_len_hist, the
 * single-pass ingest entry points; 5: escat_chain_both_f32 / _bwd_f32). */
int escat_abi_version(void);

/* Y = A X.  Replaces one torch.sparse.mm(P_sparse, Ptx) call
 * (pyg_utilities.py:365, :401, :478).  X is (n, C) with row stride ldx,
 * Y (n, C) with row stride ldy, 1 <= C <= ESCAT_MAX_CHANNELS. */
int escat_csr_spmm_f32(const int32_t *rowptr, const int32_t *col, const float *val,
                       int64_t n, int64_t nnz, const float *X, int ldx, float *Y, int ldy,
                       int C, void *stream);

/* y = Q x, Q in BSR-3, x and y of length 3*nb.  Replaces torch.sparse.mm(Q, v)
 * on the vector track (models/escgnn_modular.py:345-351 -> pyg_utilities.py:365). */
int escat_bsr3_spmv_f32(const int32_t *rowptr, const int32_t *col, const float *blk,
                        int64_t nb, int64_t nnzb, const float *x, float *y, void *stream);

/* bytes of workspace the chain entry points need for n scalar rows and C
 * channels when they take the multi-launch path (two ping-pong vectors). */
size_t escat_chain_workspace_bytes(int64_t n_rows, int C);

/* Forward filter bank, scalar track.  Replaces the whole of
 * _get_Wjxs_from_shared_powers (pyg_utilities.py:350-385), the per-channel
 * variant (:392-444) and the P^1 case (:476-479): T products, the kept powers,
 * the differences and the stack, written once into OUT (n, C, F) contiguous.
 *   events       host pointer, n_events <= ESCAT_MAX_EVENTS, sorted by step.
 *   part_ptr     device int32[n_parts+1] or NULL: row ranges that are closed
 *                under the operator's columns (the diagonal blocks of a PyG
 *                batch, models/escgnn_data_classes.py:20-27).  When given and
 *                max_part_rows fits the LDS budget, each range is processed by
 *                one workgroup with the vectors resident in LDS for all T
 *                steps; otherwise T step kernels run over the whole operator
 *                using `workspace`.
 *   sell         host pointer to a packed copy of the operator (escat_sell
 *                below) or NULL.  With it, and parts of <= 512 rows, one thread
 *                owns one row and keeps the row's entries in registers for all
 *                T steps: the operator is read from HBM once per chain.
 */
int escat_chain_csr_f32(const int32_t *rowptr, const int32_t *col, const float *val,
                        int64_t n, int64_t nnz, const float *X, int C, int T,
                        const escat_event *events, int n_events, int F, float *OUT,
                        const int32_t *part_ptr, int n_parts, int max_part_rows, const escat_sell *sell,
                        void *workspace, size_t workspace_bytes, void *stream);

/* The one entry point that ALLOCATES NOTHING.  It has no counterpart in the reference; it is for callers that cannot
 * call hipMalloc or pass a workspace they allocated: code under HIP graph capture, tight serving loops, harnesses that
 * forbid allocation.  Same arguments and results as escat_chain_csr_f32 minus the workspace: when the call takes the
 * multi-launch path (no part_ptr / parts that do not fit LDS) the two ping-pong vectors live in a static device array
 * of ESCAT_STATIC_SCRATCH_BYTES that every GPU the library runs on holds until the process ends.  Calls whose
 * escat_chain_workspace_bytes(n, C) exceeds that are rejected with ESCAT_ERANGE.  The array's device address is looked
 * up on the first call on each device (hipGetSymbolAddress; make that call outside graph capture).  All calls on one GPU
 * share the array: calls that overlap on one GPU (two streams) race; calls on one stream are ordered and safe.  The
 * vectors are fully written before they are read, so nothing is cleared between calls. */
#define ESCAT_STATIC_SCRATCH_BYTES (8u << 20)
int escat_chain_csr_static_f32(const int32_t *rowptr, const int32_t *col, const float *val,
                               int64_t n, int64_t nnz, const float *X, int C, int T,
                               const escat_event *events, int n_events, int F, float *OUT,
                               const int32_t *part_ptr, int n_parts, int max_part_rows, const escat_sell *sell,
                               void *stream);

/* Backward of the above w.r.t. X (what autograd of the reference's chain
 * computes when x.requires_grad): dX = sum_events sign * (A^T)^step G[:, chan, slot].
 * Takes the TRANSPOSED operator in CSR (rows of A^T sorted by column) and the
 * upstream gradient G (n, C, F) contiguous. */
int escat_chain_csr_bwd_f32(const int32_t *rowptr_t, const int32_t *col_t, const float *val_t,
                            int64_t n, int64_t nnz, const float *G, int C, int T,
                            const escat_event *events, int n_events, int F, float *dX,
                            const int32_t *part_ptr, int n_parts, int max_part_rows, const escat_sell *sell,
                            void *workspace, size_t workspace_bytes, void *stream);

/* Forward filter bank, vector track: x is the (3*nb,) flattening of (nb, 3)
 * node vectors (models/escgnn_modular.py:345), OUT is (3*nb, 1, F) contiguous
 * (what get_Batch_Wjxs returns for x of shape (3N, 1)).  part_ptr counts
 * BLOCK rows. */
int escat_chain_bsr3_f32(const int32_t *rowptr, const int32_t *col, const float *blk,
                         int64_t nb, int64_t nnzb, const float *x, int T,
                         const escat_event *events, int n_events, int F, float *OUT,
                         const int32_t *part_ptr, int n_parts, int max_part_rows, const escat_sell *sell,
                         void *workspace, size_t workspace_bytes, void *stream);

/* Backward of the vector-track bank; takes Q^T in BSR-3 (block (j,i) of Q^T is
 * the transpose of block (i,j) of Q). */
int escat_chain_bsr3_bwd_f32(const int32_t *rowptr_t, const int32_t *col_t, const float *blk_t,
                             int64_t nb, int64_t nnzb, const float *G, int T,
                             const escat_event *events, int n_events, int F, float *dx,
                             const int32_t *part_ptr, int n_parts, int max_part_rows, const escat_sell *sell,
                             void *workspace, size_t workspace_bytes, void *stream);

/* BOTH tracks of a batch in one launch (round 3).  ESCGNNModular.forward runs _scatter twice per batch -- scalar track on
 * P (models/escgnn_modular.py:674), vector track on Q (:685) -- i.e. two get_Batch_Wjxs chains (pyg_utilities.py:350-385) on
 * one pattern.  When Q has the factored form Q = D Pt D^T (escat_sell.frames, val_planes = 4) AND |Pt_ij| = P_ij -- what
 * process_pyg_data builds (process_pyg_data.py:306-313, :492, :823-857) -- one thread serves its row for both tracks: the
 * operator is streamed once, every step costs one barrier and one LDS latency chain instead of two.
 *   rowptr / col_signed / mag   the signed operator's global arrays (col | signs << 29, |value| = P's values >= 0); only
 *                               rows longer than the pack's capacity read them
 *   Xs (n, 3)                   scalar-track input; the scalar track uses |value|, i.e. P itself
 *   Xv (n, 3)                   vector-track input = the (3n,) flattening of the node vectors, global coordinates
 *   OUTs, OUTv (n, 3, F)        the two banks ((3n, 1, F) for the vector track), as the two separate entry points write
 *                               them -- bit for bit (same arithmetic, same order)
 *   sell                        the sign-bit pack (val_planes = 4, layout 1, frames, pos_of_row), orders = 0; part_list
 *                               views are honoured
 * Register-resident path only (parts of <= 512 rows); event tables in which every slot is one power or the difference of
 * two powers, with at most one difference open at a time and no per-channel or ESCAT_EVENT_ABS events (all shared-scale
 * banks of get_Batch_Wjxs).  Anything else returns ESCAT_ERANGE and the caller runs the two chains separately.  The
 * kernel holds no output slab in LDS: finished bands go from registers straight to OUT. */
int escat_chain_both_f32(const int32_t *rowptr, const int32_t *col_signed, const float *mag, int64_t n, int64_t nnz,
                         const float *Xs, const float *Xv, int T, const escat_event *events, int n_events, int F,
                         float *OUTs, float *OUTv, const int32_t *part_ptr, int n_parts, int max_part_rows,
                         const escat_sell *sell, void *stream);
/* its backward: the TRANSPOSED signed operator (pack, global arrays), upstream gradients Gs, Gv (n, 3, F) -> dXs, dXv (n, 3);
 * at most two events per step. */
int escat_chain_both_bwd_f32(const int32_t *rowptr_t, const int32_t *col_signed_t, const float *mag_t, int64_t n, int64_t nnz,
                             const float *Gs, const float *Gv, int T, const escat_event *events, int n_events, int F,
                             float *dXs, float *dXv, const int32_t *part_ptr, int n_parts, int max_part_rows,
                             const escat_sell *sell, void *stream);

/* Orders 0/1/2 of ESCGNNModular._scatter (models/escgnn_modular.py:333-341,
 * :355-360): W (rows, Wtot) = [x0 | W1 | W1[i]*W1[j] for i<j in triu order
 * (:507-516)], Wtot = 1 + F + (second_order ? F(F-1)/2 : 0).  rows = n*C. */
int escat_scatter_orders_f32(const float *x0, const float *W1, int64_t rows, int F,
                             int second_order, float *W, void *stream);
/* its backward: dW (rows, Wtot) -> dx0 (rows,), dW1 (rows, F).  W1 is read with row stride ldw1 floats (F for a
 * dense bank; Wtot with W1 = W + 1 to read the bank out of the forward's own output). */
int escat_scatter_orders_bwd_f32(const float *W1, int64_t ldw1, const float *dW, int64_t rows, int F,
                                 int second_order, float *dx0, float *dW1, void *stream);

/* ---------------------------------------------------------------------------
 * Operator build (process_pyg_data, data_processing/process_pyg_data.py:640-919),
 * batched over the graphs of a dataset.  pos is (n_nodes, 3) fp32, graph_ptr
 * int32[n_graphs+1] gives each graph's node range.
 * ------------------------------------------------------------------------- */

/* workspace bytes for escat_build_knn_P. */
size_t escat_build_workspace_bytes(int64_t n_nodes, int n_graphs, int k);

/* kNN graph (k nearest, no self loops: knn_graph, ellipsoid_data_classes.py:209)
 * -> symmetrised (to_undirected, :211) -> P = 1/2 (I + D^-1 A) (get_P,
 * process_pyg_data.py:499-600, unit edge weights) in CSR with sorted columns,
 * diagonal included: exactly the coalesced pattern of the reference's P.
 * cap = capacity of col / pval; n_nodes*(2k+1) always suffices.  *nnz_out
 * (device) receives the number of nonzeros.  Graphs of up to ~1090 nodes. */
int escat_build_knn_P(const float *pos, const int32_t *graph_ptr, int n_graphs, int64_t n_nodes,
                      int max_graph_nodes, int k, int32_t *rowptr, int32_t *col, float *pval,
                      int64_t cap, int32_t *nnz_out, void *workspace, size_t workspace_bytes, void *stream);

/* The same outputs for ONE point cloud of any size (cfg3: a 2 M-point cloud; data_processing/ellipsoid_data_classes.py:
 * 209-211 builds knn_graph + to_undirected over the whole cloud): the points are binned into a uniform grid in HBM
 * (cell edge chosen from the measured occupancy, ~4 points per occupied cell), every point searches rings of cells
 * until its k-th neighbour lies inside the scanned cube -- exact kNN, fp64 distances, ties to the smaller index, i.e.
 * the lists of escat_build_knn_P -- and the symmetrisation runs on the lists themselves instead of an LDS bitmap.
 * Columns are local to the cloud.  Synchronises `stream` (it reads the bounding box and the occupancy back). */
size_t escat_build_grid_workspace_bytes(int64_t n_nodes, int k);
int escat_build_knn_P_grid(const float *pos, int64_t n_nodes, int k, int32_t *rowptr, int32_t *col, float *pval,
                           int64_t cap, int32_t *nnz_out, void *workspace, size_t workspace_bytes, void *stream);

/* Local-PCA frames O_i (get_C_i_dict :330-451 + get_l_singular_vecs :238-286):
 * frames[i][a][c] = component a of the c-th left singular vector of
 * C_i = (w_j (v_j - v_i))_j over the neighbours j of i in the CSR pattern
 * (entries with col == row are skipped).  kernel: 0 gaussian (exp(-r^2/eps),
 * or exp(-r^2) without eps), 1 cosine_cutoff, 2 epanechnikov (:79-138);
 * take_sqrt as the reference's default.  *n_isolated (device, caller zeroes)
 * counts nodes without neighbours (the reference raises KeyError for them). */
int escat_build_frames(const int32_t *rowptr, const int32_t *col, const float *pos, int64_t n_nodes,
                       int kernel, float gaussian_eps, int has_eps, float r_cut, int has_rcut,
                       int take_sqrt, int mean_recenter, float *frames, int32_t *n_isolated, void *stream);

/* The same with the reference's rank_deficiency_strategy='Tikhonov' (get_l_singular_vecs, :262-279): when tikhonov != 0
 * a node with fewer than 3 neighbours takes the eigenvectors of C_i C_i^T + tikhonov_eps * I, by descending eigenvalue. */
int escat_build_frames_ex(const int32_t *rowptr, const int32_t *col, const float *pos, int64_t n_nodes,
                          int kernel, float gaussian_eps, int has_eps, float r_cut, int has_rcut,
                          int take_sqrt, int mean_recenter, int tikhonov, float tikhonov_eps,
                          float *frames, int32_t *n_isolated, void *stream);

/* Q blocks on P's pattern (calc_O_ij :454-493, loop :823-857): blk[e] =
 * pval[e] * O_i O_j'^T with O_j' = column-dot sign alignment of O_j to O_i
 * (procrustes = 0) or the reference's Procrustes formula (procrustes = 1);
 * enforce_sign = 0 skips the alignment.  blk is BSR-3 on (rowptr, col). */
int escat_build_Q_bsr3(const int32_t *rowptr, const int32_t *col, const float *pval, const float *frames,
                       int64_t n_nodes, int enforce_sign, int procrustes, float *blk, void *stream);

/* ---- neighbour pooling of per-edge values (ESCGNNModular._get_vector_invariants, models/escgnn_modular.py:431-502).
 * vals (E, W): one row per edge, edges grouped by destination node: node n owns rows seg_ptr[n] .. seg_ptr[n+1]
 * (at most max_deg <= 64 of them).  Per node and column: mean (stats bit 0), max (1), min (2), variance
 * E[x^2]-E[x]^2 clamped at 0 (3), then n_levels lower quantiles: the k-th smallest with k = k_table[l*65 + deg]
 * (the caller tabulates floor((deg-1)*p_l) exactly as the reference's Python does, :493-496).  Nodes without
 * incoming edges get 0 everywhere.  out (n_nodes, m*W) = the reference's torch.cat of (N, W) blocks in the order
 * mean, max, min, var, levels; arg (n_nodes, na*W) bytes, na = [max] + [min] + n_levels: position inside the node's
 * segment of the selected edge (first in edge order among equals), for the backward pass; may be NULL. */
int escat_neighbor_pool_f32(const float *vals, const int32_t *seg_ptr, int64_t n_nodes, int W, int max_deg,
                            int stats, int n_levels, const int32_t *k_table, float *out, uint8_t *arg, void *stream);
/* gradient w.r.t. vals: gout (n_nodes, m*W) -> gvals (E, W); every element of gvals is written. */
int escat_neighbor_pool_bwd_f32(const float *vals, const int32_t *seg_ptr, int64_t n_nodes, int W, int stats,
                                int n_levels, const float *gout, const uint8_t *arg, float *gvals, void *stream);

/* ---- the whole of _get_vector_invariants in one kernel per direction (models/escgnn_modular.py:414-504):
 * v (n_nodes, W, d) coefficients (d = 2 or 3); in-edges of node n = src[seg_ptr[n] .. seg_ptr[n+1]) (edge_index
 * grouped by destination, edge order kept).  Per node and wavelet: |v| and the pooled statistics (as above) of
 * cos(v[n], v[src]) with both norms clamped at 1e-8 (:425-429).  out (n_nodes, (1+m)*W) = [norm | mean | max | min |
 * var | levels]; arg as above.  Nothing of size (E, W) is materialised. */
int escat_vector_invariants_f32(const float *v, const int32_t *src, const int32_t *seg_ptr, int64_t n_nodes, int W, int d,
                                int max_deg, int stats, int n_levels, const int32_t *k_table, float *out, uint8_t *arg,
                                void *stream);
/* gradient w.r.t. v.  Two forms:
 *  - gedge == NULL: one kernel, float atomics; dv (n_nodes, W, d) must be ZERO on entry;
 *  - gedge != NULL (workspace of E*W floats): two passes without atomics, dv is fully written; needs the edges also
 *    grouped by source: dst[e] = destination of edge e (in the destination-grouped order the forward used),
 *    out_edge[out_ptr[n] .. out_ptr[n+1]) = the edges e whose source is node n. */
int escat_vector_invariants_bwd_f32(const float *v, const int32_t *src, const int32_t *seg_ptr, int64_t n_nodes, int W, int d,
                                    int max_deg, int stats, int n_levels, const float *gout, const uint8_t *arg, float *dv,
                                    const int32_t *dst, const int32_t *out_ptr, const int32_t *out_edge, float *gedge,
                                    void *stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Infogain scale selection (models/infogain.py:172-664): per-graph divergence of every diffusion power against the
 * last one.  Replaces the Python loops of get_divergence_by_channel (models/infogain.py:810-900) + normalize_channels
 * (:780-807) + get_floor (:760-777) + nn_utilities.norm_tensor_to_prob_mass / minmax_scale_tensor /
 * get_mid_btw_min_and_2nd_low_vector_vals (models/nn_utilities.py:76-245) over graphs x channels x steps.
 *   W          (N, C, F) fp32: the powers P^t x, t = 0..F-1, step fastest -- what escat_chain_*_f32 writes for the
 *              all-powers event table {step t -> slot t, +1} (the stack of :397-404 without the (T+1, N, C) copy)
 *   graph_ptr  (n_graphs + 1) row offsets of the graphs in W (rows = d * nodes for a flattened vector feature)
 *   start      first step compared (Ptx_i_start, :317); steps start .. F-2 are compared with step F-1
 *   metric     0 = KL(q_T || p_t) = sum q (log q - log(p + 1e-12)) (:857-863), 1 = L1 (:866)
 *   fixed_floor / floor_val   1: above_zero_floor = floor_val (divergence_metric_kwargs['above_zero_floor']);
 *              0: midpoint of the lowest and second-lowest distinct value of each column (nn_utilities.py:202-245)
 *   max_graph_rows  largest graph_ptr[g+1] - graph_ptr[g] if the caller knows it (the kernel then keeps each graph in
 *              registers and reads W once; it must be an upper bound), 0 = unknown (three passes from L2)
 *   out        (n_graphs, F-1-start, C) fp32, summed over the graph's rows (:891-898)
 *   status     (n_graphs) int32, set to 1 where a column cannot be normalised (no variance / zero floor: the
 *              reference fails with a TypeError there); the corresponding out entries are NaN */
int escat_infogain_divergence_f32(const float *W, const int32_t *graph_ptr, int64_t n_graphs, int C, int F, int start,
                                  int metric, int fixed_floor, float floor_val, int max_graph_rows, float *out,
                                  int32_t *status, void *stream);

/* Rotation by the local frames O_n of the P/Q builder (data_processing/process_pyg_data.py:238-286), for the factored
 * form of the vector operator Q = D Pt D^T (escat_sell.val_planes = 3; the reference only ever forms the blocks
 * p_ij O_i S_ij O_j^T, :454-493, :839-847):  Y[n][a][f] = sum_c O[n][a][c] X[n][c][f]  (transpose != 0: O[n][c][a]).
 * frames (n_nodes, 3, 3), X and Y (n_nodes, 3, F), X != Y. */
int escat_rotate_frames_f32(const float *frames, const float *X, float *Y, int64_t n_nodes, int F, int transpose,
                            void *stream);

/* The factored form of a BSR-3 vector operator recovered from its blocks alone (csrc/escat_factor.hip).  process_pyg_data
 * builds Q_ij = p_ij O_i S_ij O_j^T, S_ij a diagonal sign matrix (process_pyg_data.py:306-313, :492), but hands the caller
 * only the blocks (models/escgnn_modular.py:345-351; the frames are not in the HDF5 layout,
 * generate_ellipsoid_dataset.py:1185-1206).  Per part (part_ptr: block rows, closed under the columns, <= 512 rows) this
 * finds frames G (nb, 3, 3) and per-channel values val3 (nnzb, 3) = p_ij * diag(S_ij) such that Q_ij = p_ij G_i S_ij G_j^T:
 * what escat_sell.frames + a val_planes = 3 / 4 operator take.  *max_err (device float, caller zeroes) receives the largest
 * |p_ij G_i S_ij G_j^T - Q_ij| over all block entries: the caller accepts the factorisation iff that is within its
 * tolerance (blocks rounded to fp16, the 'procrustes' alignment or any other operator leave it large). */
size_t escat_recover_factored_lds_bytes(int max_part_rows, int max_part_nnz);
int escat_recover_factored(const int32_t *rowptr, const int32_t *col, const float *blk, int64_t nb, int64_t nnzb,
                           const int32_t *part_ptr, int n_parts, int max_part_rows, int max_part_nnz /* most entries of a part */,
                           float *frames_out, float *val3_out, float *max_err, void *stream);

/* host utility (no GPU): Jenkins lookup3 "hashlittle", the checksum of HDF5 version-2 object headers
 * (esc_gnn_amd/hdf5_min.py, the writer of the reference's operator layout). */
uint32_t escat_lookup3(const uint8_t *data, size_t n, uint32_t initval);

#ifdef __cplusplus
}
#endif
#endif /* ESCAT_H */
