"""
Operator ingest: torch sparse tensors -> the canonical device formats the
kernels consume (include/escat.h).

What arrives: P and Q as torch sparse COO, int64 indices,
fp32 values, flagged un-coalesced, block-diagonal over the graphs of a PyG batch
(models/escgnn_data_classes.py:20-27, :137-142); Q in block-emission order
(process_pyg_data.py:847-857).  What the kernels want: int32 CSR sorted by
(row, col) -- the order torch's own coalesce() gives, so the index structure is
bit-identical to the reference's coalesced operator -- or BSR-3 for an operator
made of 3x3 blocks, the transposed operator for the backward walk, and the row
ranges closed under the operator's columns (the diagonal blocks) so that one
workgroup can own one graph.

The sort/unique calls here are torch device ops (plumbing, done once per
operator and cached on the tensor object); the products themselves never touch
torch.
"""
import os
import weakref
from typing import Optional
import numpy as np
import torch

PART_TARGET_ROWS = 256      # merge consecutive diagonal blocks up to this many (scalar) rows per workgroup


def _rowptr(rows: torch.Tensor, n: int) -> torch.Tensor:
    cnt = torch.bincount(rows, minlength=n)
    rp = torch.zeros(n + 1, dtype=torch.int64, device=rows.device)
    torch.cumsum(cnt, 0, out=rp[1:])
    return rp.to(torch.int32)


_PARTS_CACHE = {}        # one entry: operators that share a pattern (P and Q of a batch) find their parts once


def _closed_row_ranges(rowptr: torch.Tensor, col: torch.Tensor, n: int, target: int):
    """
    Boundaries of the finest partition of rows 0..n into consecutive ranges that
    no entry leaves (cut after row r iff every row <= r has cols <= r and every
    row > r has cols > r), then greedily merged up to `target` rows.
    Returns (part_ptr int32 device tensor, n_parts, max_part_rows).
    On the GPU: one kernel marks every row's column span (escat_row_cover), a prefix sum finds the cuts no span crosses,
    the greedy merge of the (few thousand) blocks is a C loop on the host (escat_merge_row_ranges).
    """
    dev = rowptr.device
    if n == 0:
        return torch.zeros(1, dtype=torch.int32, device=dev), 0, 0
    key = (rowptr.data_ptr(), col.data_ptr(), rowptr._version, col._version, n, int(col.numel()), target)
    hit = _PARTS_CACHE.get("last")
    if hit is not None:
        crp, ccl = hit[1](), hit[2]()
        if hit[0] == key and crp is rowptr and ccl is col:
            return hit[3]
        # the same pattern in other tensors: P and Q of one batch arrive as separate COO tensors whose canonical patterns
        # (scalar rows of P = block rows of Q) are equal element for element; two compares instead of the scan
        if crp is not None and ccl is not None and hit[0][4:] == key[4:] and crp.device == dev and crp.dtype == rowptr.dtype \
                and ccl.dtype == col.dtype and crp._version == hit[0][2] and ccl._version == hit[0][3] \
                and bool(torch.equal(ccl, col)) and bool(torch.equal(crp, rowptr)):
            return hit[3]
    if rowptr.is_cuda and rowptr.dtype == torch.int32 and col.dtype == torch.int32:
        import ctypes
        from . import _lib
        lib = _lib.load()
        delta = torch.zeros(n + 1, dtype=torch.int32, device=dev)
        with torch.cuda.device(dev):
            _lib.check(lib.escat_row_cover(ctypes.c_void_p(rowptr.data_ptr()), ctypes.c_void_p(col.data_ptr() if col.numel() else 0),
                                           n, ctypes.c_void_p(delta.data_ptr()),
                                           ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)), "escat_row_cover")
        cover = torch.cumsum(delta[:n - 1], 0, dtype=torch.int32) if n > 1 else delta[:0]
        cuts = (torch.nonzero(cover == 0).flatten() + 1).cpu().numpy()
        bounds = np.ascontiguousarray(np.concatenate([[0], cuts, [n]]).astype(np.int64))
        out = np.empty(bounds.shape[0], dtype=np.int32)
        mx = ctypes.c_int32(0)
        k = lib.escat_merge_row_ranges(bounds.ctypes.data_as(ctypes.c_void_p), bounds.shape[0], target,
                                       out.ctypes.data_as(ctypes.c_void_p), ctypes.byref(mx))
        if k < 0:
            _lib.check(int(k), "escat_merge_row_ranges")
        if k < 2:
            raise RuntimeError(f"escat_merge_row_ranges returned {int(k)} boundaries for n = {n} rows (needs >= 2)")
        res = (torch.from_numpy(out[:k].copy()).to(dev), int(k) - 1, int(mx.value))
        _PARTS_CACHE["last"] = (key, weakref.ref(rowptr), weakref.ref(col), res)    # weak: the cache pins no pattern
        return res
    rp = rowptr.long()
    ar = torch.arange(n, device=dev)
    if col.numel() == 0:
        lo = hi = ar
    else:
        nonempty = rp[1:] > rp[:-1]
        cl = col.long()
        lo = torch.where(nonempty, cl[rp[:-1].clamp(max=cl.numel() - 1)], ar)   # rows are sorted by column
        hi = torch.where(nonempty, cl[(rp[1:] - 1).clamp(min=0)], ar)
        lo = torch.minimum(lo, ar)
        hi = torch.maximum(hi, ar)
    pm = torch.cummax(hi, 0).values                       # max col over rows <= r
    sm = torch.flip(torch.cummin(torch.flip(lo, [0]), 0).values, [0])   # min col over rows >= r
    ok = pm[:-1] <= ar[:-1]
    ok &= sm[1:] >= ar[1:]
    cuts = (torch.nonzero(ok).flatten() + 1).cpu().numpy()
    bounds = np.concatenate([[0], cuts, [n]]).astype(np.int64)
    # greedy merge (host, O(#blocks), once per operator)
    out = [0]
    start = 0
    for i in range(1, len(bounds)):
        if bounds[i] - start > target and bounds[i - 1] > start:
            out.append(int(bounds[i - 1])); start = int(bounds[i - 1])
    out.append(n)
    out = np.unique(np.asarray(out, dtype=np.int64))
    max_rows = int(np.diff(out).max())
    return torch.from_numpy(out.astype(np.int32)).to(dev), len(out) - 1, max_rows


def _row_blocks_from_marks(delta: torch.Tensor, rowptr: torch.Tensor, col: torch.Tensor, n: int, target: int, bad=None):
    """_closed_row_ranges for a pattern whose row-cover marks are already there (escat_ingest_coo_cover): one library
    call (three small kernels, one copy of the cuts to the host, the merge there) and the upload of the part
    boundaries.  The result enters the parts cache, where Q of the same batch finds it.  bad: the ingest's flag word
    (device), read in the same round trip -- returns None when it is set."""
    import ctypes
    from . import _lib
    lib = _lib.load()
    dev = delta.device
    ws = torch.empty(lib.escat_row_blocks_workspace_bytes(n) // 4, dtype=torch.int32, device=dev)
    cap = n + 2                                          # (never touched beyond the boundaries written)
    out = np.empty(cap, dtype=np.int32)
    mx, flag = ctypes.c_int32(0), ctypes.c_int32(0)
    with torch.cuda.device(dev):
        k = lib.escat_row_blocks(ctypes.c_void_p(delta.data_ptr()), n, target, ctypes.c_void_p(ws.data_ptr()), ws.numel() * 4,
                                 out.ctypes.data_as(ctypes.c_void_p), cap, ctypes.byref(mx),
                                 ctypes.c_void_p(bad.data_ptr() if bad is not None else 0), ctypes.byref(flag),
                                 ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
    if bad is not None and flag.value:
        return None
    if k < 0:
        _lib.check(int(k), "escat_row_blocks")
    if k < 2:
        raise RuntimeError(f"escat_row_blocks returned {int(k)} boundaries for n = {n} rows (needs >= 2)")
    res = (torch.from_numpy(out[:k].copy()).to(dev), int(k) - 1, int(mx.value))
    key = (rowptr.data_ptr(), col.data_ptr(), rowptr._version, col._version, n, int(col.numel()), target)
    _PARTS_CACHE["last"] = (key, weakref.ref(rowptr), weakref.ref(col), res)
    return res


REORDER_MIN_ROWS = 16384     # a connected block this large cannot live in LDS: its x gather goes to L2/HBM every step


def locality_order(rowptr: torch.Tensor, col: torch.Tensor, n: int, max_components: int = 64) -> torch.Tensor:
    """
    Breadth-first (Cuthill-McKee style) ordering of the rows of a sparse structure: returns perm (new -> old) such that
    rows that reference each other get nearby new indices.  Inside a BFS level rows are ordered by the rank of their
    first-placed parent.  Torch index ops on the operator's device; O(levels) small launches, once per operator.
    Point clouds arrive in arbitrary node order (the reference keeps the sampler's), which makes every step's gather of
    x a random walk over HBM; after this ordering the rows a step touches together sit within a few BFS rings.
    """
    dev = rowptr.device
    rp, cl = rowptr.long(), col.long()
    deg = rp[1:] - rp[:-1]
    rank = torch.full((n,), -1, dtype=torch.int64, device=dev)
    placed, comps = 0, 0
    while placed < n:
        unv = torch.nonzero(rank < 0).flatten()
        if comps >= max_components:                       # many small pieces left: keep them in index order
            rank[unv] = placed + torch.arange(unv.numel(), device=dev)
            break
        comps += 1
        frontier = unv[:1]
        rank[frontier] = placed
        placed += 1
        while frontier.numel() > 0:
            cnt = deg[frontier]
            tot = int(cnt.sum())
            if tot == 0:
                break
            owner = torch.repeat_interleave(torch.arange(frontier.numel(), device=dev), cnt)
            first = torch.cumsum(cnt, 0) - cnt
            nb = cl[rp[frontier][owner] + (torch.arange(tot, device=dev) - first[owner])]
            prank = rank[frontier][owner]
            fresh = rank[nb] < 0
            nb, prank = nb[fresh], prank[fresh]
            if nb.numel() == 0:
                break
            key, order = torch.sort(nb * n + prank)       # by node, then by parent rank: first of a run = earliest parent
            nb_s = nb[order]
            head = torch.ones_like(nb_s, dtype=torch.bool)
            head[1:] = nb_s[1:] != nb_s[:-1]
            new_nodes, new_prank = nb_s[head], (key[head] - nb_s[head] * n)
            o2 = torch.argsort(new_prank, stable=True)    # level order: by first parent, ties by node index
            frontier = new_nodes[o2]
            rank[frontier] = placed + torch.arange(frontier.numel(), device=dev)
            placed += int(frontier.numel())
    perm = torch.empty(n, dtype=torch.int64, device=dev)
    perm[rank] = torch.arange(n, device=dev)
    return perm


SELL_MAX_PART_ROWS = 512
_SELL_DMAX = {1: (8, 16, 20, 24, 32), 3: (8, 16, 20, 24), 9: (8, 12, 16, 20, 22)}     # register capacities the kernels are instantiated for
# rows longer than the capacity stream their tail from L2 every step (slow for their whole workgroup), but the largest
# BSR-3 capacity leaves the kernel short of registers: accept a capacity that leaves at most this fraction of rows a tail
_SELL_TAIL_ROWS = 5e-5
# order of a 3x3 block's entries in the fused-chain pack: b00 b10 b01 b11 | b02 b12 b20 b21 | b22 (include/escat.h)
_BSR3_PIECE_ORDER = (0, 3, 1, 4, 2, 5, 6, 7, 8)


def _cumsum0(t):
    out = torch.zeros(t.numel() + 1, dtype=torch.int64, device=t.device)
    torch.cumsum(t, 0, out=out[1:])
    return out


def sell_pad_rows(rows: int) -> int:
    """R of include/escat.h: rows of one LDS vector copy (part rows + the zero row, rounded up to 16)."""
    return (rows + 1 + 15) & ~15


def sell_copy_b(p, R: int):
    """row of local column p inside copy B (see sell_copyB in csrc/escat.hip); p may be an int or an integer ndarray/tensor."""
    m = 8 if R >= 256 else max(R.bit_length() - 1, 0)
    if m < 5:
        return p
    blk, lim = 1 << m, (R >> m) << m
    lo = p & (blk - 1)
    q = (p & ~(blk - 1)) | (lo >> 4) | ((lo & 15) << (m - 4))
    if isinstance(p, int):
        return q if p < lim else p
    return np.where(p < lim, q, p) if isinstance(p, np.ndarray) else torch.where(p < lim, q, p)


def sell_swz_b(p):
    """row of position p inside copy B of a layout-1 pack (sell_swzB in csrc/escat_dev.h); int or integer array."""
    return (p & ~15) | ((p + 1 + ((p >> 5) % 15)) & 15)


_PLAN_CACHE = {}         # (pattern identity, parts, dmax, layout) -> weakref to the last pack built on it (see Sell)

# Half-valued BSR-3 packs (include/escat.h: escat_sell.half_val): "auto" -- when every block value is exactly
# representable as an IEEE half (operators that went through the reference's fp16 HDF5 storage) the pack holds halves
# and the chain runs escat_sell_bsr3h.hip, bit-identical to the fp32 pack's results; "0" -- never; "both" -- keep the
# fp32 pack beside it (tests; the library then takes the half pack unless ESCAT_Q_HALF=0 is in ITS environment).
HALF_MODE = os.environ.get("ESCAT_Q_HALF", "auto")
HALF_MAX_DMAX = 20

# Factored form of a BSR-3 vector operator recovered from its blocks (escat_recover_factored, csrc/escat_factor.hip):
# "auto" -- a BSR-3 operator whose blocks are p_ij G_i S_ij G_j^T to within FACTOR_TOL x max|Q| runs the vector chain in
# the factored kernel (3 FMAs per entry instead of 9, two workgroups per CU); "0" -- never.
FACTOR_MODE = os.environ.get("ESCAT_Q_FACTOR", "auto")
# "direct" BSR-3 chains (no value pack, include/escat.h: escat_sell.ent_of_slot) for single-use operators; "0": never
DIRECT_MODE = os.environ.get("ESCAT_Q_DIRECT", "auto")
FACTOR_TOL = 1e-6
# one large connected cloud (relabelled breadth-first at ingest, BASELINE cfg3): recover the frames across the whole device
# (esc_gnn_amd/recover_large.py; ~3.5 s once for 2 M nodes, the chain then runs 2.5x faster).  "0": such operators stay BSR-3.
LARGE_RECOVERY = os.environ.get("ESCAT_Q_FACTOR_LARGE", "auto") != "0"

# How much to invest in an operator at ingest.
#   "full": everything that makes the chains fast -- the matching slot plan of the SELL pack, the factored form of Q -- a
#     few milliseconds per 4096-graph batch that pay back after a handful of steps (stores, cached batches, epochs).
#   "single_use": unplanned packs, no recovery of the factored form; the half-valued pack is still taken when the values
#     are exact in fp16 (it costs less than the fp32 one).
#   "auto" (default): operators built explicitly (from_csr / from_bsr3 / from_torch / stores) are "full"; a torch sparse
#     tensor handed straight to get_Batch_Wjxs / scatter -- what the reference's DataLoader yields, fresh tensors every
#     step (models/escgnn_data_classes.py:137-142) -- starts "single_use" and is promoted to "full" the SECOND time the
#     same tensor object comes back (ScatterOperator.promote): a batch seen once never pays for a plan it cannot use, a
#     batch that is reused pays once, at its second use.
INGEST_POLICY = os.environ.get("ESCAT_INGEST", "auto")


def _fp16_probe(vals: torch.Tensor) -> torch.Tensor:
    """a few thousand values: are they exact in fp16?  (0-d bool tensor on the device; the caller decides when to read it)"""
    v = vals.reshape(-1)[:: max(1, vals.numel() // 4096)]
    return (v.to(torch.float16).to(torch.float32) == v).all()


def _policy_of(form) -> str:
    """the effective policy ("full" / "single_use") for a canonical form"""
    p = getattr(form, "policy", None) or INGEST_POLICY
    return "full" if p == "auto" else p


class Sell:
    """
    SELL-64 pack of a CSR (val (nnz,), or (nnz,3): one value per channel) or BSR-3 (val (nnz,3,3)) operator; layout in
    include/escat.h (escat_sell).  Built on the GPU by libescat (escat_sell_pack_plan / escat_sell_pack_fill: row
    ordering, slice sizing, bank-conflict-aware slot planning and the coalesced fill are HIP kernels); this class only
    allocates the arrays.  Scalar and BSR-3 packs use layout 1 (LDS rows = owner positions: conflict-free vector
    stores), per-channel packs layout 0.
    Operators that share the very same rowptr / col tensors, parts, capacity and layout -- an operator and its transpose
    on a symmetric pattern, P and Q of one batch -- share ONE plan: the second pack only refills the values
    (escat_sell_pack_refill) and aliases lc / lane_meta / pos_of_row of the first.
    """

    def __init__(self, form, keep_plan: bool = False, layout=None):
        from . import _lib
        import ctypes
        lib = _lib.load()
        dev = form.rowptr.device
        if not form.rowptr.is_cuda:
            raise RuntimeError("escat: SELL packs are built on the GPU (escat_sell_pack_*); operator is on " + str(dev))
        V = 9 if form.val.dim() == 3 else (3 if form.val.dim() == 2 else 1)
        if V == 3 and form.val.shape[1] != 3:
            raise ValueError("escat: per-channel CSR values must be (nnz, 3)")
        signed = form.signed_pack() if V == 3 else None
        self.val_planes = (4 if signed is not None else 3) if V == 3 else 1
        self.layout = 1 if layout is None else int(layout)   # (override: tests of the pack itself; the kernels take layout 1)
        n, nnz = form.n, form.nnz
        if form.max_part_rows >= 0x3ff:
            raise ValueError("escat: part too large for the SELL lane record")
        if self.val_planes == 4 and 2 * sell_pad_rows(form.max_part_rows) >= (1 << 11):
            raise ValueError("escat: LDS row index does not leave room for the sign bits")
        vp = lambda t_: ctypes.c_void_p(t_.data_ptr())
        per = 576 if V == 9 else (192 if (V == 3 and signed is None) else 64)         # floats per slot of a slice
        key = (form.rowptr.data_ptr(), form.col.data_ptr(), form.rowptr._version, form.col._version,
               form.part_ptr.data_ptr(), form.n_parts, form.max_part_rows, n, nnz, self.layout, str(dev))
        shared = None
        if not keep_plan and layout is None:
            for cap_key, cand_ref in list(_PLAN_CACHE.items()):
                cand = cand_ref()
                if cand is None:                                  # its operator is gone: so is the entry
                    del _PLAN_CACHE[cap_key]
                    continue
                if cand.layout == self.layout and cand._ent is not None and cand.planned == (_policy_of(form) != "single_use") \
                        and self._same_pattern(cand, form, cap_key[:-1] == key) \
                        and cand.dmax == self._capacity_for(V, cand):
                    shared = cand
                    break
        with torch.cuda.device(dev):
            st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
            self.half_val = self.half_aux = None
            self.direct = False
            if shared is not None:
                dims = shared._dims
                self.lc, self.lane_meta, ws, self._ent = shared.lc, shared.lane_meta, shared._ws, shared._ent
                self.planned = shared.planned
                if V == 9:
                    self._try_half(lib, form, dims, ws, st)
                if signed is not None:           # |value| + sign bits: own lc = the shared rows with the signs in bits 11..13
                    self.val = torch.empty((dims.total_d + 1) * per, dtype=torch.float32, device=dev)
                    self.lc = torch.empty(dims.lc_words, dtype=torch.int32, device=dev)
                    bits = signed[2].to(torch.int32).contiguous()
                    _lib.check(lib.escat_sell_pack_refill_signed(vp(form.rowptr), vp(signed[1]), vp(bits), n, vp(form.part_ptr),
                                                                 form.n_parts, ctypes.byref(dims), vp(shared._ws), shared._ws.numel() * 4,
                                                                 vp(shared._ent), vp(shared.lc), vp(self.val), vp(self.lc), st),
                               "escat_sell_pack_refill_signed")
                    self._ent = None             # (its lc carries sign bits: not a plan others may share)
                elif V == 9 and self.half_val is None and _policy_of(form) == "single_use" and DIRECT_MODE != "0" \
                        and self._ent is not None:
                    # an operator seen once, on a plan that is already there (P's of the same batch): no value pack -- the
                    # chain reads the blocks from the canonical array through the plan (escat_sell.ent_of_slot)
                    self.val = None
                    self.direct = True
                elif self.half_val is None or HALF_MODE == "both":
                    self.val = torch.empty((dims.total_d + 1) * per, dtype=torch.float32, device=dev)
                    _lib.check(lib.escat_sell_pack_refill(vp(form.rowptr), vp(form.val), n, vp(form.part_ptr), form.n_parts, V,
                                                          ctypes.byref(dims), vp(shared._ws), shared._ws.numel() * 4,
                                                          vp(shared._ent), vp(self.val), st), "escat_sell_pack_refill")
                else:
                    self.val = None                      # the halves are the values: no fp32 copy of the pack
            else:
                ws = torch.empty(lib.escat_sell_pack_workspace_bytes(n, form.n_parts) // 4, dtype=torch.int32, device=dev)
                dims = _lib.EscatSellDims()
                _lib.check(lib.escat_sell_pack_plan(vp(form.rowptr), n, vp(form.part_ptr), form.n_parts, form.max_part_rows, V,
                                                    self.layout, vp(ws), ws.numel() * 4, ctypes.byref(dims), st),
                           "escat_sell_pack_plan")
                ns = dims.n_slices
                self.val = torch.empty((dims.total_d + 1) * per, dtype=torch.float32, device=dev)
                self.lc = torch.empty(dims.lc_words, dtype=torch.int32, device=dev)
                self.lane_meta = torch.empty((ns * 64, 3), dtype=torch.int32, device=dev)
                self._ent = torch.empty(max(dims.total_d, 1) * 64, dtype=torch.int8, device=dev) if signed is None else None
                if keep_plan:                                    # per-entry plan, for tests / diagnostics (8 B per entry)
                    self.slot_of_e = torch.empty(max(nnz, 1), dtype=torch.int32, device=dev)
                    self.lc_of_e = torch.zeros(max(nnz, 1), dtype=torch.int32, device=dev)
                if signed is not None:
                    src_val, bits = signed[1], signed[2].to(torch.int32).contiguous()
                else:
                    src_val, bits = form.val, None
                null = ctypes.c_void_p(0)
                self.planned = _policy_of(form) != "single_use"
                fill = lib.escat_sell_pack_fill if self.planned else lib.escat_sell_pack_fill_unplanned
                _lib.check(fill(vp(form.rowptr), vp(form.col), vp(src_val),
                                                    vp(bits) if bits is not None else null, n, nnz,
                                                    vp(form.part_ptr), form.n_parts, V, ctypes.byref(dims), vp(ws), ws.numel() * 4,
                                                    vp(self.val), vp(self.lc), vp(self.lane_meta),
                                                    vp(self.slot_of_e) if keep_plan else null,
                                                    vp(self.lc_of_e) if keep_plan else null,
                                                    vp(self._ent) if self._ent is not None else null, st),
                           "escat_sell_pack_fill")
                if V == 9 and self._ent is not None:
                    self._try_half(lib, form, dims, ws, st)
        self._dims = dims
        self.dmax = int(dims.dmax)
        self.slices_per_part = int(dims.slices_per_part)
        self.n_slices = ns = int(dims.n_slices)
        # views of the workspace the kernels left behind: hist[64] | w[ns] | doff[ns+1] | poff[ns+1] | perm[ns*64] | pos[n]
        o = 64
        self._hist = ws[:34]
        self.slice_w = ws[o:o + ns]; o += ns
        self.slice_doff = ws[o:o + ns + 1]; o += ns + 1
        self.slice_poff = ws[o:o + ns + 1]; o += ns + 1
        self.perm = ws[o:o + ns * 64]; o += ns * 64
        self.pos_of_row = ws[o:o + n]
        self._ws = ws
        self._pattern = (form.rowptr, form.col)          # keeps the key's tensors alive (identity is part of the key)
        self._parts = (form.part_ptr, form.n_parts, form.max_part_rows)
        self.total_d = int(dims.total_d)
        if shared is None and self._ent is not None and layout is None:
            _PLAN_CACHE.clear()                          # one pattern at a time: batches come and go
            _PLAN_CACHE[key + (self.dmax,)] = weakref.ref(self)   # weak: a dropped operator frees its pack

    @staticmethod
    def _same_pattern(cand, form, same_key: bool) -> bool:
        """does `form` have the pattern and parts the pack `cand` was planned on?  By identity of the tensors (an operator
        and its transpose; operators built on the same arrays), else by content: P and Q of one batch arrive as separate
        COO tensors (models/escgnn_data_classes.py:137-142) whose canonical patterns are equal element for element."""
        crp, ccl = cand._pattern
        if same_key and crp is form.rowptr and ccl is form.col:
            return True
        cpp = cand._parts
        if crp.shape != form.rowptr.shape or ccl.shape != form.col.shape or cpp[0].shape != form.part_ptr.shape \
                or cpp[1] != form.n_parts or cpp[2] != form.max_part_rows or crp.device != form.rowptr.device:
            return False
        return bool(torch.equal(crp, form.rowptr)) and bool(torch.equal(cpp[0], form.part_ptr)) and bool(torch.equal(ccl, form.col))

    def _try_half(self, lib, form, dims, ws, st):
        """build the half-valued form of a BSR-3 pack on the plan at hand; keeps it iff every value is exact in fp16."""
        import ctypes
        from . import _lib
        if HALF_MODE == "0" or self.layout != 1 or int(dims.dmax) > HALF_MAX_DMAX:
            return
        if form.fp16_probe is None:                  # a cheap probe before the full pass: a few thousand values settle
            form.fp16_probe = bool(_fp16_probe(form.val))                    # most operators
        if not form.fp16_probe:
            return
        dev = form.rowptr.device
        vp = lambda t_: ctypes.c_void_p(t_.data_ptr())
        hv = torch.empty(((dims.total_d + 1) * 64, 4), dtype=torch.int32, device=dev)
        ha = torch.empty((dims.total_d + 1) * 64, dtype=torch.int32, device=dev)
        flag = torch.zeros(1, dtype=torch.int32, device=dev)
        _lib.check(lib.escat_sell_pack_refill_half(vp(form.rowptr), vp(form.val), form.n, vp(form.part_ptr), form.n_parts,
                                                   ctypes.byref(dims), vp(ws), ws.numel() * 4, vp(self._ent), vp(self.lc),
                                                   vp(hv), vp(ha), vp(flag), st), "escat_sell_pack_refill_half")
        if int(flag.item()) == 0:
            self.half_val, self.half_aux = hv, ha

    @staticmethod
    def _capacity_for(V, cand) -> int:
        """the capacity escat_sell_pack_plan would choose for V values per entry on cand's row-length histogram"""
        if getattr(cand, "_hist_host", None) is None:
            cand._hist_host = list(cand._dims.row_len_hist)       # came back with the plan: no round trip here
            cand._n_rows = int(cand.pos_of_row.numel())
        caps = _SELL_DMAX[V]
        allowed = int(_SELL_TAIL_ROWS * cand._n_rows)
        for c in caps:
            if sum(cand._hist_host[c + 1:34]) <= allowed:
                return c
        return caps[-1]

    @property
    def pad_ratio(self) -> float:
        """slots stored / entries kept (diagnostic; reads the row-length histogram back)"""
        h = self._hist.cpu().tolist()
        kept = sum(min(d, self.dmax) * h[d] for d in range(34))
        return float(self.total_d * 64) / max(float(kept), 1.0)


class SellGlobal:
    """
    SELL-64 pack of a whole BSR-3 operator with GLOBAL columns (escat_sell.col32) for the multi-launch path of graphs
    that do not fit LDS: rows sorted by length inside windows of `window` rows, slices of 64, no length cap.
    """

    def __init__(self, form, window: int = 2048):
        assert form.val.dim() == 3 and window % 64 == 0
        dev = form.rowptr.device
        n, nnz = form.n, form.nnz
        rp = form.rowptr.long()
        deg = rp[1:] - rp[:-1]
        maxdeg = int(deg.max()) if n else 0
        order = torch.argsort((torch.arange(n, device=dev) // window) * (maxdeg + 1) + (maxdeg - deg), stable=True)
        n_slices = (n + 63) // 64
        perm = torch.full((n_slices * 64,), -1, dtype=torch.int32, device=dev)
        perm[:n] = order.to(torch.int32)
        slot_deg = torch.zeros(n_slices * 64, dtype=torch.int64, device=dev)
        slot_deg[:n] = deg[order]
        w = slot_deg.view(n_slices, 64).max(dim=1).values
        doff = _cumsum0(w)
        total_d = int(doff[-1])
        row_slot = torch.empty(n, dtype=torch.int64, device=dev)
        row_slot[order] = torch.arange(n, device=dev)
        row_of_e = torch.repeat_interleave(torch.arange(n, device=dev), deg)
        d_of_e = torch.arange(nnz, device=dev) - rp[row_of_e]
        sl = row_slot[row_of_e]
        drow = doff[sl // 64] + d_of_e
        lane = sl % 64
        val = torch.zeros(max(total_d, 1) * 576, dtype=torch.float32, device=dev)
        src = form.val.reshape(nnz, 9)
        for q in range(9):
            val[drow * 576 + (q // 4) * 256 + (lane * 4 + q % 4 if q < 8 else lane)] = src[:, q]
        col32 = torch.full((max(total_d, 1) * 64,), -1, dtype=torch.int32, device=dev)
        col32[drow * 64 + lane] = form.col
        self.val, self.col32, self.perm = val, col32, perm
        self.slice_doff = doff.to(torch.int32)
        self.n_slices = n_slices
        self.pad_ratio = float(total_d * 64) / max(float(nnz), 1.0)


class _Csr:
    __slots__ = ("rowptr", "col", "val", "n", "nnz", "part_ptr", "n_parts", "max_part_rows", "_max_row_nnz", "_sell", "_sell_g",
                 "_signed", "policy", "fp16_probe", "_pp_host")

    def __init__(self, rowptr, col, val, n, parts=None):
        self.rowptr, self.col, self.val, self.n = rowptr, col, val, n
        self.nnz = int(col.numel())
        if parts is None:
            parts = _closed_row_ranges(rowptr, col, n, PART_TARGET_ROWS)
        self.part_ptr, self.n_parts, self.max_part_rows = parts
        self._max_row_nnz = None        # diagnostic, on demand (a reduction + a host round trip)
        self._sell = None
        self._sell_g = None
        self._signed = None
        self.policy = None              # None: the module's INGEST_POLICY; set by the owning ScatterOperator
        self.fp16_probe = None          # _fp16_probe(val), when somebody has asked already
        self._pp_host = None            # part_ptr_host(): numpy copy for the host-side bookkeeping of mini-batch views

    def part_ptr_host(self) -> np.ndarray:
        if self._pp_host is None:
            self._pp_host = self.part_ptr.detach().cpu().numpy().astype(np.int64)
        return self._pp_host

    @property
    def max_row_nnz(self) -> int:
        if self._max_row_nnz is None:
            self._max_row_nnz = int((self.rowptr[1:] - self.rowptr[:-1]).max()) if self.n > 0 else 0
        return self._max_row_nnz

    def signed_pack(self):
        """per-channel values (nnz, 3) that differ in sign only (the signed operator of a factored vector operator):
        (col | signs << 29, |value|) -- 8 B per nonzero for the kernels (escat_sell.val_planes = 4).  None otherwise
        (then the three planes are used as they are, val_planes = 3), or with ESCAT_FACTORED_FORMAT=planes."""
        if self._signed is None:
            self._signed = False
            if self.val.dim() == 2 and self.val.shape[1] == 3 and self.n < (1 << 29) \
                    and os.environ.get("ESCAT_FACTORED_FORMAT", "") != "planes":
                mag = self.val.abs()
                if self.nnz == 0 or bool((mag == mag[:, :1]).all()):
                    neg = (torch.signbit(self.val) & (mag > 0)).to(torch.int32)
                    bits = neg[:, 0] | (neg[:, 1] << 1) | (neg[:, 2] << 2)
                    self._signed = ((self.col | (bits << 29)).contiguous(), mag[:, 0].contiguous(), bits)
        return self._signed or None

    def sell_global(self):
        """whole-operator pack with global columns for the multi-launch path (BSR-3 only)."""
        if self._sell_g is None and self.val.dim() == 3 and self.nnz > 0:   # (nnz, 3, 3) only
            self._sell_g = SellGlobal(self)
        return self._sell_g

    def sell(self, keep_plan: bool = False):
        """packed SELL-64 copy for the register-resident kernels (None if the parts are too large)."""
        if (self._sell is None or (keep_plan and not hasattr(self._sell, "slot_of_e"))) \
                and 0 < self.max_part_rows <= SELL_MAX_PART_ROWS and self.nnz > 0:
            self._sell = Sell(self, keep_plan)
        return self._sell


class ScatterOperator:
    """
    Canonical device form of a diffusion operator (P or Q).  Built once from a
    torch sparse tensor (from_torch) or from arrays already in CSR / BSR-3
    (from_csr / from_bsr3); the transposed forms are built on first backward.

    Attributes (device tensors): csr.{rowptr,col,val}; bsr.{rowptr,col,val}
    where val is (nnzb,3,3); *.part_ptr the closed row ranges.
    """

    def __init__(self, n: int, device):
        self.n = int(n)                 # scalar rows
        self.device = device
        self._csr: Optional[_Csr] = None
        self._csr_t: Optional[_Csr] = None
        self._bsr: Optional[_Csr] = None
        self._bsr_t: Optional[_Csr] = None
        self._coo = None                # (rows, cols, vals) coalesced, int64: source for lazy forms
        self._graph_ptr = None          # optional explicit parts (scalar rows), one per sample of a store
        self.bsr_fill = 0.0
        self.row_perm = None            # set by _reorder_if_large: scalar rows, new -> old (x' = x[row_perm])
        self.row_inv = None             # old -> new (y = y'[row_inv])
        self._factored = None           # see factored(): FactoredVectorOperator, or False once recovery was refused
        self.factor_err = None          # relative reconstruction error of the recovery (diagnostic)
        self._frames_hint = None        # hint_frames(): the frames the blocks were built from, when the builder knows them
        self.policy = None              # None: the module's INGEST_POLICY ("auto" = full); as_operator sets "single_use"
        self._fp16_probe = None         # values_look_fp16(), once

    def _stamp(self, form):
        if form is not None and self.policy is not None:
            form.policy = self.policy
        return form

    def promote(self):
        """a single-use operator that came back: from now on it gets the full treatment (planned packs, factored form)"""
        if self.policy != "single_use":
            return
        self.policy = "full"
        for f in (self._csr, self._csr_t, self._bsr, self._bsr_t):
            if f is not None:
                f.policy = "full"
                if f._sell is not None and not f._sell.planned:
                    f._sell = None      # rebuilt, planned, at the next chain
        self._factored = None

    def _parts_for(self, unit: int):
        """explicit parts for a form whose rows are `unit` scalar rows each (None: find the closed row ranges)"""
        if self._graph_ptr is None:
            return None
        return ScatterOperator.parts_from_graph_ptr(self._graph_ptr // unit)

    # ---------------------------------------------------------------- constructors
    @staticmethod
    def from_torch(A: torch.Tensor, allow_cpu: bool = False, graph_ptr=None) -> "ScatterOperator":
        # allow_cpu exists for host-logic tests of the canonical index structure only;
        # get_Batch_Wjxs refuses CPU tensors.
        # graph_ptr (scalar rows, [B+1]): the samples of a block-diagonal store -> one part each (see select()).
        if not A.is_cuda and not allow_cpu:
            raise RuntimeError("escat: the operator must live on a GPU (no CPU fallback in the product path)")
        if A.dim() != 2 or A.shape[0] != A.shape[1]:
            raise ValueError(f"escat: square 2-D operator expected, got {tuple(A.shape)}")
        n = A.shape[0]
        if A.layout == torch.sparse_csr:
            crow, col, val = A.crow_indices(), A.col_indices(), A.values()
            rows = torch.repeat_interleave(torch.arange(n, device=A.device), crow[1:] - crow[:-1])
            key = rows * n + col.long()
            if key.numel() > 1 and not bool((key[1:] > key[:-1]).all()):
                A = A.to_sparse_coo().coalesce()
                rows, col, val = A.indices()[0], A.indices()[1], A.values()
        elif A.layout == torch.sparse_coo:
            fast = None
            if not A.is_coalesced():
                fast = ScatterOperator._from_emission_order(A, n, graph_ptr)
                if fast is not None:
                    return fast
                A = A.coalesce()
            rows, col, val = A.indices()[0], A.indices()[1], A.values()
        else:
            raise TypeError(f"escat: unsupported operator layout {A.layout}")
        op = ScatterOperator(n, A.device)
        op._coo = (rows.long(), col.long(), val.to(torch.float32))
        if graph_ptr is not None:
            op._graph_ptr = torch.as_tensor(graph_ptr, device=A.device).to(torch.int64)
            if int(op._graph_ptr[-1]) != n or int(op._graph_ptr[0]) != 0:
                raise ValueError("escat: graph_ptr must run from 0 to the operator size")
        if n % 3 == 0 and op._coo[0].numel() > 0:
            op._try_bsr3()
        op._reorder_if_large()
        return op

    @staticmethod
    def _from_emission_order(A: torch.Tensor, n: int, graph_ptr=None):
        """
        Sort-free ingest of the two layouts the reference's pipeline actually produces (the tensors are flagged
        un-coalesced although they hold no duplicates, and torch.sparse.mm re-sorts them on every call):
          * P: indices already row-major sorted (process_pyg_data.py:775 coalesces per sample; the collate adds offsets);
          * Q: "block-emission order" (process_pyg_data.py:839-857): entries come in groups of nine, the 3x3 block of one
            P-nonzero (i, j) row-major, blocks in P's order -- never sorted by scalar row, never coalesced (:863-868).
        One elementwise pass checks the pattern (a single host sync); on success CSR / BSR-3 is formed directly.
        Returns None when the tensor is anything else (the caller then coalesces).
        """
        idx, val = A._indices(), A._values()
        nnz = int(idx.shape[1])
        if nnz == 0:
            return None
        dev = A.device
        r, c = idx[0], idx[1]
        if A.is_cuda and idx.dtype == torch.int64 and n < (1 << 31) and nnz < (1 << 31):
            fast = ScatterOperator._from_emission_order_gpu(r.contiguous(), c.contiguous(), val, n, nnz, graph_ptr, dev)
            if fast is not None:
                return fast
            # neither layout: fall through to the torch formulation below (which will say so and return None)
        if n % 3 == 0 and nnz % 9 == 0:
            r9, c9 = r.view(-1, 9), c.view(-1, 9)
            a = torch.arange(3, device=dev)
            br, bc = r9[:, 0] // 3, c9[:, 0] // 3
            ok = (r9 == (3 * br)[:, None] + a.repeat_interleave(3)[None, :]).all() & (c9 == (3 * bc)[:, None] + a.repeat(3)[None, :]).all()
            nb = n // 3
            key = br * nb + bc
            ok = ok & (key[1:] > key[:-1]).all() if key.numel() > 1 else ok
            if bool(ok):
                op = ScatterOperator(n, dev)
                if graph_ptr is not None:
                    op._graph_ptr = torch.as_tensor(graph_ptr, device=dev).to(torch.int64)
                parts = op._parts_for(3) if op._graph_ptr is not None and bool((op._graph_ptr % 3 == 0).all()) else None
                op._bsr = _Csr(_rowptr(br, nb), bc.to(torch.int32).contiguous(), val.to(torch.float32).reshape(-1, 3, 3).contiguous(), nb, parts)
                op.bsr_fill = 1.0
                op._reorder_if_large()
                return op
        key = r * n + c
        if key.numel() > 1 and not bool((key[1:] > key[:-1]).all()):
            return None
        op = ScatterOperator(n, dev)
        if graph_ptr is not None:
            op._graph_ptr = torch.as_tensor(graph_ptr, device=dev).to(torch.int64)
        op._coo = (r.long(), c.long(), val.to(torch.float32))
        if n % 3 == 0:
            op._try_bsr3()
        op._reorder_if_large()
        return op

    @staticmethod
    def _from_emission_order_gpu(r, c, val, n, nnz, graph_ptr, dev):
        """the same check + conversion as one HIP pass per candidate layout (escat_ingest_coo): no torch index ops."""
        import ctypes
        from . import _lib
        lib = _lib.load()
        vp = lambda t_: ctypes.c_void_p(t_.data_ptr())
        gp = None
        if graph_ptr is not None:
            gp = torch.as_tensor(graph_ptr, device=dev).to(torch.int64)
        with torch.cuda.device(dev):
            st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
            blocks_possible = n % 3 == 0 and nnz % 9 == 0
            if blocks_possible and gp is None:
                # Q of a batch whose P was ingested a moment ago: one pass checks the block-emission order AND that the
                # blocks sit on P's pattern; P's rowptr / col / parts are then Q's (and the packs share P's plan by identity)
                hit = _PARTS_CACHE.get("last")
                crp, ccl = (hit[1](), hit[2]()) if hit is not None else (None, None)
                if crp is not None and ccl is not None and crp.device == dev and crp.dtype == torch.int32 and ccl.dtype == torch.int32 \
                        and crp.numel() == n // 3 + 1 and ccl.numel() == nnz // 9 and hit[0][6] == PART_TARGET_ROWS \
                        and crp._version == hit[0][2] and ccl._version == hit[0][3]:
                    flags = torch.zeros(1, dtype=torch.int32, device=dev)
                    _lib.check(lib.escat_ingest_coo_like(vp(r), vp(c), nnz, n, vp(crp), vp(ccl), nnz // 9, vp(flags), st),
                               "escat_ingest_coo_like")
                    v32 = val.to(torch.float32)
                    probe = _fp16_probe(v32)             # queued behind the check: both answers in one round trip
                    f = int(flags.item())
                    if f == 0:
                        op = ScatterOperator(n, dev)
                        op._bsr = _Csr(crp, ccl, v32.reshape(-1, 3, 3).contiguous(), n // 3, hit[3])
                        op._fp16_probe = op._bsr.fp16_probe = bool(probe)
                        op.bsr_fill = 1.0
                        op._reorder_if_large()
                        return op
                    if f & 1:
                        blocks_possible = False          # not block-emission order at all
            tries = ([1] if blocks_possible else []) + [0]
            for blocks3 in tries:
                rows_out = n // 3 if blocks3 else n
                ne = nnz // 9 if blocks3 else nnz
                rowptr = torch.empty(rows_out + 1, dtype=torch.int32, device=dev)
                col32 = torch.empty(ne, dtype=torch.int32, device=dev)
                bad = torch.zeros(1, dtype=torch.int32, device=dev)
                delta = None
                if blocks3 or gp is not None:
                    _lib.check(lib.escat_ingest_coo(vp(r), vp(c), nnz, n, blocks3, vp(rowptr), vp(col32), vp(bad), st),
                               "escat_ingest_coo")
                else:                                    # the row-cover marks in the same pass (parts below)
                    delta = torch.empty(n + 1, dtype=torch.int32, device=dev)
                    _lib.check(lib.escat_ingest_coo_cover(vp(r), vp(c), nnz, n, vp(rowptr), vp(col32), vp(bad), vp(delta), st),
                               "escat_ingest_coo_cover")
                marks_parts = None
                if delta is not None:                    # the flag comes back with the cuts: one round trip
                    marks_parts = _row_blocks_from_marks(delta, rowptr, col32, n, PART_TARGET_ROWS, bad=bad)
                    if marks_parts is None:
                        continue
                elif int(bad.item()):
                    continue
                op = ScatterOperator(n, dev)
                op._graph_ptr = gp
                v32 = val.to(torch.float32)
                if blocks3:
                    parts = op._parts_for(3) if gp is not None and bool((gp % 3 == 0).all()) else None
                    op._bsr = _Csr(rowptr, col32, v32.reshape(-1, 3, 3).contiguous(), n // 3, parts)
                    op.bsr_fill = 1.0
                else:
                    parts = op._parts_for(1) if gp is not None else marks_parts
                    op._csr = _Csr(rowptr, col32, v32.contiguous(), n, parts)
                    if n % 3 == 0:
                        op._maybe_bsr3_from_csr()
                op._reorder_if_large()
                return op
        return None

    def _maybe_bsr3_from_csr(self):
        """a sorted scalar COO whose size is a multiple of 3 may still be an operator of 3x3 blocks (a coalesced Q): take
        the general route for the block detection (rare: the reference never coalesces Q)"""
        c = self._csr
        rp = c.rowptr.long()
        rows = torch.repeat_interleave(torch.arange(c.n, device=self.device), rp[1:] - rp[:-1])
        self._coo = (rows, c.col.long(), c.val)
        self._try_bsr3()

    @staticmethod
    def from_csr(rowptr, col, val, n, validate=True, parts=None) -> "ScatterOperator":
        op = ScatterOperator(n, val.device)
        rowptr, col = rowptr.to(torch.int32).contiguous(), col.to(torch.int32).contiguous()
        if validate:
            _validate(rowptr, col, n)
        op._csr = _Csr(rowptr, col, val.to(torch.float32).contiguous(), n, parts)
        if parts is None or parts[1] == 0:
            op._reorder_if_large()
        return op

    @staticmethod
    def from_bsr3(rowptr, col, blk, nb, validate=True, parts=None) -> "ScatterOperator":
        op = ScatterOperator(3 * nb, blk.device)
        rowptr, col = rowptr.to(torch.int32).contiguous(), col.to(torch.int32).contiguous()
        if validate:
            _validate(rowptr, col, nb)
        if blk.shape[1:] != (3, 3):
            raise ValueError("blk must be (nnzb, 3, 3)")
        op._bsr = _Csr(rowptr, col, blk.to(torch.float32).contiguous(), nb, parts)
        op.bsr_fill = 1.0
        if parts is None or parts[1] == 0:
            op._reorder_if_large()
        return op

    # ---------------------------------------------------------------- locality reordering of huge connected blocks
    def _reorder_if_large(self):
        """If the operator has a connected block too large for LDS (one big point cloud: BASELINE cfg3), relabel its rows
        in breadth-first order so that each step's gather of x stays inside L2.  Transparent: the chain permutes x on
        the way in and the result on the way out (esc_gnn_amd.pyg_utilities._run_chain)."""
        flag = os.environ.get("ESCAT_REORDER", "auto")
        if flag == "0" or self._graph_ptr is not None:
            return
        form = self._bsr if self._bsr is not None else self.csr()
        unit = 3 if form is self._bsr else 1
        big = form.n_parts == 0 or form.max_part_rows > REORDER_MIN_ROWS
        if form.n < 2 or form.nnz == 0 or not (big or flag == "1"):
            return
        n, dev = form.n, self.device
        perm = locality_order(form.rowptr, form.col, n)
        inv = torch.empty_like(perm)
        inv[perm] = torch.arange(n, device=dev)
        rp = form.rowptr.long()
        rows_old = torch.repeat_interleave(torch.arange(n, device=dev), rp[1:] - rp[:-1])
        r2, c2 = inv[rows_old], inv[form.col.long()]
        order = torch.argsort(r2 * n + c2)
        new = _Csr(_rowptr(r2[order], n), c2[order].to(torch.int32).contiguous(), form.val[order].contiguous(), n)
        self._coo = None
        self._csr = self._csr_t = self._bsr = self._bsr_t = None
        if unit == 3:
            self._bsr = new
            a = torch.arange(3, device=dev)
            self.row_perm = (3 * perm[:, None] + a[None, :]).reshape(-1)
            self.row_inv = (3 * inv[:, None] + a[None, :]).reshape(-1)
        else:
            self._csr = new
            self.row_perm, self.row_inv = perm, inv

    # ---------------------------------------------------------------- forms
    def _try_bsr3(self):
        rows, cols, vals = self._coo
        nb = self.n // 3
        bkey = (rows // 3) * nb + (cols // 3)
        ub, inv = torch.unique(bkey, return_inverse=True)
        nnzb = int(ub.numel())
        self.bsr_fill = rows.numel() / (9.0 * nnzb)
        if self.bsr_fill < 0.5:
            return
        blk = torch.zeros(nnzb * 9, dtype=torch.float32, device=vals.device)
        blk.index_put_((inv * 9 + (rows % 3) * 3 + (cols % 3),), vals)
        brow = ub // nb
        parts = None
        if self._graph_ptr is not None and bool((self._graph_ptr % 3 == 0).all()):
            parts = self._parts_for(3)
        self._bsr = _Csr(_rowptr(brow, nb), (ub % nb).to(torch.int32), blk.view(nnzb, 3, 3), nb, parts)

    @property
    def has_bsr3(self) -> bool:
        return self._bsr is not None

    def csr(self) -> _Csr:
        if self._csr is None:
            if self._coo is not None:
                rows, cols, vals = self._coo
                self._csr = _Csr(_rowptr(rows, self.n), cols.to(torch.int32), vals.contiguous(), self.n, self._parts_for(1))
            else:   # expand BSR-3 to scalar CSR
                b = self._bsr
                rp = b.rowptr.long()
                brow = torch.repeat_interleave(torch.arange(b.n, device=self.device), rp[1:] - rp[:-1])
                a = torch.arange(3, device=self.device)
                rows = (3 * brow)[:, None, None] + a[None, :, None] + 0 * a[None, None, :]
                cols = (3 * b.col.long())[:, None, None] + 0 * a[None, :, None] + a[None, None, :]
                key = (rows * self.n + cols).reshape(-1)
                order = torch.argsort(key)
                rows, cols, vals = rows.reshape(-1)[order], cols.reshape(-1)[order], b.val.reshape(-1)[order]
                self._csr = _Csr(_rowptr(rows, self.n), cols.to(torch.int32), vals.contiguous(), self.n)
        return self._stamp(self._csr)

    def bsr3(self) -> _Csr:
        if self._bsr is None:
            raise RuntimeError("operator has no BSR-3 form")
        return self._stamp(self._bsr)

    def _transpose_sym(self, c: _Csr):
        """A^T on A's own pattern when the pattern is symmetric (always, for P / Q of undirected graphs): one HIP kernel
        (escat_transpose_sym) instead of a sort.  None if the pattern is not symmetric (or the operator is on the CPU)."""
        if not c.val.is_cuda or c.nnz == 0:
            return None
        from . import _lib
        import ctypes
        V = 9 if c.val.dim() == 3 else (3 if c.val.dim() == 2 else 1)
        if V == 3 and c.val.shape[1] != 3:
            return None
        val_t = torch.empty_like(c.val)
        flag = torch.zeros(1, dtype=torch.int32, device=c.val.device)
        vp = lambda t_: ctypes.c_void_p(t_.data_ptr())
        with torch.cuda.device(c.val.device):
            rc = _lib.load().escat_transpose_sym(vp(c.rowptr), vp(c.col), vp(c.val), c.n, c.nnz, V, vp(val_t), vp(flag),
                                                 ctypes.c_void_p(torch.cuda.current_stream(c.val.device).cuda_stream))
        _lib.check(rc, "escat_transpose_sym")
        if int(flag.item()):
            return None
        return _Csr(c.rowptr, c.col, val_t, c.n, (c.part_ptr, c.n_parts, c.max_part_rows))

    def values_look_fp16(self) -> bool:
        """a cheap probe (a few thousand values): were the blocks rounded to fp16 in storage?  Such an operator takes the
        half-valued pack (checked in full when the pack is built), not the factored form (its blocks are no longer
        p_ij x orthogonal)."""
        b = self._bsr
        if b is None or b.nnz == 0:
            return False
        if self._fp16_probe is None:
            if b.fp16_probe is None:
                b.fp16_probe = bool(_fp16_probe(b.val))
            self._fp16_probe = b.fp16_probe
        return self._fp16_probe

    def hint_frames(self, frames: torch.Tensor, pval: Optional[torch.Tensor] = None):
        """The local frames (nb, 3, 3) this BSR-3 operator's blocks were built from, in the row order the operator was
        GIVEN in (process_pyg_data's builder knows them; the reference's files do not carry them).  factored() then
        needs no recovery: it reads the signs off the blocks in the given frames and checks every block -- which also
        serves operators the recovery kernel does not (one cloud larger than a workgroup's part: BASELINE cfg3).
        pval: P's values in the operator's entry order (exact magnitudes, so that P pairs with the factored form)."""
        if frames.dim() != 3 or frames.shape[1:] != (3, 3) or 3 * frames.shape[0] != self.n:
            raise ValueError("escat: frames must be (rows / 3, 3, 3)")
        self._frames_hint = (frames, pval)
        self._factored = None

    def _factored_from_hint(self):
        """the factored form in the frames of hint_frames(), or None if they do not reproduce the blocks"""
        frames, pval = self._frames_hint
        frames = frames.to(device=self._bsr.val.device, dtype=torch.float32).contiguous()
        if self.row_perm is not None:                   # the operator was relabelled (_reorder_if_large): follow it
            return self._factored_in_frames(frames[self.row_perm[::3] // 3], None, frames)   # (entry order changed: magnitudes from the blocks)
        return self._factored_in_frames(frames, pval, frames)

    def _factored_in_frames(self, fr, pval, frames_given):
        """fr: frames in the row order of the stored blocks, frames_given: the same in the order the operator was given in.
        val3[e, c] = sign(O_i[:, c]^T Q_e O_j[:, c]) p_e and the largest entry of O_i^T Q_e O_j - diag(val3[e]) over
        all blocks (fp64, in chunks); the signs are read off the blocks, not recomputed from the frames: a column pair
        with <O_i[:, c], O_j[:, c]> ~ 0 would otherwise flip with the rounding of the dot product."""
        b = self._bsr
        dev = b.val.device
        permn = invn = None
        if self.row_perm is not None:
            permn, invn = self.row_perm[::3] // 3, self.row_inv[::3] // 3
        rp = b.rowptr.long()
        val3 = torch.empty((b.nnz, 3), dtype=torch.float32, device=dev)
        err = torch.zeros((), dtype=torch.float64, device=dev)
        fr64 = fr.double()
        CH = 1 << 22
        for e0 in range(0, b.nnz, CH):
            e1 = min(b.nnz, e0 + CH)
            rows = torch.searchsorted(rp, torch.arange(e0, e1, device=dev), right=True) - 1
            M = fr64[rows].transpose(1, 2) @ b.val[e0:e1].double() @ fr64[b.col[e0:e1].long()]
            d = torch.diagonal(M, dim1=1, dim2=2)
            mag = pval[e0:e1].double()[:, None] if pval is not None else d.abs().mean(dim=1, keepdim=True)
            v = torch.where(d >= 0, mag, -mag)
            err = torch.maximum(err, (M - torch.diag_embed(v)).abs().amax())
            val3[e0:e1] = v.float()
        scale = float(b.val.abs().max()) if b.nnz else 0.0
        self.factor_err = float(err) / scale if scale > 0 else float("inf")
        if not self.factor_err <= FACTOR_TOL:
            return None
        signed = ScatterOperator(b.n, dev)
        signed._csr = _Csr(b.rowptr, b.col, val3, b.n, (b.part_ptr, b.n_parts, b.max_part_rows))
        signed.row_perm, signed.row_inv = permn, invn
        signed.policy = self.policy
        return FactoredVectorOperator(frames_given, signed)

    def _factored_large(self):
        """one large connected cloud (relabelled breadth-first at ingest): frames from the blocks alone across the whole
        device (esc_gnn_amd.recover_large), then the same check of every block as for hinted frames"""
        from .recover_large import recover_frames_large, workspace_bytes
        b = self._bsr
        free = torch.cuda.mem_get_info(b.val.device)[0]
        if workspace_bytes(b.nnz, b.n) > free // 2:
            return None
        G = recover_frames_large(b.rowptr, b.col, b.val)
        if G is None:
            return None
        return self._factored_in_frames(G, None, G[self.row_inv[::3] // 3].contiguous())

    def factored(self):
        """Q = D Pt D^T recovered from the blocks (include/escat.h: escat_recover_factored): a FactoredVectorOperator that
        reproduces every block to FACTOR_TOL x max|Q|, or None (fp16-rounded blocks, 'procrustes' alignment, parts too
        large, FACTOR_MODE "0").  With hint_frames() the frames are given and only checked.  Tried once per operator."""
        if self._factored is None and self._frames_hint is not None and self._bsr is not None and FACTOR_MODE != "0" \
                and self._bsr.val.is_cuda and self._bsr.nnz > 0:
            self._factored = self._factored_from_hint() or None         # (None: fall through to the recovery below)
        if self._factored is None and self.row_perm is not None and self._bsr is not None and FACTOR_MODE != "0" and LARGE_RECOVERY \
                and _policy_of(self._stamp(self._bsr)) != "single_use" and self._bsr.val.is_cuda and self._bsr.nnz > 0 \
                and not self.values_look_fp16():
            self._factored = self._factored_large() or False
        if self._factored is None:
            self._factored = False
            b = self._bsr
            if b is not None and FACTOR_MODE != "0" and _policy_of(self._stamp(b)) != "single_use" and b.val.is_cuda and b.nnz > 0 and 0 < b.max_part_rows <= SELL_MAX_PART_ROWS \
                    and self.row_perm is None and not self.values_look_fp16():
                import ctypes
                from . import _lib
                lib = _lib.load()
                dev = b.val.device
                frames = torch.empty((b.n, 3, 3), dtype=torch.float32, device=dev)
                val3 = torch.empty((b.nnz, 3), dtype=torch.float32, device=dev)
                err = torch.zeros(1, dtype=torch.float32, device=dev)
                vp = lambda t_: ctypes.c_void_p(t_.data_ptr())
                pe = b.rowptr[b.part_ptr.long()]
                max_part_nnz = int((pe[1:] - pe[:-1]).max())
                if lib.escat_recover_factored_lds_bytes(b.max_part_rows, max_part_nnz) > 160 * 1024:
                    return None
                with torch.cuda.device(dev):
                    _lib.check(lib.escat_recover_factored(vp(b.rowptr), vp(b.col), vp(b.val), b.n, b.nnz, vp(b.part_ptr), b.n_parts,
                                                          b.max_part_rows, max_part_nnz, vp(frames), vp(val3), vp(err),
                                                          ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)),
                               "escat_recover_factored")
                scale = float(b.val.abs().max())
                self.factor_err = float(err.item()) / scale if scale > 0 else float("inf")
                if self.factor_err <= FACTOR_TOL:
                    signed = ScatterOperator.from_csr(b.rowptr, b.col, val3, b.n, validate=False,
                                                      parts=(b.part_ptr, b.n_parts, b.max_part_rows))
                    self._factored = FactoredVectorOperator(frames, signed)
        return self._factored or None

    def csr_t(self) -> _Csr:
        if self._csr_t is None:
            c = self.csr()
            self._csr_t = self._transpose_sym(c)
        if self._csr_t is None:
            c = self.csr()
            rp = c.rowptr.long()
            rows = torch.repeat_interleave(torch.arange(c.n, device=self.device), rp[1:] - rp[:-1])
            order = torch.argsort(c.col.long() * c.n + rows)
            # a block-diagonal operator and its transpose have the same diagonal blocks: same parts
            self._csr_t = _Csr(_rowptr(c.col.long()[order], c.n), rows[order].to(torch.int32),
                               c.val[order].contiguous(), c.n, (c.part_ptr, c.n_parts, c.max_part_rows))
        return self._stamp(self._csr_t)

    def bsr3_t(self) -> _Csr:
        if self._bsr_t is None:
            self._bsr_t = self._transpose_sym(self.bsr3())
        if self._bsr_t is None:
            b = self.bsr3()
            rp = b.rowptr.long()
            rows = torch.repeat_interleave(torch.arange(b.n, device=self.device), rp[1:] - rp[:-1])
            order = torch.argsort(b.col.long() * b.n + rows)
            self._bsr_t = _Csr(_rowptr(b.col.long()[order], b.n), rows[order].to(torch.int32),
                               b.val[order].transpose(1, 2).contiguous(), b.n, (b.part_ptr, b.n_parts, b.max_part_rows))
        return self._stamp(self._bsr_t)


    # ---------------------------------------------------------------- batches drawn from a stored operator
    def select(self, part_ids) -> "OperatorBatch":
        """the mini-batch made of parts `part_ids` (any order, repeats allowed) of this operator -- a view: nothing is
        copied, the kernels walk the listed parts of the packed operator in place (escat_sell.part_list)."""
        return OperatorBatch(self, part_ids)

    @staticmethod
    def parts_from_graph_ptr(graph_ptr: torch.Tensor):
        """explicit parts = graphs (pass as `parts=` to from_csr / from_bsr3 when building a store: one part per sample)"""
        gp = graph_ptr.to(torch.int32).contiguous()
        return gp, int(gp.numel()) - 1, int((gp[1:] - gp[:-1]).max()) if gp.numel() > 1 else 0


class OperatorBatch:
    """
    A mini-batch drawn from a whole dataset's operator kept canonical + packed in HBM (build it once with one part
    per sample: ScatterOperator.from_csr / from_bsr3(..., parts=ScatterOperator.parts_from_graph_ptr(graph_ptr))).
    Rows of the dense operands follow the order of `part_ids`.  Accepted wherever a ScatterOperator is
    (get_Batch_Wjxs, scatter); only the register-resident kernels support it.
    """

    def __init__(self, base: ScatterOperator, part_ids):
        if base.row_perm is not None:
            raise ValueError("escat: a locality-relabelled operator (one large connected block) cannot be batched by parts")
        self.base = base
        self.device = base.device
        form = base.bsr3() if base.has_bsr3 else base.csr()
        # the bookkeeping of a view is done on the host (a numpy copy of the part ranges, made once per stored operator): a
        # 32-sample batch costs one small copy instead of a chain of device reductions with a host round trip each
        if isinstance(part_ids, OperatorBatch):         # the same parts of another stored operator: share the lists
            ids_h, ids = part_ids._ids_host, part_ids.part_ids
        elif torch.is_tensor(part_ids):
            ids_h = part_ids.detach().to("cpu", torch.int64).numpy().reshape(-1)
            ids = part_ids.detach().to(device=base.device, dtype=torch.int64).flatten()
        else:
            ids_h = np.asarray(part_ids, dtype=np.int64).reshape(-1)
            ids = torch.from_numpy(ids_h).to(base.device)
        if ids_h.size and (int(ids_h.min()) < 0 or int(ids_h.max()) >= form.n_parts):
            raise IndexError("escat: part id outside the stored operator")
        self.part_ids = ids
        self._ids_host = ids_h
        self._ref_parts = form.n_parts
        pp = form.part_ptr_host()
        self._unit = 3 if base.has_bsr3 else 1          # scalar rows per row of the reference form
        self.n = int((pp[1:] - pp[:-1])[ids_h].sum()) * self._unit
        self._lists = {}

    def same_parts(self, other: "OperatorBatch") -> bool:
        return self.part_ids is other.part_ids or np.array_equal(self._ids_host, other._ids_host)

    def lists_for(self, form):
        """(part_list, part_row0) int32 device tensors in the row units of `form`."""
        key = id(form)
        if key not in self._lists:
            if form.n_parts != self._ref_parts:
                raise RuntimeError("escat: this form of the stored operator is partitioned differently; build the store "
                                   "with explicit parts (one per sample)")
            pp = form.part_ptr_host()
            rows = (pp[1:] - pp[:-1])[self._ids_host]
            row0 = np.cumsum(rows) - rows
            both = torch.from_numpy(np.stack([self._ids_host, row0]).astype(np.int32)).to(self.device)
            self._lists[key] = (both[0], both[1])
        return self._lists[key]

    def row_index(self) -> torch.Tensor:
        """scalar rows of the stored operator, in batch order (to gather x / targets kept per dataset)."""
        base = self.base
        form = base.bsr3() if base.has_bsr3 else base.csr()
        return rows_of_ranges(form.part_ptr_host(), self._ids_host, self.device, self._unit,
                              form.part_ptr if self.part_ids.is_cuda else None, self.part_ids)


def rows_of_ranges(ptr_host: np.ndarray, ids_host: np.ndarray, device, unit: int = 1, ptr_dev=None, ids_dev=None) -> torch.Tensor:
    """concatenation of the index ranges [ptr[i], ptr[i+1]) for i in ids (each index expanded to `unit` consecutive ones):
    built on the host and uploaded when short, on the device (sizes known from the host: no round trip) when long"""
    lens = (ptr_host[1:] - ptr_host[:-1])[ids_host]
    total = int(lens.sum())
    if total <= (1 << 17) or ptr_dev is None:
        off = np.cumsum(lens) - lens
        idx = np.arange(total, dtype=np.int64) + np.repeat(ptr_host[:-1][ids_host] - off, lens)
        if unit != 1:
            idx = (unit * idx[:, None] + np.arange(unit, dtype=np.int64)[None, :]).reshape(-1)
        return torch.from_numpy(idx).to(device)
    pp = ptr_dev.long()
    start, rows = pp[:-1][ids_dev], (pp[1:] - pp[:-1])[ids_dev]
    off = torch.cumsum(rows, 0) - rows
    idx = torch.arange(total, device=device) + torch.repeat_interleave(start - off, rows, output_size=total)
    if unit != 1:
        idx = (unit * idx[:, None] + torch.arange(unit, device=device)[None, :]).reshape(-1)
    return idx


def _validate(rowptr, col, n):
    if rowptr.numel() != n + 1:
        raise ValueError("rowptr must have n+1 entries")
    rp = rowptr.long()
    if int(rp[0]) != 0 or int(rp[-1]) != col.numel() or bool((rp[1:] < rp[:-1]).any()):
        raise ValueError("rowptr is not a monotone prefix ending at nnz")
    if col.numel() and (int(col.min()) < 0 or int(col.max()) >= n):
        raise ValueError("column index out of range")


class FactoredVectorOperator:
    """
    The vector diffusion operator in factored form, Q = D Pt D^T (opt-in; the default remains BSR-3).

    process_pyg_data builds Q_ij = p_ij O_i S_ij O_j^T with S_ij = diag(sign<O_i[:,c], O_j[:,c]>) (column_dot alignment,
    data_processing/process_pyg_data.py:306-313, :492; S = I without sign enforcement).  With D = blockdiag(O_i), which
    is orthogonal, Q = D Pt D^T where Pt is P's pattern with one signed value per frame column, Pt_c[i,j] = p_ij s_ij,c.
    Hence Q^t = D Pt^t D^T for every t: a chain on Q is one rotation into the local frames, the SAME chain on Pt (a
    3-channel scalar operator with 12 B of values per nonzero instead of 36), and one rotation back per band.
    Only operators with that structure qualify (built here from the frames; not the 'procrustes' alignment, not values
    that were rounded to fp16 in storage).  Results agree with the BSR-3 chain to fp32 rounding (tests state 1e-5).

    frames (N, 3, 3): columns are the frame vectors O_i[:, c];  signed: ScatterOperator over N rows, val (nnz, 3).
    """

    def __init__(self, frames: torch.Tensor, signed):
        base = signed.base if isinstance(signed, OperatorBatch) else signed
        if frames.dim() != 3 or frames.shape[1:] != (3, 3) or frames.shape[0] != base.n:
            raise ValueError("escat: frames must be (N, 3, 3) with N = rows of the signed operator")
        self.frames = frames.to(torch.float32).contiguous()
        self.signed = signed
        self.n = 3 * signed.n              # scalar rows, like the BSR-3 operator it stands for
        self.device = frames.device
        self.row_perm = None

    def select(self, part_ids) -> "FactoredVectorOperator":
        """mini-batch of parts (samples of a store) as a view, like ScatterOperator.select; the frames stay whole (the
        kernel indexes them by the operator's own rows)."""
        if isinstance(self.signed, OperatorBatch):
            raise ValueError("escat: select() applies to a stored operator, not to a view of one")
        return FactoredVectorOperator(self.frames, self.signed.select(part_ids))

    def row_index(self) -> torch.Tensor:
        """scalar rows (3 per node) of the stored operator that a view covers, in view order (to gather x)."""
        ri = self.signed.row_index() if isinstance(self.signed, OperatorBatch) else torch.arange(self.signed.n, device=self.device)
        return (3 * ri[:, None] + torch.arange(3, device=ri.device)[None, :]).reshape(-1)

    def fits_register_path(self, F: int) -> bool:
        """True: one launch with the rotations inside the chain kernel; False (graphs too large for a workgroup): the
        rotations are two escat_rotate_frames_f32 launches around the multi-launch chain."""
        from .pyg_utilities import chain_lds_bytes, LDS_BUDGET
        base = self.signed.base if isinstance(self.signed, OperatorBatch) else self.signed
        form = base.csr()
        return (base.row_perm is None and form.n_parts > 0 and form.sell() is not None
                and max(chain_lds_bytes(form.max_part_rows, 3, F, False, False), chain_lds_bytes(form.max_part_rows, 3, F, False, True)) <= LDS_BUDGET)

    @staticmethod
    def from_frames(rowptr, col, pval, frames, enforce_sign: bool = True, parts=None) -> "FactoredVectorOperator":
        """rowptr/col/pval: P in CSR (diagonal included); frames (N,3,3) from the local PCA."""
        n = frames.shape[0]
        rp = rowptr.long()
        rows = torch.repeat_interleave(torch.arange(n, device=frames.device), rp[1:] - rp[:-1])
        if enforce_sign:
            s = torch.where((frames[rows] * frames[col.long()]).sum(1) >= 0, 1.0, -1.0)       # (nnz, 3)
            val3 = pval[:, None] * s
        else:
            val3 = pval[:, None].expand(-1, 3)
        return FactoredVectorOperator(frames, ScatterOperator.from_csr(rowptr, col, val3.contiguous(), n, validate=False, parts=parts))


def _tensor_versions(A):
    """version counters of a sparse tensor's components (in-place edits through the tensor's own values/indices bump them)"""
    try:
        if A.layout == torch.sparse_coo:
            return (A._values()._version, A._indices()._version)
        if A.layout == torch.sparse_csr:
            return (A.values()._version, A.col_indices()._version, A.crow_indices()._version)
    except Exception:
        pass
    return None


def as_operator(P_sparse):
    """torch sparse tensor (or a ScatterOperator) -> ScatterOperator, cached on the tensor object together with the
    version counters of its values / indices: an in-place edit of the operator after the first call rebuilds the pack."""
    if isinstance(P_sparse, (ScatterOperator, OperatorBatch, FactoredVectorOperator)):
        return P_sparse
    op = getattr(P_sparse, "_escat_operator", None)
    ver = _tensor_versions(P_sparse)
    if op is not None and getattr(P_sparse, "_escat_versions", ver) != ver:
        op = None
    if op is None:
        op = ScatterOperator.from_torch(P_sparse)
        if INGEST_POLICY == "auto":
            op.policy = "single_use"        # seen once so far: see INGEST_POLICY
        try:
            P_sparse._escat_operator = op
            P_sparse._escat_versions = ver
        except Exception:
            pass
    elif op.policy == "single_use" and INGEST_POLICY == "auto":
        op.promote()                        # the same tensor object again: worth a plan
    return op
