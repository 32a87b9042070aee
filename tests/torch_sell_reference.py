"""
Round 1's torch-op construction of the SELL-64 pack (layout 0), kept as the REFERENCE the GPU pack
(esc_gnn_amd/csrc/escat_pack.hip: escat_sell_pack_plan / _fill) is compared with bit for bit.  Test infrastructure only.
"""
import torch
from esc_gnn_amd import _lib
from esc_gnn_amd.operators import _SELL_DMAX, _SELL_TAIL_ROWS, _BSR3_PIECE_ORDER, _cumsum0


class TorchSell:
    """
    SELL-64 pack of a CSR (val (nnz,), or (nnz,3): one value per channel) or BSR-3 (val (nnz,3,3)) operator; layout in
    include/escat.h (escat_sell).  Built with torch index ops on the operator's device;
    the slot order of every row's entries and the LDS vector copy each gather reads come
    from libescat's escat_sell_plan_slots (bank-conflict-aware, one wave per slice).
    """

    def __init__(self, form):
        dev = form.rowptr.device
        V = 9 if form.val.dim() == 3 else (3 if form.val.dim() == 2 else 1)
        if V == 3 and form.val.shape[1] != 3:
            raise ValueError("escat: per-channel CSR values must be (nnz, 3)")
        signed = form.signed_pack() if V == 3 else None
        self.val_planes = (4 if signed is not None else 3) if V == 3 else 1
        caps = _SELL_DMAX[V]
        deg_all = (form.rowptr[1:] - form.rowptr[:-1]).long()
        hist = torch.bincount(deg_all.clamp(max=caps[-1] + 1), minlength=caps[-1] + 2).cumsum(0)
        allowed = int(_SELL_TAIL_ROWS * form.n)
        self.dmax = next((c for c in caps if form.n - int(hist[c]) <= allowed), caps[-1])
        dmax = self.dmax
        n, nnz = form.n, form.nnz
        rp = form.rowptr.long()
        deg = rp[1:] - rp[:-1]
        pp = form.part_ptr.long()
        n_parts = form.n_parts
        part_rows = pp[1:] - pp[:-1]
        part_of_row = torch.repeat_interleave(torch.arange(n_parts, device=dev), part_rows)
        dcl = deg.clamp(max=dmax)
        order = torch.argsort(part_of_row * (dmax + 1) + (dmax - dcl), stable=True)     # rows in lane order
        # every part owns nsl slices (= waves of the workgroup that serves it); trailing ones may be empty
        nsl = 4 if form.max_part_rows <= 256 else 8
        self.slices_per_part = nsl
        part_slices = torch.full((n_parts,), nsl, dtype=torch.int64, device=dev)
        psp = _cumsum0(part_slices)
        n_slices = n_parts * nsl
        # sorted position i (row order[i], part part_of_row[i]) -> its lane
        slot0 = psp[part_of_row] * 64 + (torch.arange(n, device=dev) - pp[part_of_row])
        perm = torch.full((n_slices * 64,), -1, dtype=torch.int32, device=dev)
        perm[slot0] = (order - pp[part_of_row]).to(torch.int32)
        slot_cnt = torch.zeros(n_slices * 64, dtype=torch.int64, device=dev)
        slot_cnt[slot0] = dcl[order]
        w = slot_cnt.view(n_slices, 64).max(dim=1).values
        doff, poff = _cumsum0(w), _cumsum0(((w + 3) // 4) * 2)      # LDS-row pairs cover w rounded up to 4 slots
        row_slot0 = torch.empty(n, dtype=torch.int64, device=dev)
        row_slot0[order] = slot0
        slice_part = torch.repeat_interleave(torch.arange(n_parts, device=dev), part_slices)
        # slot (column of the slice) and LDS row of every kept entry
        slot_of_e = torch.full((max(nnz, 1),), -1, dtype=torch.int32, device=dev)
        lc_of_e = torch.zeros(max(nnz, 1), dtype=torch.int32, device=dev)
        if not form.rowptr.is_cuda:
            raise RuntimeError("escat: SELL packs are planned on the GPU (escat_sell_plan_slots); operator is on " + str(dev))
        sp32 = slice_part.to(torch.int32)
        with torch.cuda.device(dev):
            rc = _lib.load().escat_sell_plan_slots(
                form.rowptr.data_ptr(), form.col.data_ptr(), form.part_ptr.data_ptr(), sp32.data_ptr(), perm.data_ptr(),
                n_slices, dmax, slot_of_e.data_ptr(), lc_of_e.data_ptr(), torch.cuda.current_stream(dev).cuda_stream)
        _lib.check(rc, "escat_sell_plan_slots")
        e_k = torch.nonzero(slot_of_e[:nnz] >= 0).flatten()
        row_of_e = torch.repeat_interleave(torch.arange(n, device=dev), deg)
        r_k = row_of_e[e_k]
        sl = row_slot0[r_k]
        s_k, lane = sl // 64, sl % 64
        drow = doff[s_k] + slot_of_e[e_k].long()
        total_d, total_p = int(doff[-1]), int(poff[-1])
        if V == 9:
            val = torch.zeros((total_d + 1) * 576, dtype=torch.float32, device=dev)     # + one padding slot (empty slices read it)
            src = form.val.reshape(nnz, 9)[e_k]
            for sp, q in enumerate(_BSR3_PIECE_ORDER):      # stored position sp holds block entry q (row-major)
                dst = drow * 576 + (sp // 4) * 256 + (lane * 4 + sp % 4 if sp < 8 else lane)
                val[dst] = src[:, q]
        elif V == 3 and signed is not None:          # one plane of magnitudes; the signs go beside the LDS rows below
            val = torch.zeros((total_d + 1) * 64, dtype=torch.float32, device=dev)
            val[drow * 64 + lane] = signed[1][e_k]
        elif V == 3:
            val = torch.zeros((total_d + 1) * 192, dtype=torch.float32, device=dev)
            src = form.val[e_k]
            for c in range(3):
                val[(drow * 3 + c) * 64 + lane] = src[:, c]
        else:
            val = torch.zeros((total_d + 1) * 64, dtype=torch.float32, device=dev)
            val[drow * 64 + lane] = form.val[e_k]
        # LDS rows, one int per (slice, slot, lane); slots past a row's end point at the part's all-zero LDS row
        zc = part_rows[slice_part]                                                        # per slice
        slice_of_drow = torch.repeat_interleave(torch.arange(n_slices, device=dev), w)
        lcu = zc[slice_of_drow].repeat_interleave(64)                                     # (total_d*64,)
        lcu[drow * 64 + lane] = lc_of_e[e_k].long()
        if V == 3 and signed is not None:
            if int(lc_of_e.max() if nnz else 0) >= (1 << 11):
                raise ValueError("escat: LDS row index does not leave room for the sign bits")
            lcu[drow * 64 + lane] |= signed[2][e_k].long() << 11
        npair = ((w + 3) // 4) * 2
        slice_of_pair = torch.repeat_interleave(torch.arange(n_slices, device=dev), npair)
        j2 = torch.arange(total_p, device=dev) - poff[slice_of_pair]
        has_lo = (2 * j2) < w[slice_of_pair]
        has_hi = (2 * j2 + 1) < w[slice_of_pair]
        d_first = doff[slice_of_pair]
        lanes = torch.arange(64, device=dev)
        zrow = zc[slice_of_pair][:, None].expand(-1, 64)

        def rows_of(d_abs, has):
            d_safe = torch.where(has, d_abs, d_first).clamp(max=max(total_d - 1, 0))
            got = lcu[(d_safe[:, None] * 64 + lanes[None, :]).reshape(-1)].view(-1, 64)
            return torch.where(has[:, None], got, zrow)
        if total_p and total_d:
            lo, hi = rows_of(d_first + 2 * j2, has_lo), rows_of(d_first + 2 * j2 + 1, has_hi)
        else:
            lo = hi = torch.zeros((0, 64), dtype=torch.int64, device=dev)
        lc = torch.cat([(lo | (hi << 16)).reshape(-1), torch.zeros(64, dtype=torch.int64, device=dev)])   # + padding pair
        # per-lane start-up record: {row | tail_n << 10, tail_e0, slice word}
        lm = torch.zeros((n_slices * 64, 3), dtype=torch.int64, device=dev)
        lm[:, 0] = 0x3ff
        tail_n = (deg - dmax).clamp(min=0)
        rows_idx = torch.arange(n, device=dev)
        lm[row_slot0, 0] = (rows_idx - pp[part_of_row]) | (tail_n << 10)
        lm[row_slot0, 1] = rp[:-1] + dmax
        sw = lm[:, 2].view(n_slices, 64)
        sw[:, 0], sw[:, 1], sw[:, 2] = doff[:-1], poff[:-1], w
        sw[:, 3], sw[:, 4] = pp[slice_part], part_rows[slice_part]
        if int(tail_n.max() if n else 0) >= (1 << 21) or form.max_part_rows >= 0x3ff:
            raise ValueError("escat: part too large for the SELL lane record")
        self.val = val
        self.lc = lc.to(torch.int32)          # bit pattern of the packed uint32 pairs (values < 2^26 here)
        self.perm = perm
        self.lane_meta = lm.to(torch.int32).contiguous()
        self.slice_doff = doff.to(torch.int32)
        self.slice_poff = poff.to(torch.int32)
        self.n_slices = n_slices
        self.pad_ratio = float(total_d * 64) / max(float(dcl.sum()), 1.0)
        self.slot_of_e, self.lc_of_e = slot_of_e, lc_of_e      # kept for tests / diagnostics (8 B per entry)


