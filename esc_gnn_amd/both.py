"""
Both tracks of a batch in ONE launch (include/escat.h: escat_chain_both_f32 / _bwd_f32; csrc/escat_sell_both.hip).

ESCGNNModular.forward calls _scatter twice per batch -- the scalar track on P (/root/reference/models/escgnn_modular.py:674)
and the vector track on Q (:685) -- and each call is one get_Batch_Wjxs chain (pyg_utilities.py:350-385) over the SAME
graph: Q has one 3x3 block per nonzero of P (process_pyg_data.py:823-857), and with the column-dot alignment the blocks
are p_ij O_i S_ij O_j^T (:306-313, :492), i.e. Q = D Pt D^T with |Pt_ij| = P_ij.  For such a pair

    W_scalar, W_vector = get_Batch_Wjxs_both(x_scalar, P, x_vector, Q, **diffusion_kwargs)

returns exactly what the two separate get_Batch_Wjxs calls return (bit for bit when the factored form's magnitudes are
P's values, which `pair` arranges), from one kernel per direction in which a thread serves its row for both tracks.
Anything that does not qualify -- another channel count on the scalar track, a Q without the factored form, per-channel
scales, graphs beyond 512 nodes -- silently takes the two separate chains, so the function is a safe replacement.
"""
import ctypes
import os
from typing import Optional

import torch

from . import _lib
from .operators import ScatterOperator, OperatorBatch, FactoredVectorOperator, as_operator, SELL_MAX_PART_ROWS
from .scales import plan_chain
from . import pyg_utilities as pu

BOTH_MODE = os.environ.get("ESCAT_BOTH", "auto")       # "0": never fuse (diagnostic)
MAG_TOL = 1e-6                                        # |Pt| against P, relative to max P: the recovery's own tolerance


class BothTracksOperator:
    """P and the factored Q of one batch as one signed operator: values P_ij * s_ij,c on P's pattern + the frames."""

    def __init__(self, fop: FactoredVectorOperator, mag_err: float):
        self.fop = fop                   # FactoredVectorOperator whose magnitudes ARE P's values
        self.n = fop.signed.n            # nodes
        self.mag_err = mag_err           # max | |Pt| - P | / max P of the factored form this was made from
        self.device = fop.device

    def fits(self, plan) -> bool:
        form = self.fop.signed.csr()
        return (0 < form.max_part_rows <= SELL_MAX_PART_ROWS and form.sell() is not None and plan.T >= 1
                and not plan.per_channel and form.n * 3 * plan.F < (1 << 31))

    @staticmethod
    def pair(P_op, Q_op) -> Optional["BothTracksOperator"]:
        """the fused operator of a batch's P and Q, or None when they do not pair; tried once per (P, Q)."""
        if BOTH_MODE == "0" or not isinstance(P_op, ScatterOperator):
            return None
        cached = getattr(P_op, "_both", None)
        if cached is not None and cached[0] is Q_op:
            return cached[1]
        res = BothTracksOperator._pair(P_op, Q_op)
        P_op._both = (Q_op, res)
        return res

    @staticmethod
    def _pair(P_op, Q_op):
        if isinstance(Q_op, ScatterOperator):
            fop = Q_op.factored() if Q_op.has_bsr3 else None
        elif isinstance(Q_op, FactoredVectorOperator):
            fop = Q_op
        else:
            fop = None
        if fop is None or isinstance(fop.signed, OperatorBatch) or P_op.row_perm is not None or fop.signed.row_perm is not None:
            return None
        pc, sc = P_op.csr(), fop.signed.csr()
        pc.sell()                        # P's plan first: the fused pack is then a sign-bit refill on it
        if pc.val.dim() != 1 or sc.val.dim() != 2 or pc.n != sc.n or pc.nnz != sc.nnz or pc.nnz == 0:
            return None
        sp = sc.signed_pack()
        if sp is None:
            return None
        same = (pc.rowptr is sc.rowptr and pc.col is sc.col) or (bool(torch.equal(pc.rowptr, sc.rowptr)) and bool(torch.equal(pc.col, sc.col)))
        if not same or pc.n_parts != sc.n_parts or not (pc.part_ptr is sc.part_ptr or bool(torch.equal(pc.part_ptr, sc.part_ptr))):
            return None
        pmax = float(pc.val.max())
        if float(pc.val.min()) < 0.0 or pmax <= 0.0:
            return None                  # the scalar track multiplies by |value|
        err = float((sp[1] - pc.val).abs().max()) / pmax
        if err > MAG_TOL:
            return None
        sign = 1.0 - 2.0 * torch.stack([(sp[2] >> c) & 1 for c in range(3)], dim=1).to(torch.float32)
        signed = ScatterOperator.from_csr(pc.rowptr, pc.col, (pc.val[:, None] * sign).contiguous(), pc.n, validate=False,
                                          parts=(pc.part_ptr, pc.n_parts, pc.max_part_rows))
        return BothTracksOperator(FactoredVectorOperator(fop.frames, signed), err)


def _both_launch(bop: BothTracksOperator, plan, ins: torch.Tensor, inv: torch.Tensor, backward: bool, view=None):
    """one call into the C ABI.  forward: ins, inv (n, 3) -> (n, 3, F) each; backward: (n, 3, F) each -> (n, 3) each.
    view: an OperatorBatch of the fused signed operator (parts of a stored operator, walked in place: escat_sell.part_list)."""
    lib = _lib.load()
    signed = bop.fop.signed
    form = signed.csr_t() if backward else signed.csr()
    sell = form.sell()
    sp = form.signed_pack()
    if sell is None or sp is None or sell.val_planes != 4:
        raise _lib.EscatError("escat: the both-tracks chain needs the sign-bit pack of the fused operator")
    n, F = (view.n if view is not None else form.n), plan.F
    dev = ins.device
    shape = (n, 3) if backward else (n, 3, F)
    outs = torch.empty(shape, dtype=torch.float32, device=dev)
    outv = torch.empty(shape, dtype=torch.float32, device=dev)
    if n == 0:
        return outs, outv
    ev = pu._events_array(plan.events)
    frames = bop.fop.frames
    pl, pr, nl = 0, 0, 0
    if view is not None:
        plist, prow0 = view.lists_for(form)
        pl, pr, nl = plist.data_ptr(), prow0.data_ptr(), int(plist.numel())
    sell_struct = _lib.EscatSell(sell.val.data_ptr(), sell.lc.data_ptr(), sell.lane_meta.data_ptr(), sell.dmax, sell.n_slices,
                                 sell.slices_per_part, 4, 0, 0, 0, pl, pr, nl, 0, frames.data_ptr(), sell.layout,
                                 sell.pos_of_row.data_ptr(), 0, 0, 0)
    fn = lib.escat_chain_both_bwd_f32 if backward else lib.escat_chain_both_f32
    with torch.cuda.device(dev):
        rc = fn(pu._ptr(form.rowptr), pu._ptr(sp[0]), pu._ptr(sp[1]), form.n, form.nnz, pu._ptr(ins), pu._ptr(inv), plan.T,
                ctypes.cast(ev, ctypes.c_void_p), len(plan.events), F, pu._ptr(outs), pu._ptr(outv),
                pu._ptr(form.part_ptr), form.n_parts, form.max_part_rows, ctypes.cast(ctypes.pointer(sell_struct), ctypes.c_void_p),
                pu._stream(dev))
    _lib.check(rc, "escat_chain_both_bwd_f32" if backward else "escat_chain_both_f32")
    return outs, outv


class _BothBank(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xs, xv, bop, plan, view=None):
        ctx.bop, ctx.plan, ctx.view = bop, plan, view
        return _both_launch(bop, plan, xs.contiguous(), xv.contiguous(), False, view)

    @staticmethod
    def backward(ctx, gWs, gWv):
        ns, nv = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        if not (ns or nv):
            return None, None, None, None, None
        gs = gWs.contiguous() if gWs is not None else None
        gv = gWv.contiguous() if gWv is not None else None
        if gs is None:
            gs = torch.zeros_like(gv)
        if gv is None:
            gv = torch.zeros_like(gs)
        dxs, dxv = _both_launch(ctx.bop, ctx.plan, gs, gv, True, ctx.view)
        return (dxs if ns else None), (dxv if nv else None), None, None, None


def _fusable(x_s, x_v, plan) -> bool:
    return (torch.is_tensor(x_s) and torch.is_tensor(x_v) and x_s.is_cuda and x_v.is_cuda and x_s.dtype == torch.float32
            and x_v.dtype == torch.float32 and x_s.dim() == 2 and x_s.shape[1] == 3 and x_v.dim() == 2 and x_v.shape[1] == 1
            and x_v.shape[0] == 3 * x_s.shape[0] and plan.T >= 1 and not plan.per_channel
            and not os.environ.get("ESCAT_CHAIN_PATH", ""))


def bank_both(x_s, P_op, x_v, Q_op, plan):
    """(W_scalar (N, 3, F), W_vector (3N, 1, F)) from one launch, or None when the pair does not qualify."""
    if not _fusable(x_s, x_v, plan):
        return None
    view = None
    if isinstance(P_op, OperatorBatch):
        # mini-batches drawn from a stored operator (ScatterOperator.select / FactoredVectorOperator.select): the stored P
        # and Q pair once, the batch is the same list of parts of the fused operator
        Qv = Q_op.signed if isinstance(Q_op, FactoredVectorOperator) else Q_op
        if not isinstance(Qv, OperatorBatch) or not P_op.same_parts(Qv):
            return None
        Qb = Qv.base
        if isinstance(Q_op, FactoredVectorOperator):
            Qb = getattr(Qv.base, "_as_factored", None)
            if Qb is None or Qb.frames is not Q_op.frames:
                Qb = Qv.base._as_factored = FactoredVectorOperator(Q_op.frames, Qv.base)
        bop = BothTracksOperator.pair(P_op.base, Qb)
        if bop is None or not bop.fits(plan):
            return None
        view = getattr(P_op, "_both_view", None)
        if view is None or view.base is not bop.fop.signed:
            view = P_op._both_view = bop.fop.signed.select(P_op)
        if view.n != x_s.shape[0]:
            return None
    else:
        bop = BothTracksOperator.pair(P_op, Q_op)
        if bop is None or bop.n != x_s.shape[0] or not bop.fits(plan):
            return None
    try:
        Ws, Wv = _BothBank.apply(x_s, x_v.view(-1, 3), bop, plan, view)
    except _lib.EscatError as e:         # an event table the fused kernel does not serve (ESCAT_ERANGE)
        if e.code == _lib.ESCAT_ERANGE:
            return None
        raise
    return Ws, Wv.view(x_v.shape[0], 1, plan.F)


def get_Batch_Wjxs_both(x_scalar: torch.Tensor, P_sparse, x_vector: torch.Tensor, Q_sparse, scales_type='dyadic',
                        diffusion_scales=None, J: int = 5, include_lowpass: bool = True, filter_stack_dim: int = -1,
                        rescale_filtrations: bool = False):
    """get_Batch_Wjxs (pyg_utilities.py:283-560 of the reference, same arguments) for BOTH tracks of one batch:
    returns (get_Batch_Wjxs(x_scalar, P_sparse, ...), get_Batch_Wjxs(x_vector, Q_sparse, ...)).  x_scalar (N, C),
    x_vector (3N, 1) -- the flattening ESCGNNModular._scatter hands over (models/escgnn_modular.py:345).  One launch per
    direction when C = 3 and Q is P's factored companion; the two separate chains otherwise."""
    kw = dict(scales_type=scales_type, diffusion_scales=diffusion_scales, J=J, include_lowpass=include_lowpass,
              filter_stack_dim=filter_stack_dim, rescale_filtrations=rescale_filtrations)
    if not rescale_filtrations and filter_stack_dim in (-1, 2):
        plan = plan_chain(scales_type=scales_type, diffusion_scales=diffusion_scales, J=J, include_lowpass=include_lowpass)
        if torch.is_tensor(x_scalar) and torch.is_tensor(x_vector) and x_scalar.dim() == 2 and x_vector.dim() == 2:
            P_op, Q_op = as_operator(P_sparse), as_operator(Q_sparse)
            if isinstance(P_op, (ScatterOperator, OperatorBatch)) and P_op.n == x_scalar.shape[0] and getattr(Q_op, "n", -1) == x_vector.shape[0]:
                res = bank_both(x_scalar, P_op, x_vector, Q_op, plan)
                if res is not None:
                    return res
    return pu.get_Batch_Wjxs(x_scalar, P_sparse, **kw), pu.get_Batch_Wjxs(x_vector, Q_sparse, **kw)
