"""
Drop-in for the reference's pyg_utilities.get_Batch_Wjxs
(/root/reference/pyg_utilities.py:283-560): same name, same arguments, same
return layout, same error behaviour -- computed by libescat.so's gfx950 kernels.

    from esc_gnn_amd.pyg_utilities import get_Batch_Wjxs

There is no CPU path: tensors must live on the GPU and the HIP library must be
built, otherwise this raises.
"""
import ctypes
import functools
import os
from typing import Optional
import torch

from . import _lib
from .operators import ScatterOperator, OperatorBatch, FactoredVectorOperator, as_operator, SELL_MAX_PART_ROWS
from .scales import ChainPlan, plan_chain

LDS_BUDGET = 160 * 1024
MAX_C = 8
ESCAT_EVENT_ABS = 256          # include/escat.h


@functools.lru_cache(maxsize=512)
def _events_array(events):
    """ctypes copy of a plan's event tuple (cached: plans are few and immutable; filling the array costs ~1 us an event)"""
    arr = (_lib.EscatEvent * max(len(events), 1))()
    for k, (step, slot, sign, chan) in enumerate(events):
        arr[k].step, arr[k].slot, arr[k].sign, arr[k].chan = step, slot, sign, chan
    return arr


def _stream(device):
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _ptr(t: Optional[torch.Tensor]):
    return ctypes.c_void_p(t.data_ptr() if t is not None and t.numel() else 0) if t is not None else ctypes.c_void_p(0)


def chain_lds_bytes(max_part_rows: int, C: int, F: int, bsr3: bool, backward: bool = False, orders: int = 0) -> int:
    """LDS a register-resident chain workgroup needs at least: two ping-pong vectors of two copies each, the output
    slab, and (forward) the staged input and the row -> position table.  (Backward adds a second slab when it fits -- decided in the library.)"""
    from .operators import sell_pad_rows
    CP = 4 if bsr3 else (1 if C == 1 else 2 if C == 2 else 4 if C <= 4 else 8)
    cw = 3 if bsr3 else C
    slab = (max_part_rows * cw * (F + (1 if orders and not backward else 0)) + 3) & ~3
    stage = 0 if backward else (max_part_rows * cw + 3) & ~3
    pos = 0 if backward else (max_part_rows + 3) & ~3          # forward: the row -> position table (escat_sell_lds_bytes)
    return (2 * 2 * sell_pad_rows(max_part_rows) * CP + slab + stage + pos) * 4


class NoFusedOrders(Exception):
    """the launch cannot fuse the orders epilogue (not on the register-resident path): the caller uses two kernels"""


def _chain_launch(op, plan_events, T, F, inp: torch.Tensor, C: int, backward: bool, orders: int = 0, frames=None) -> torch.Tensor:
    """one call into the C ABI for <= MAX_C channels.  inp: fwd (n,C) / bwd (n,C,F); returns fwd (n,C,F) / bwd (n,C).
    op: a ScatterOperator, or an OperatorBatch (parts of a stored operator, walked in place).
    orders (forward): 1 / 2 = return (n, C, Wtot) = [x | bank (| pair products)] from the chain's own epilogue."""
    lib = _lib.load()
    dev = inp.device
    view = op if isinstance(op, OperatorBatch) else None
    if view is not None:
        op = view.base
    n = view.n if view is not None else op.n
    ev = _events_array(plan_events)
    use_bsr = C == 1 and op.has_bsr3
    form = (op.bsr3_t() if backward else op.bsr3()) if use_bsr else (op.csr_t() if backward else op.csr())
    rows_per = 3 if use_bsr else 1
    Wout = F if orders == 0 else 1 + F + (F * (F - 1) // 2 if orders == 2 else 0)
    out = torch.empty((n, C) if backward else (n, C, Wout), dtype=torch.float32, device=dev)
    lds = chain_lds_bytes(form.max_part_rows, C, F, use_bsr, backward, orders)
    path = os.environ.get("ESCAT_CHAIN_PATH", "")
    fused = form.n_parts > 0 and lds <= LDS_BUDGET and not path.startswith("multi")
    if not use_bsr and form.val.dim() == 2 and (not fused or form.sell() is None):
        fused = False                   # one value per channel: register-resident path or multi-launch, nothing between
    ws = None
    if not fused and T > 0 and view is None:
        ws = torch.empty(lib.escat_chain_workspace_bytes(n, C) // 4, dtype=torch.float32, device=dev)
    ws_bytes = ws.numel() * 4 if ws is not None else 0
    sell = form.sell() if fused else None
    if sell is not None and (form.max_part_rows * rows_per * C > 0xffff or path == "stream"):
        sell = None
    if orders and sell is None:
        raise NoFusedOrders()
    if orders:
        # the fused epilogue expands whole output rows in an LDS tile made of the two vector buffers (csrc/escat.hip,
        # chain_epilogue_orders): a row that does not fit (tiny parts, many bands) takes the two-kernel form
        from .operators import sell_pad_rows
        CPt = 4 if use_bsr else (1 if C == 1 else 2 if C == 2 else 4 if C <= 4 else 8)
        if Wout > 2 * 2 * sell_pad_rows(form.max_part_rows) * CPt:
            raise NoFusedOrders()
    if frames is not None and (use_bsr or form.val.dim() != 2):
        raise ValueError("escat: frames go with a CSR operator that has one value per channel")
    per_channel_vals = not use_bsr and form.val.dim() == 2
    if per_channel_vals and C != 3:
        raise NotImplementedError("escat: a CSR operator with one value per channel takes C = 3 input channels")
    if frames is not None and sell is None:
        raise NotImplementedError("escat: in-kernel frame rotation needs the register-resident path")
    if view is not None and sell is None:
        raise NotImplementedError("escat: an OperatorBatch needs the register-resident path (parts of <= 512 rows that fit LDS)")
    if view is not None and n == 0:
        return out
    sell_ref = ctypes.c_void_p(0)
    if sell is not None:
        pl, pr, nl = 0, 0, 0
        if view is not None:
            plist, prow0 = view.lists_for(form)
            pl, pr, nl = plist.data_ptr(), prow0.data_ptr(), int(plist.numel())
        half = getattr(sell, "half_val", None) is not None
        sell_struct = _lib.EscatSell(sell.val.data_ptr() if sell.val is not None else 0, sell.lc.data_ptr(), sell.lane_meta.data_ptr(),
                                     sell.dmax, sell.n_slices, sell.slices_per_part, sell.val_planes, 0, 0, 0, pl, pr, nl, orders,
                                     frames.data_ptr() if frames is not None else 0, sell.layout, sell.pos_of_row.data_ptr(),
                                     sell.half_val.data_ptr() if half else 0, sell.half_aux.data_ptr() if half else 0,
                                     sell._ent.data_ptr() if getattr(sell, "direct", False) else 0)
        sell_ref = ctypes.cast(ctypes.pointer(sell_struct), ctypes.c_void_p)
    elif not fused and use_bsr and T > 0 and path == "multi_packed":
        # experimental (opt-in): whole-operator SELL pack with global columns for the multi-launch path.  Measured on
        # cfg3 it does not beat the lane-group step kernel (0.61 vs 0.53 ms/step on a locality-sorted cloud, equal on
        # random node order where the x gather dominates), so the default multi-launch path stays unpacked.
        sg = form.sell_global()
        sell_struct = _lib.EscatSell(sg.val.data_ptr(), 0, 0, 0, sg.n_slices, 0, 0,
                                     sg.col32.data_ptr(), sg.perm.data_ptr(), sg.slice_doff.data_ptr(), 0, 0, 0, 0, 0, 0, 0)
        sell_ref = ctypes.cast(ctypes.pointer(sell_struct), ctypes.c_void_p)
    elif per_channel_vals:
        # no pack (graphs too large for the register path): a descriptor that only says how val / col are laid out
        sell_struct = _lib.EscatSell(0, 0, 0, 0, 0, 0, 4 if form.signed_pack() is not None else 3, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0)
        sell_ref = ctypes.cast(ctypes.pointer(sell_struct), ctypes.c_void_p)
    col_t, val_t = form.col, form.val
    if per_channel_vals and (sell.val_planes == 4 if sell is not None else form.signed_pack() is not None):
        col_t, val_t = form.signed_pack()[0], form.signed_pack()[1]      # column | signs << 29, |value|
    with torch.cuda.device(dev):
        st = _stream(dev)
        if use_bsr:
            fn = lib.escat_chain_bsr3_bwd_f32 if backward else lib.escat_chain_bsr3_f32
            rc = fn(_ptr(form.rowptr), _ptr(form.col), _ptr(form.val), form.n, form.nnz, _ptr(inp), T,
                    ctypes.cast(ev, ctypes.c_void_p), len(plan_events), F, _ptr(out),
                    _ptr(form.part_ptr), form.n_parts, form.max_part_rows, sell_ref, _ptr(ws), ws_bytes, st)
        else:
            fn = lib.escat_chain_csr_bwd_f32 if backward else lib.escat_chain_csr_f32
            rc = fn(_ptr(form.rowptr), _ptr(col_t), _ptr(val_t), form.n, form.nnz, _ptr(inp), C, T,
                    ctypes.cast(ev, ctypes.c_void_p), len(plan_events), F, _ptr(out),
                    _ptr(form.part_ptr), form.n_parts, form.max_part_rows, sell_ref, _ptr(ws), ws_bytes, st)
    _lib.check(rc, fn.__name__ if hasattr(fn, "__name__") else "escat chain")
    return out


def _run_chain(op, plan: ChainPlan, inp: torch.Tensor, backward: bool) -> torch.Tensor:
    base = op.base if isinstance(op, OperatorBatch) else op
    if base.row_perm is not None:       # operator relabelled for locality (operators._reorder_if_large): x' = x[perm], y = y'[inv]
        return _run_chain_cols(op, plan, inp[base.row_perm], backward)[base.row_inv]
    return _run_chain_cols(op, plan, inp, backward)


def _run_chain_cols(op, plan: ChainPlan, inp: torch.Tensor, backward: bool) -> torch.Tensor:
    C = inp.shape[1]
    if C <= MAX_C:
        return _chain_launch(op, plan.events, plan.T, plan.F, inp.contiguous(), C, backward)
    # wider feature matrices: column chunks of <= MAX_C channels (per-channel events are re-based per chunk)
    outs = []
    for c0 in range(0, C, MAX_C):
        c1 = min(C, c0 + MAX_C)
        evs = tuple((s, f, sg, ch if ch < 0 else ch - c0) for (s, f, sg, ch) in plan.events if ch < 0 or c0 <= ch < c1)
        outs.append(_chain_launch(op, evs, plan.T, plan.F, inp[:, c0:c1].contiguous(), c1 - c0, backward))
    return torch.cat(outs, dim=1)


class _WaveletBank(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, op, plan):
        ctx.op, ctx.plan = op, plan
        return _run_chain(op, plan, x, backward=False)

    @staticmethod
    def backward(ctx, gW):
        if not ctx.needs_input_grad[0]:
            return None, None, None
        return _run_chain(ctx.op, ctx.plan, gW.contiguous(), backward=True), None, None


class _FactoredBank(torch.autograd.Function):
    """bank of the factored vector operator in one launch: x (N, 3) global coordinates -> (N, 3, F) global."""

    @staticmethod
    def forward(ctx, x, fop, plan):
        ctx.fop, ctx.plan = fop, plan
        return _chain_launch(fop.signed, plan.events, plan.T, plan.F, x.contiguous(), 3, False, frames=fop.frames)

    @staticmethod
    def backward(ctx, gW):
        if not ctx.needs_input_grad[0]:
            return None, None, None
        fop, plan = ctx.fop, ctx.plan
        return _chain_launch(fop.signed, plan.events, plan.T, plan.F, gW.contiguous(), 3, True, frames=fop.frames), None, None


class _RotateFrames(torch.autograd.Function):
    """Y[n, :, f] = O_n X[n, :, f]  (transpose: O_n^T), O (N,3,3), X (N,3,F): escat_rotate_frames_f32."""

    @staticmethod
    def forward(ctx, frames, X, transpose):
        ctx.frames, ctx.transpose = frames, transpose
        return _rotate(frames, X, transpose)

    @staticmethod
    def backward(ctx, gY):
        return None, _rotate(ctx.frames, gY.contiguous(), not ctx.transpose), None


def _rotate(frames, X, transpose):
    lib = _lib.load()
    Xc = X.contiguous()
    Y = torch.empty_like(Xc)
    with torch.cuda.device(Xc.device):
        rc = lib.escat_rotate_frames_f32(_ptr(frames), _ptr(Xc), _ptr(Y), Xc.shape[0], Xc.shape[2], int(bool(transpose)),
                                         _stream(Xc.device))
    _lib.check(rc, "escat_rotate_frames_f32")
    return Y


def rotate_frames(frames: torch.Tensor, X: torch.Tensor, transpose: bool) -> torch.Tensor:
    return _RotateFrames.apply(frames, X, transpose)


def vector_route(op, x, F: int, events=()):
    """the operator the vector track (x of shape (3N, 1) on a BSR-3 operator) really runs on: the factored form (recovered
    from the blocks, or checked against them in the builder's frames: ScatterOperator.hint_frames) when the operator has
    one and its graphs fit the register-resident path or the operator takes the multi-launch path anyway, else the
    operator itself.
    Not when a chain path is forced (ESCAT_CHAIN_PATH: a diagnostic of the BSR-3 kernels), and not for a |.| bank (the
    modulus must be taken in global coordinates; the factored kernel's slab is in the local frames)."""
    if os.environ.get("ESCAT_CHAIN_PATH", "") or any(ev[2] not in (1, -1) for ev in events):     # (+-1 + ESCAT_EVENT_ABS)
        return op
    if not (x.dim() == 2 and x.shape[1] == 1 and op.n == x.shape[0]):
        return op
    if isinstance(op, ScatterOperator) and op.has_bsr3:
        f = op.factored()
        if f is not None and f.fits_register_path(F):
            return f
        # a cloud too large for the register-resident path (BASELINE cfg3): both forms run one step kernel per power, and
        # the factored one streams 8 instead of 40 B per nonzero (2 M points, J = 6: 8.4 instead of 20.8 ms per chain)
        if f is not None and (op.row_perm is not None or op.bsr3().n_parts == 0 or op.bsr3().max_part_rows > SELL_MAX_PART_ROWS):
            return f
    elif isinstance(op, OperatorBatch) and op.base.has_bsr3:
        f = op.base.factored()
        if f is not None and f.fits_register_path(F):
            if getattr(op, "_factored_view", None) is None:
                op._factored_view = f.select(op)               # the same parts of the signed operator, the frames whole
            return op._factored_view
    return op


def bank(x: torch.Tensor, op, plan: ChainPlan) -> torch.Tensor:
    """(rows, C) -> (rows, C, F) for a chain plan on a canonical operator (differentiable w.r.t. x)."""
    op = vector_route(op, x, plan.F, plan.events)
    if isinstance(op, FactoredVectorOperator):
        if x.shape[1] != 1:
            raise ValueError("escat: the factored vector operator takes the flattened vector feature (3N, 1)")
        # Q^t = D Pt^t D^T: into the local frames, the chain on the signed scalar operator, every band back out --
        # all inside the chain kernel (escat_sell.frames) when the graphs fit the register-resident path; otherwise
        # (or with ESCAT_FACTORED_ROTATE=separate) two rotation launches around the chain
        if os.environ.get("ESCAT_FACTORED_ROTATE", "") == "separate" or not op.fits_register_path(plan.F):
            u = rotate_frames(op.frames, x.view(-1, 3, 1), transpose=True).view(-1, 3)
            return rotate_frames(op.frames, _WaveletBank.apply(u, op.signed, plan), transpose=False).view(x.shape[0], 1, -1)
        return _FactoredBank.apply(x.view(-1, 3), op, plan).view(x.shape[0], 1, -1)
    return _WaveletBank.apply(x, op, plan)


def abs_events(events):
    """the event tuple with ESCAT_EVENT_ABS added to the sign of the last event of every (slot, channel): the chain then
    keeps |value| in each slot, in the same pass (include/escat.h)."""
    last = {}
    for k, (_, slot, _, chan) in enumerate(events):
        last[(slot, chan)] = k
    closing = set(last.values())
    return tuple((st, sl, sg + ESCAT_EVENT_ABS if k in closing else sg, ch) for k, (st, sl, sg, ch) in enumerate(events))


def get_Batch_Wjxs(
    x: torch.Tensor,
    P_sparse,
    scales_type='dyadic',
    diffusion_scales=None,
    J: int = 5,
    include_lowpass: bool = True,
    filter_stack_dim: int = -1,
    rescale_filtrations: bool = False,
    *,
    apply_abs: bool = False,
) -> torch.Tensor:
    """
    Diffusion-wavelet filter bank W_j x = (P^{a_{j-1}} - P^{a_j}) x (+ low-pass)
    by recursive sparse products; see pyg_utilities.py:283-347 of the reference
    for the argument semantics, which are unchanged.  `P_sparse` may be a torch
    sparse COO/CSR tensor (canonicalised once and cached on the tensor) or a
    prebuilt esc_gnn_amd.ScatterOperator.  Returns (N, C, F), fp32, on x.device,
    differentiable w.r.t. x.

    apply_abs (keyword-only, NOT a reference argument, default off): return |W_j x| -- the modulus of LEGS-style callers
    (models/comparisons/legs.py:299-302), applied by the chain kernel where it finalises each slot.  The drop-in path of
    ESCGNNModular never sets it: the reference's _scatter keeps signed coefficients (models/escgnn_modular.py:326-341).
    """
    plan = plan_chain(scales_type=scales_type, diffusion_scales=diffusion_scales, J=J,
                      include_lowpass=include_lowpass)
    if apply_abs and not (torch.is_grad_enabled() and torch.is_tensor(x) and x.requires_grad):
        plan = ChainPlan(plan.T, plan.F, abs_events(plan.events), plan.per_channel, plan.n_channels)
        apply_abs = False
    if rescale_filtrations:
        raise NotImplementedError("Rescaling filtrations within 'get_Batch_Wjxs' has been deprecated.")
    if not torch.is_tensor(x) or x.dim() != 2:
        raise ValueError("escat: x must be a 2-D (N, C) tensor")
    if not x.is_cuda:
        raise RuntimeError("escat: x must live on a GPU; the MI355X path has no CPU fallback "
                           "(use the reference implementation on CPU)")
    if x.dtype != torch.float32:
        raise TypeError(f"escat: x must be float32 (torch.sparse.mm in the reference requires matching fp32), got {x.dtype}")
    op = as_operator(P_sparse)
    if op.n != x.shape[0]:
        raise RuntimeError(f"escat: size mismatch, operator is {op.n}x{op.n} but x has {x.shape[0]} rows")
    if plan.per_channel and plan.n_channels != x.shape[1]:
        raise IndexError("index out of range: per-channel diffusion_scales rows != channels of x")
    W = bank(x, op, plan)
    if apply_abs:                       # with a gradient to x: the signed bank from the kernel, |.| by autograd
        W = W.abs()
    if not plan.per_channel and plan.T >= 0 and filter_stack_dim not in (-1, 2) \
            and not (diffusion_scales is None and scales_type is None):
        W = W.movedim(-1, filter_stack_dim)
    return W
