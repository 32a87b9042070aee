"""
Drop-in for ESCGNNModular._scatter (/root/reference/models/escgnn_modular.py:309-362):
orders 0/1/2 of the scattering transform on one track.

    scatter(track='scalar', x0=(N,C), P_or_Q=P, kwargs={'diffusion_kwargs': {...}}) -> (N, C, W)
    scatter(track='vector', x0=(N,d), P_or_Q=Q, kwargs=...)                         -> (N, 1, d, W)

W = 1 + F + F(F-1)/2: [x0 | first-order bank | products W1[i]*W1[j], i<j in
torch.triu_indices order (:507-516)].  No modulus anywhere (SURVEY.md finding 1).

`escat_scatter` has the reference method's exact keyword signature so that it
can replace it on the model class:  ESCGNNModular._scatter = escat_scatter
"""
import ctypes
import torch

from . import _lib
from .pyg_utilities import get_Batch_Wjxs, _ptr, _stream


class _Orders(torch.autograd.Function):
    """[x0 | W1 | pairwise products] along the last axis, one kernel (escat_scatter_orders_f32)."""

    @staticmethod
    def forward(ctx, x0, W1, second_order):
        lib = _lib.load()
        rows, F = W1.shape[0] * W1.shape[1], W1.shape[2]
        x0c, W1c = x0.contiguous(), W1.contiguous()
        Wtot = 1 + F + (F * (F - 1) // 2 if second_order else 0)
        W = torch.empty(W1.shape[0], W1.shape[1], Wtot, dtype=torch.float32, device=W1.device)
        with torch.cuda.device(W1.device):
            rc = lib.escat_scatter_orders_f32(_ptr(x0c), _ptr(W1c), rows, F, int(second_order), _ptr(W), _stream(W1.device))
        _lib.check(rc, "escat_scatter_orders_f32")
        ctx.save_for_backward(W1c)
        ctx.second_order = second_order
        return W

    @staticmethod
    def backward(ctx, dW):
        (W1c,) = ctx.saved_tensors
        lib = _lib.load()
        rows, F = W1c.shape[0] * W1c.shape[1], W1c.shape[2]
        dWc = dW.contiguous()
        dx0 = torch.empty(W1c.shape[0], W1c.shape[1], dtype=torch.float32, device=W1c.device)
        dW1 = torch.empty_like(W1c)
        with torch.cuda.device(W1c.device):
            rc = lib.escat_scatter_orders_bwd_f32(_ptr(W1c), F, _ptr(dWc), rows, F, int(ctx.second_order),
                                                  _ptr(dx0), _ptr(dW1), _stream(W1c.device))
        _lib.check(rc, "escat_scatter_orders_bwd_f32")
        return dx0, dW1, None


def scatter(*, track, x0, P_or_Q, kwargs, ablate_second_order_wavelets=False):
    if track == 'scalar':
        W1 = get_Batch_Wjxs(x=x0, P_sparse=P_or_Q, **kwargs['diffusion_kwargs'])          # (N, C, F)
        return _Orders.apply(x0, W1, not ablate_second_order_wavelets)                     # (N, C, W)
    N, d = x0.shape
    flat = x0.reshape(N * d, 1)
    W1v = get_Batch_Wjxs(x=flat, P_sparse=P_or_Q, **kwargs['diffusion_kwargs'])           # (N*d, 1, F)
    W_tot = _Orders.apply(flat, W1v, not ablate_second_order_wavelets)                     # (N*d, 1, W)
    return W_tot.view(N, d, 1, -1).permute(0, 2, 1, 3)                                     # (N, 1, d, W)


def escat_scatter(self, *, track, x0, P_or_Q, kwargs):
    """method form: assign to ESCGNNModular._scatter (reads self.ablate_second_order_wavelets)."""
    return scatter(track=track, x0=x0, P_or_Q=P_or_Q, kwargs=kwargs,
                   ablate_second_order_wavelets=bool(getattr(self, 'ablate_second_order_wavelets', False)))
