"""
Drop-in for ESCGNNModular._get_vector_invariants (/root/reference/models/escgnn_modular.py:403-504): per node and
wavelet, the norm of the vector-track coefficient and pooled statistics (mean / max / min / var / lower quantiles) of
its cosine similarity with the coefficients of the node's in-neighbours.

    get_vector_invariants(Wv (N, W', 1, d), edge_index (2, E), pool_stats=[...], quantiles_stride=0.2) -> (N, W' + W'*m)
    escat_get_vector_invariants(self, Wv, batch)      # method form: ESCGNNModular._get_vector_invariants = ...

The reference pools with torch_scatter and computes the quantiles in a Python loop over nodes (one torch.kthvalue per
node and level, :484-497); here one kernel per direction does all statistics (csrc/escat_pool.hip).  Output block order
as in the reference: norms | mean | max | min | var | quantile levels ascending (whatever the order in pool_stats).
torch_scatter is absent from this image: its conventions (mean of an empty segment = 0, max/min of an empty segment = 0)
are restated from its documentation -- parity unpinned on that point, see oracle/invariants_oracle.py.
"""
import ctypes
from typing import Sequence
import torch

from . import _lib
from .pyg_utilities import _ptr, _stream

MAX_DEG = 64
_ST = {"mean": 1, "max": 2, "min": 4, "var": 8}


def quantile_levels(stride: float):
    """the reference's levels {s, 2s, ...} < 1 as Python floats of fp32 values (:472-476, :484)"""
    if not (0.0 < stride < 1.0):
        return []
    return torch.arange(stride, 1.0, stride, dtype=torch.float32).tolist()


def _k_table(levels) -> torch.Tensor:
    """k = int((cnt - 1) * p) for cnt = 0..MAX_DEG (:494), tabulated on the host exactly as the reference computes it"""
    return torch.tensor([[int((cnt - 1) * p) if cnt > 0 else 0 for cnt in range(MAX_DEG + 1)] for p in levels],
                        dtype=torch.int32).reshape(len(levels), MAX_DEG + 1)


class EdgeGroups:
    """edge_index grouped by destination (stable: edge order kept, :478) and, for the backward pass, by source.
    Built once per edge_index tensor and cached on it (the graph of a batch does not change between epochs)."""

    def __init__(self, edge_index: torch.Tensor, n_nodes: int):
        src, dst = edge_index[0], edge_index[1]
        dev = edge_index.device
        self.perm = torch.argsort(dst, stable=True)
        src_s, dst_s = src[self.perm], dst[self.perm]
        counts = torch.bincount(dst, minlength=n_nodes)
        self.max_deg = int(counts.max()) if counts.numel() else 0
        self.seg_ptr = torch.zeros(n_nodes + 1, dtype=torch.int32, device=dev)
        self.seg_ptr[1:] = torch.cumsum(counts, 0)
        self.src_s, self.dst_s = src_s.to(torch.int32).contiguous(), dst_s.to(torch.int32).contiguous()
        self.src_l, self.dst_l = src_s, dst_s                   # int64, for the torch path
        self.out_edge = torch.argsort(src_s, stable=True).to(torch.int32).contiguous()
        self.out_ptr = torch.zeros(n_nodes + 1, dtype=torch.int32, device=dev)
        self.out_ptr[1:] = torch.cumsum(torch.bincount(src, minlength=n_nodes), 0)
        self.n_nodes, self.n_edges = n_nodes, int(src.numel())


def edge_groups(edge_index: torch.Tensor, n_nodes: int) -> EdgeGroups:
    eg = getattr(edge_index, "_escat_groups", None)
    if eg is None or eg.n_nodes != n_nodes or eg.n_edges != int(edge_index.shape[1]) \
            or getattr(edge_index, "_escat_groups_version", -1) != edge_index._version:      # edited in place since
        eg = EdgeGroups(edge_index, n_nodes)
        try:
            edge_index._escat_groups = eg
            edge_index._escat_groups_version = edge_index._version
        except Exception:
            pass
    return eg


class _NeighborPool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, vals, seg_ptr, n_nodes, stats, ktab, max_deg):
        lib = _lib.load()
        vals = vals.contiguous()
        W = vals.shape[1]
        n_levels = int(ktab.shape[0])
        m = bin(stats).count("1") + n_levels
        na = (1 if stats & 2 else 0) + (1 if stats & 4 else 0) + n_levels
        out = torch.empty(n_nodes, m * W, dtype=torch.float32, device=vals.device)
        arg = torch.empty(n_nodes, max(na, 1) * W, dtype=torch.uint8, device=vals.device)
        with torch.cuda.device(vals.device):
            rc = lib.escat_neighbor_pool_f32(_ptr(vals), _ptr(seg_ptr), n_nodes, W, max_deg, stats, n_levels, _ptr(ktab),
                                             _ptr(out), _ptr(arg), _stream(vals.device))
        _lib.check(rc, "escat_neighbor_pool_f32")
        ctx.save_for_backward(vals, seg_ptr, arg)
        ctx.meta = (n_nodes, stats, n_levels)
        return out

    @staticmethod
    def backward(ctx, gout):
        vals, seg_ptr, arg = ctx.saved_tensors
        n_nodes, stats, n_levels = ctx.meta
        lib = _lib.load()
        gvals = torch.empty_like(vals)
        with torch.cuda.device(vals.device):
            rc = lib.escat_neighbor_pool_bwd_f32(_ptr(vals), _ptr(seg_ptr), n_nodes, vals.shape[1], stats, n_levels,
                                                 _ptr(gout.contiguous()), _ptr(arg), _ptr(gvals), _stream(vals.device))
        _lib.check(rc, "escat_neighbor_pool_bwd_f32")
        return gvals, None, None, None, None, None


class _VectorInvariants(torch.autograd.Function):
    """norms + pooled neighbour cosines, fused (escat_vector_invariants[_bwd]_f32)"""

    @staticmethod
    def forward(ctx, v, eg, stats, ktab):
        lib = _lib.load()
        v = v.contiguous()
        N, W, d = v.shape
        n_levels = int(ktab.shape[0])
        m = bin(stats).count("1") + n_levels
        na = (1 if stats & 2 else 0) + (1 if stats & 4 else 0) + n_levels
        out = torch.empty(N, (1 + m) * W, dtype=torch.float32, device=v.device)
        arg = torch.empty(N, max(na, 1) * W, dtype=torch.uint8, device=v.device)
        with torch.cuda.device(v.device):
            rc = lib.escat_vector_invariants_f32(_ptr(v), _ptr(eg.src_s), _ptr(eg.seg_ptr), N, W, d, eg.max_deg, stats, n_levels,
                                                 _ptr(ktab), _ptr(out), _ptr(arg), _stream(v.device))
        _lib.check(rc, "escat_vector_invariants_f32")
        ctx.save_for_backward(v, arg)
        ctx.eg, ctx.meta = eg, (stats, n_levels)
        return out

    @staticmethod
    def backward(ctx, gout):
        v, arg = ctx.saved_tensors
        eg = ctx.eg
        stats, n_levels = ctx.meta
        lib = _lib.load()
        N, W, d = v.shape
        dv = torch.empty_like(v)
        gedge = torch.empty(max(eg.n_edges, 1) * W, dtype=torch.float32, device=v.device)     # two-pass form: no atomics
        with torch.cuda.device(v.device):
            rc = lib.escat_vector_invariants_bwd_f32(_ptr(v), _ptr(eg.src_s), _ptr(eg.seg_ptr), N, W, d, eg.max_deg, stats, n_levels,
                                                     _ptr(gout.contiguous()), _ptr(arg), _ptr(dv), _ptr(eg.dst_s),
                                                     _ptr(eg.out_ptr), _ptr(eg.out_edge), ctypes.c_void_p(gedge.data_ptr()),
                                                     _stream(v.device))
        _lib.check(rc, "escat_vector_invariants_bwd_f32")
        return dv, None, None, None


def get_vector_invariants(Wv: torch.Tensor, edge_index: torch.Tensor,
                          pool_stats: Sequence[str] = ('percentiles', 'min', 'max', 'mean'),
                          quantiles_stride: float = 0.2) -> torch.Tensor:
    if not Wv.is_cuda:
        raise RuntimeError("escat: Wv must live on a GPU; the MI355X path has no CPU fallback")
    N, Wp = Wv.shape[0], Wv.shape[1]
    v = Wv.squeeze(2)                                              # (N, W', d)
    stats = 0
    for name, bit in _ST.items():
        if name in pool_stats:
            stats |= bit
    levels = quantile_levels(float(quantiles_stride)) if ('quantiles' in pool_stats or 'percentiles' in pool_stats) else []
    if stats == 0 and not levels:
        return torch.cat([torch.linalg.norm(v, dim=-1), torch.zeros(N, 0, device=Wv.device)], dim=1)
    eg = edge_groups(edge_index, N)
    if eg.max_deg > MAX_DEG:
        raise NotImplementedError(f"escat: a node has {eg.max_deg} incoming edges; the pooling kernel holds at most {MAX_DEG}")
    ktab = _k_table(levels).to(Wv.device)
    if v.shape[-1] in (2, 3) and v.dtype == torch.float32:
        # fused: cosines, sort and statistics per (node, wavelet) thread; no (E, W', d) intermediates
        return _VectorInvariants.apply(v, eg, stats, ktab)
    norms = torch.linalg.norm(v, dim=-1)                           # (N, W')
    v_norm = torch.clamp(torch.linalg.norm(v, dim=-1, keepdim=True), min=1e-8)        # (N, W', 1)
    denom = v_norm[eg.dst_l].squeeze(-1) * v_norm[eg.src_l].squeeze(-1)               # (E, W')
    cos_sorted = (v[eg.dst_l] * v[eg.src_l]).sum(dim=-1) / denom                      # (E, W')
    pooled = _NeighborPool.apply(cos_sorted.to(torch.float32), eg.seg_ptr, N, stats, ktab, eg.max_deg)
    return torch.cat([norms, pooled.to(norms.dtype)], dim=1)


def escat_get_vector_invariants(self, Wv, batch):
    """method form: assign to ESCGNNModular._get_vector_invariants (reads self.neighbor_kwargs, batch.edge_index)."""
    kw = getattr(self, 'neighbor_kwargs', {}) or {}
    return get_vector_invariants(Wv, batch.edge_index, kw.get('pool_stats', ['percentiles', 'min', 'max', 'mean']),
                                 float(kw.get('quantiles_stride', 0.2)))
