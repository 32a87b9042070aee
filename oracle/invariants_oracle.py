"""
TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  CPU restatement of ESCGNNModular._get_vector_invariants
(/root/reference/models/escgnn_modular.py:403-504), line for line, with torch_scatter.scatter replaced by its documented
semantics (torch_scatter is absent from this image: parity unpinned on the empty-segment conventions):
    scatter(src, index, dim=0, dim_size=N, reduce='mean'|'max'|'min') -- rows of `src` reduced per index; segments
    without entries give 0.
The quantile branch keeps the reference's Python loop over nodes and torch.kthvalue.
"""
from typing import Sequence
import torch


def _scatter(src, index, N, reduce):
    out = torch.zeros((N,) + tuple(src.shape[1:]), dtype=src.dtype)
    idx = index.view(-1, *([1] * (src.dim() - 1))).expand_as(src)
    if reduce == 'mean':
        out = out.scatter_add(0, idx, src)
        cnt = torch.bincount(index, minlength=N).clamp(min=1).to(src.dtype)
        return out / cnt.view(-1, *([1] * (src.dim() - 1)))
    red = 'amax' if reduce == 'max' else 'amin'
    out = out.scatter_reduce(0, idx, src, reduce=red, include_self=False)
    return out


def ref_vector_invariants(Wv, edge_index, pool_stats: Sequence[str] = ('percentiles', 'min', 'max', 'mean'),
                          quantiles_stride: float = 0.2):
    N, Wp = Wv.shape[0], Wv.shape[1]
    v = Wv.squeeze(2)                                                   # :416
    norms = torch.linalg.norm(v, dim=-1)                                # :417
    src, dst = edge_index[0], edge_index[1]
    v_norm = torch.clamp(torch.linalg.norm(v, dim=-1, keepdim=True), min=1e-8)
    denom = v_norm[dst].squeeze(-1) * v_norm[src].squeeze(-1)
    cos_all = (v[dst] * v[src]).sum(dim=-1) / denom                     # :429
    stats = pool_stats
    pooled = []
    pooled_mean = None
    if ('mean' in stats) or ('var' in stats):
        pooled_mean = _scatter(cos_all, dst, N, 'mean')
        if 'mean' in stats:
            pooled.append(pooled_mean)
    if 'max' in stats:
        pooled.append(_scatter(cos_all, dst, N, 'max'))
    if 'min' in stats:
        pooled.append(_scatter(cos_all, dst, N, 'min'))
    if 'var' in stats:
        msq = _scatter(cos_all * cos_all, dst, N, 'mean')
        pooled.append(torch.clamp(msq - pooled_mean.pow(2), min=0.0))
    if ('quantiles' in stats) or ('percentiles' in stats):
        stride = float(quantiles_stride)
        if 0.0 < stride < 1.0:
            levels = torch.arange(stride, 1.0, stride)
            if levels.numel() > 0:
                perm = torch.argsort(dst, stable=True)
                dst_sorted = dst[perm]
                cos_sorted = cos_all[perm]
                counts = torch.bincount(dst_sorted, minlength=N)
                starts = torch.cumsum(counts, dim=0) - counts
                for p in levels.tolist():
                    rows = []
                    for n in range(N):
                        cnt = int(counts[n].item())
                        if cnt <= 0:
                            rows.append(torch.zeros(Wp, dtype=cos_all.dtype))
                            continue
                        seg = cos_sorted[starts[n]: starts[n] + cnt, :]
                        k = int((cnt - 1) * p)
                        rows.append(torch.kthvalue(seg, k=k + 1, dim=0).values)
                    pooled.append(torch.stack(rows))
    pooled_t = torch.cat(pooled, dim=1) if pooled else torch.zeros(N, 0)
    return torch.cat([norms, pooled_t], dim=1)
