"""
Minimal data-parallel ESc-GNN training step (BASELINE cfg4).

What a step does, per rank (one process per GPU, graphs sharded, RCCL only for the gradients):
  1. orders-0/1/2 scattering of both tracks on the rank's batch -- esc_gnn_amd.scatter (the hot path; like the
     reference, x and the operators carry no grad, so the chain sits outside the autograd graph,
     training/train.py:303-321, models/escgnn_modular.py:674-685);
  2. the vector track's invariants for track mixing -- esc_gnn_amd.invariants.get_vector_invariants
     (models/escgnn_modular.py:403-504: per-wavelet norms + pooled neighbour cosines), then a small head on
     [scalar coefficients | invariants] (~1e5 parameters like the reference's mixers + readout), graph-mean readout,
     MSE loss;
  3. backward into ONE flat fp32 gradient bucket, one all-reduce (sum) / world, AdamW step.
The reference gets 3 from Accelerate/DDP (training/train.py:408, :483-492); the heads themselves are plain PyTorch
and out of the hot-path scope.
"""
from typing import Dict
import torch
import torch.nn as nn

from .dist_utils import FlatGradBucket
from .scatter import scatter
from .invariants import get_vector_invariants


class ScatteringHead(nn.Module):
    def __init__(self, n_scalar_feats: int, n_vector_feats: int, hidden: int = 256):
        super().__init__()
        self.bn = nn.BatchNorm1d(n_scalar_feats + n_vector_feats)
        self.mlp = nn.Sequential(nn.Linear(n_scalar_feats + n_vector_feats, hidden), nn.SiLU(),
                                 nn.Linear(hidden, hidden), nn.SiLU(), nn.Linear(hidden, 1))

    def forward(self, Ws: torch.Tensor, inv: torch.Tensor, graph_id: torch.Tensor, n_graphs: int) -> torch.Tensor:
        N = Ws.shape[0]
        feats = torch.cat([Ws.reshape(N, -1), inv], dim=1)                             # (N, C*W + W*(1+m))
        y = self.mlp(self.bn(feats)).squeeze(-1).float()
        out = torch.zeros(n_graphs, device=y.device, dtype=y.dtype).index_add_(0, graph_id, y)
        cnt = torch.zeros(n_graphs, device=y.device, dtype=y.dtype).index_add_(0, graph_id, torch.ones_like(y))
        return out / cnt.clamp(min=1)


class DPTrainer:
    POOL_STATS = ('percentiles', 'min', 'max', 'mean')        # the reference's default neighbor_kwargs (:431-434)

    def __init__(self, C: int, diffusion_kwargs: Dict, lr: float = 1e-3, device=None, amp_bf16: bool = False):
        from .scales import plan_chain
        from .invariants import quantile_levels
        F = plan_chain(**{k: v for k, v in diffusion_kwargs.items()}).F
        W = 1 + F + F * (F - 1) // 2
        m = 3 + len(quantile_levels(0.2))
        self.kw = {"diffusion_kwargs": diffusion_kwargs}
        self.amp_bf16 = amp_bf16          # head GEMMs in bf16 (Accelerate's mixed_precision='bf16'); scattering stays fp32
        self.model = ScatteringHead(C * W, W * (1 + m)).to(device)
        self.bucket = FlatGradBucket(self.model.parameters())
        self.opt = torch.optim.AdamW(self.model.parameters(), lr=lr)

    def n_params(self) -> int:
        return self.bucket.flat.numel()

    def step(self, x_s, x_v, P, Q, graph_id, n_graphs, target, edge_index) -> float:
        with torch.no_grad():
            Ws = scatter(track="scalar", x0=x_s, P_or_Q=P, kwargs=self.kw)               # (N, C, W)
            Wv = scatter(track="vector", x0=x_v, P_or_Q=Q, kwargs=self.kw)               # (N, 1, 3, W)
            inv = get_vector_invariants(Wv.permute(0, 3, 1, 2).contiguous(), edge_index, self.POOL_STATS, 0.2)   # (N, W', 1, d) in
        self.bucket.zero()
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=self.amp_bf16):
            pred = self.model(Ws, inv, graph_id, n_graphs)
        loss = torch.nn.functional.mse_loss(pred.float(), target)
        loss.backward()                       # grads land in the flat bucket (p.grad are views of it)
        self.bucket.allreduce_mean()          # the step's only collective
        self.opt.step()
        return loss
