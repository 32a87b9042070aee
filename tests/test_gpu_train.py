"""GPU test of the minimal data-parallel training step (cfg4 in small, world size 1: the all-reduce is the identity;
the multi-rank bucket logic is covered on CPU by tests/test_dist_cpu.py)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_dp_train_step_learns():
    from esc_gnn_amd import synthetic
    from esc_gnn_amd.process_pyg_data import build_PQ_batch
    from esc_gnn_amd.train_dp import DPTrainer
    dev = torch.device("cuda:0")
    B, n = 64, 64
    clouds = synthetic.ellipsoid_clouds(B, n, seed=3)
    pos = torch.from_numpy(clouds.reshape(-1, 3)).to(dev)
    gp = torch.arange(B + 1, dtype=torch.int32, device=dev) * n
    b = build_PQ_batch(pos, gp, 5, {"kernel": "gaussian", "gaussian_eps": 0.05})
    gid = torch.arange(B, device=dev).repeat_interleave(n)
    target = torch.from_numpy((clouds.max(1) - clouds.min(1)).max(1).astype(np.float32)).to(dev)
    torch.manual_seed(0)
    tr = DPTrainer(C=3, diffusion_kwargs=dict(scales_type="dyadic", J=3, include_lowpass=True), device=dev)
    assert 5e4 < tr.n_params() < 2e5
    pc = b["P"].csr(); rp = pc.rowptr.long()
    rows = torch.repeat_interleave(torch.arange(pc.n, device=dev), rp[1:] - rp[:-1]); cols = pc.col.long()
    edge_index = torch.stack([cols[rows != cols], rows[rows != cols]])               # (src, dst), no self loops
    losses = [float(tr.step(pos, pos, b["P"], b["Q"], gid, B, target, edge_index).detach()) for _ in range(25)]
    assert all(np.isfinite(losses)) and losses[-1] < 0.2 * losses[0]
    assert all(p.grad.data_ptr() >= tr.bucket.flat.data_ptr() for p in tr.model.parameters())   # grads alias the bucket
