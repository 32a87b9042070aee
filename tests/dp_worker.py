"""child process of tests/test_gpu_dp_ranks.py (not collected by pytest): one rank of a DPTrainer run.
usage: python tests/dp_worker.py OUT.pt   with RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT in the environment.
All ranks share cuda:0 and talk over gloo (a one-GPU box); every rank builds the same global batch from a seed and keeps
its contiguous shard of graphs (dist_utils.graph_shard), world size 1 keeps all of it."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    out_path = sys.argv[1]
    import numpy as np
    import torch
    import torch.distributed as dist
    from esc_gnn_amd import synthetic
    from esc_gnn_amd.dist_utils import env_rank_world, graph_shard
    from esc_gnn_amd.process_pyg_data import build_PQ_batch
    from esc_gnn_amd.train_dp import DPTrainer
    rank, _, world = env_rank_world()
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    if world > 1:
        dist.init_process_group("gloo")
    B, n, k = 96, 64, 5
    clouds = synthetic.ellipsoid_clouds(B, n, seed=11)
    g0, g1 = graph_shard(B, rank, world)
    mine = clouds[g0:g1]
    Bl = g1 - g0
    pos = torch.from_numpy(mine.reshape(-1, 3)).to(dev)
    gp = torch.arange(Bl + 1, dtype=torch.int32, device=dev) * n
    b = build_PQ_batch(pos, gp, k, {"kernel": "gaussian", "gaussian_eps": 0.05})
    gid = torch.arange(Bl, device=dev).repeat_interleave(n)
    target = torch.from_numpy((mine.max(1) - mine.min(1)).max(1).astype(np.float32)).to(dev)
    pc = b["P"].csr(); rp = pc.rowptr.long()
    rows = torch.repeat_interleave(torch.arange(pc.n, device=dev), rp[1:] - rp[:-1]); cols = pc.col.long()
    edge_index = torch.stack([cols[rows != cols], rows[rows != cols]])
    torch.manual_seed(0)                                  # identical initial weights on every rank
    tr = DPTrainer(C=3, diffusion_kwargs=dict(scales_type="dyadic", J=3, include_lowpass=True), device=dev)
    # BatchNorm batch statistics are per rank under DDP (the reference does not sync them either); frozen here so that
    # the mean of the shard gradients IS the gradient of the whole batch and the runs can be compared
    tr.model.bn.eval()
    losses = [float(tr.step(pos, pos, b["P"], b["Q"], gid, Bl, target, edge_index).detach()) for _ in range(3)]
    torch.cuda.synchronize()
    torch.save({"params": [p.detach().cpu() for p in tr.model.parameters()], "losses": losses, "graphs": (g0, g1),
                "world": dist.get_world_size() if world > 1 else 1}, out_path)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
