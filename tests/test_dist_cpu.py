"""
world_size-2 gloo tests (CPU) of the data-parallel path:
  * graph shards need no exchange: each rank runs the chain (CPU oracle here, the
    kernels on GPUs) on its contiguous range of graphs of one seeded batch, and the
    concatenation equals the single-process result bit for bit;
  * the flat gradient bucket all-reduces to the mean of the per-rank gradients;
  * bench.py's aggregation = max over ranks.
"""
import os
import socket
import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from esc_gnn_amd import synthetic
from esc_gnn_amd.dist_utils import graph_shard, node_range_of_graphs, max_over_ranks, FlatGradBucket
from esc_gnn_amd.scales import plan_chain


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import chain_oracle as co
        torch.set_num_threads(1)
        b = synthetic.ellipsoid_batch(7, 40, 5, seed=77)                 # 7 graphs: uneven split 4 + 3
        plan = plan_chain(J=3)
        ev = np.asarray(plan.events, dtype=np.int32).reshape(-1, 4)
        g0, g1 = graph_shard(7, rank, world)
        r0, r1 = node_range_of_graphs(b["graph_ptr"], g0, g1)
        e0, e1 = int(b["rowptr"][r0]), int(b["rowptr"][r1])
        rp = b["rowptr"][r0:r1 + 1] - e0
        col = b["col"][e0:e1] - r0
        W = co.c_chain_csr(rp, col, b["pval"][e0:e1], b["pos"][r0:r1], ev, plan.T, plan.F)
        Wq = co.c_chain_bsr3(rp, col, b["blk"][e0:e1], b["pos"][r0:r1].reshape(-1), ev, plan.T, plan.F)
        # DP step of a tiny head on the local coefficients: flat bucket all-reduce
        torch.manual_seed(0)
        head = torch.nn.Sequential(torch.nn.Linear(3 * plan.F, 8), torch.nn.ReLU(), torch.nn.Linear(8, 1))
        bucket = FlatGradBucket(head.parameters())
        bucket.zero()
        feats = torch.from_numpy(W.reshape(W.shape[0], -1))
        loss = head(feats).pow(2).sum() / (7 * 40)                      # global-mean loss: local sum / global count
        loss.backward()
        local_grad = bucket.flat.clone()
        bucket.allreduce_mean()
        t = max_over_ranks(1.0 + rank)
        q.put((rank, (g0, g1), W, Wq, local_grad.numpy(), bucket.flat.clone().numpy(), t))
    finally:
        dist.destroy_process_group()


def test_graph_shards_and_flat_bucket(oracle_lib):
    from oracle import chain_oracle as co
    assert [graph_shard(7, r, 2) for r in range(2)] == [(0, 4), (4, 7)]
    assert [graph_shard(8, r, 3) for r in range(3)] == [(0, 3), (3, 6), (6, 8)]
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    b = synthetic.ellipsoid_batch(7, 40, 5, seed=77)
    plan = plan_chain(J=3)
    ev = np.asarray(plan.events, dtype=np.int32).reshape(-1, 4)
    W_all = co.c_chain_csr(b["rowptr"], b["col"], b["pval"], b["pos"], ev, plan.T, plan.F)
    Wq_all = co.c_chain_bsr3(b["rowptr"], b["col"], b["blk"], b["pos"].reshape(-1), ev, plan.T, plan.F)
    assert np.array_equal(np.concatenate([r[2] for r in res]), W_all)          # no exchange needed: exact
    assert np.array_equal(np.concatenate([r[3] for r in res]), Wq_all)
    mean_grad = (res[0][4] + res[1][4]) / 2
    for r in res:
        assert np.allclose(r[5], mean_grad, rtol=1e-6, atol=1e-8)               # all-reduce(sum)/world
        assert r[6] == 2.0                                                       # max over ranks of (1, 2)
