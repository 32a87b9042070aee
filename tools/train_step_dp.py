#!/usr/bin/env python3
"""BASELINE cfg4: full ESc-GNN-style training step, data-parallel over GPUs (one process per GPU, RCCL all-reduce of
one flat gradient bucket).  Operators are built on the GPU (build_PQ_batch), scattering runs through libescat.so.
  python tools/train_step_dp.py --steps 20                          (1 GPU)
  python -m torch.distributed.run --nproc-per-node N tools/train_step_dp.py --gpus N ...
Prints one JSON line on rank 0: graphs/s (all ranks), ms/step, parameter count."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--graphs", type=int, default=4096, help="graphs per GPU per step")
    ap.add_argument("--nodes", type=int, default=128)
    ap.add_argument("--k", type=int, default=5)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--factored", action="store_true", help="vector operator in factored form (FactoredVectorOperator)")
    ap.add_argument("--amp", action="store_true", help="bf16 autocast for the (plain PyTorch) head; scattering stays fp32")
    args = ap.parse_args()
    import numpy as np, torch, torch.distributed as dist
    from esc_gnn_amd import synthetic
    from esc_gnn_amd.dist_utils import env_rank_world, max_over_ranks
    from esc_gnn_amd.process_pyg_data import build_PQ_batch
    from esc_gnn_amd.train_dp import DPTrainer
    rank, local_rank, world = env_rank_world()
    rehearsal = os.environ.get("ESCAT_BENCH_REHEARSAL", "") == "1"    # ranks share devices, gloo: script check only
    di = local_rank % max(torch.cuda.device_count(), 1) if rehearsal else local_rank
    torch.cuda.set_device(di)
    dev = torch.device("cuda", di)
    if world > 1:
        dist.init_process_group("gloo") if rehearsal else dist.init_process_group("nccl", device_id=dev)
    clouds = synthetic.ellipsoid_clouds(args.graphs, args.nodes, seed=457892 + rank)
    pos = torch.from_numpy(clouds.reshape(-1, 3)).to(dev)
    gp = torch.arange(args.graphs + 1, dtype=torch.int32, device=dev) * args.nodes
    b = build_PQ_batch(pos, gp, args.k, {"kernel": "gaussian", "gaussian_eps": 0.05}, max_graph_nodes=args.nodes,
                       factored=args.factored)
    Q = b["Q_factored"] if args.factored else b["Q"]
    graph_id = torch.arange(args.graphs, device=dev).repeat_interleave(args.nodes)
    # edge_index of the batch (no self loops), as batch.edge_index in the reference: (src, dst) = (col, row) of P's pattern
    pc = b["P"].csr(); rp = pc.rowptr.long()
    rows = torch.repeat_interleave(torch.arange(pc.n, device=dev), rp[1:] - rp[:-1]); cols = pc.col.long()
    off = rows != cols
    edge_index = torch.stack([cols[off], rows[off]])
    # synthetic graph-level target: the cloud's diameter-like extent (max coordinate range)
    target = torch.from_numpy((clouds.max(1) - clouds.min(1)).max(1).astype(np.float32)).to(dev)
    torch.manual_seed(0)          # identical initial weights on every rank
    tr = DPTrainer(C=3, diffusion_kwargs=dict(scales_type="dyadic", J=4, include_lowpass=True), device=dev, amp_bf16=args.amp)
    b["P"].csr().sell()
    (Q.signed.csr() if args.factored else Q.bsr3()).sell()
    losses = []
    for _ in range(args.warmup):
        tr.step(pos, pos, b["P"], Q, graph_id, args.graphs, target, edge_index)
    if world > 1: dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        losses.append(tr.step(pos, pos, b["P"], Q, graph_id, args.graphs, target, edge_index))
    if world > 1: dist.barrier()
    torch.cuda.synchronize()
    dt = max_over_ranks(time.perf_counter() - t0, None if rehearsal else dev)
    if rank == 0:
        print(json.dumps({"workload": f"cfg4: DP training step, {args.graphs} graphs x {args.nodes} nodes per GPU, k={args.k}, J=4",
                          "head_precision": "bf16 autocast" if args.amp else "fp32", "vector_operator": "factored" if args.factored else "bsr3", "n_gpus": world, "graphs_per_s": args.graphs * world * args.steps / dt, "ms_per_step": dt / args.steps * 1e3,
                          "params": tr.n_params(), "loss_first": float(losses[0]), "loss_last": float(losses[-1])}))
    if world > 1: dist.destroy_process_group()

if __name__ == "__main__":
    main()
