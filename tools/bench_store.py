#!/usr/bin/env python3
"""mini-batches drawn from a dataset's operators kept packed in HBM (OperatorBatch views): throughput of P+Q
scattering fwd+bwd per shuffled batch, including drawing the batch (index tensors, gathering x), vs the same batch size
processed as one contiguous operator.  usage: python tools/bench_store.py [--store 8192 --batch 4096]"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("--store", type=int, default=8192); ap.add_argument("--batch", type=int, default=4096)
ap.add_argument("--nodes", type=int, default=500); ap.add_argument("--k", type=int, default=10); ap.add_argument("--J", type=int, default=4)
ap.add_argument("--steps", type=int, default=20)
args = ap.parse_args()
import numpy as np
from esc_gnn_amd import synthetic
b = synthetic.ellipsoid_batch(args.store, args.nodes, args.k, seed=457893, workers=8)
import torch
from esc_gnn_amd import get_Batch_Wjxs, ScatterOperator
dev = torch.device("cuda:0")
t = lambda a: torch.from_numpy(a).to(dev)
gp = torch.arange(args.store + 1, device=dev) * args.nodes
parts = ScatterOperator.parts_from_graph_ptr(gp)
t0 = time.perf_counter()
opP = ScatterOperator.from_csr(t(b["rowptr"]), t(b["col"]), t(b["pval"]), b["N"], parts=parts)
opQ = ScatterOperator.from_bsr3(t(b["rowptr"]), t(b["col"]), t(b["blk"]), b["N"], parts=parts)
for f in (opP.csr(), opP.csr_t(), opQ.bsr3(), opQ.bsr3_t()):
    f.sell()
torch.cuda.synchronize()
build_s = time.perf_counter() - t0
xs_all = t(b["pos"]); xv_all = t(b["pos"]).reshape(-1, 1)
g = torch.Generator(device="cpu").manual_seed(0)
def step(ids):
    vP, vQ = opP.select(ids), opQ.select(ids)
    xs = xs_all[vP.row_index()].requires_grad_(True); xv = xv_all[vQ.row_index()].requires_grad_(True)
    Ws = get_Batch_Wjxs(xs, vP, J=args.J); Wv = get_Batch_Wjxs(xv, vQ, J=args.J)
    (Ws.sum() + Wv.sum()).backward()
    return xs.grad, xv.grad
for _ in range(3):
    step(torch.randperm(args.store, generator=g)[:args.batch].to(dev))
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(args.steps):
    step(torch.randperm(args.store, generator=g)[:args.batch].to(dev))
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / args.steps
print(json.dumps({"store_graphs": args.store, "batch_graphs": args.batch, "store_build_s": round(build_s, 3),
                  "ms_per_shuffled_batch": round(dt * 1e3, 3), "graphs_per_s": round(args.batch / dt),
                  "note": "includes drawing the batch (randperm, index tensors, x gather) and the torch .sum() backward"}))
