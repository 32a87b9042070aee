#!/usr/bin/env python3
"""_get_vector_invariants at cfg2 shape (W' = 22 wavelets, d = 3): GPU drop-in vs the CPU restatement of the reference
(Python loop over nodes for the quantiles) on a bounded sample."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("--graphs", type=int, default=4096); ap.add_argument("--nodes", type=int, default=500); ap.add_argument("--k", type=int, default=10)
ap.add_argument("--gen-workers", type=int, default=8, help="use 1 under rocprofv3 (no fork after GPU init)")
ap.add_argument("--W", type=int, default=22); ap.add_argument("--cpu-graphs", type=int, default=4); ap.add_argument("--reps", type=int, default=5)
args = ap.parse_args()
import numpy as np
from esc_gnn_amd import synthetic
b = synthetic.ellipsoid_batch(args.graphs, args.nodes, args.k, seed=457893, workers=args.gen_workers, with_q=False)
import torch
from esc_gnn_amd.invariants import get_vector_invariants
from oracle.invariants_oracle import ref_vector_invariants
N = b["N"]
rows = np.repeat(np.arange(N), np.diff(b["rowptr"])); cols = b["col"].astype(np.int64)
off = rows != cols                                   # edge_index has no self loops
ei = torch.from_numpy(np.stack([cols[off], rows[off]]))          # (src, dst)
g = torch.Generator().manual_seed(0)
Wv = torch.randn(N, args.W, 1, 3, generator=g)
dev = torch.device("cuda:0")
x = Wv.to(dev).requires_grad_(True); eid = ei.to(dev)
res = {"N": N, "E": int(ei.shape[1]), "W": args.W}
out = get_vector_invariants(x, eid)
G = torch.randn_like(out)
for name, bwd in (("fwd", False), ("fwd+bwd", True)):
    for _ in range(2):
        o = get_vector_invariants(x, eid)
        if bwd: x.grad = None; o.backward(G)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(args.reps):
        o = get_vector_invariants(x, eid)
        if bwd: x.grad = None; o.backward(G)
    torch.cuda.synchronize(); res[name + "_ms"] = round((time.perf_counter() - t0) / args.reps * 1e3, 2)
res["gpu_graphs_per_s_fwd_bwd"] = round(args.graphs / (res["fwd+bwd_ms"] * 1e-3))
n = args.cpu_graphs * args.nodes
keep = (ei[0] < n) & (ei[1] < n)
torch.set_num_threads(1)
xc = Wv[:n].clone().requires_grad_(True)
t0 = time.perf_counter()
r = ref_vector_invariants(xc, ei[:, keep]); r.backward(torch.ones_like(r))
dt = time.perf_counter() - t0
res["cpu_port_graphs_per_s_fwd_bwd"] = round(args.cpu_graphs / dt, 2); res["cpu_sample"] = f"{args.cpu_graphs} graphs, {dt:.1f} s, 1 thread"
assert (out[:n].detach().cpu() - r.detach()).abs().max() < 1e-5
print(json.dumps(res))
