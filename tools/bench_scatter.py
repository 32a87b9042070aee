#!/usr/bin/env python3
"""_scatter (orders 0/1/2) at cfg2: scalar + vector track, forward and forward+backward."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("--graphs", type=int, default=4096); ap.add_argument("--nodes", type=int, default=500)
ap.add_argument("--k", type=int, default=10); ap.add_argument("--J", type=int, default=4); ap.add_argument("--reps", type=int, default=10)
args = ap.parse_args()
from esc_gnn_amd import synthetic
b = synthetic.ellipsoid_batch(args.graphs, args.nodes, args.k, seed=457893, workers=8)
import torch
from esc_gnn_amd import scatter, scatter_both, ScatterOperator
dev = torch.device("cuda:0")
t = lambda a: torch.from_numpy(a).to(dev)
opP = ScatterOperator.from_csr(t(b["rowptr"]), t(b["col"]), t(b["pval"]), b["N"])
opQ = ScatterOperator.from_bsr3(t(b["rowptr"]), t(b["col"]), t(b["blk"]), b["N"])
kw = {"diffusion_kwargs": {"scales_type": "dyadic", "J": args.J}}
xs = t(b["pos"]).clone().requires_grad_(True); xv = t(b["pos"]).clone().requires_grad_(True)
def run(bwd):
    Ws = scatter(track="scalar", x0=xs, P_or_Q=opP, kwargs=kw)
    Wv = scatter(track="vector", x0=xv, P_or_Q=opQ, kwargs=kw)
    if bwd:
        xs.grad = None; xv.grad = None
        torch.autograd.backward([Ws, Wv], [gs, gv])
    return Ws, Wv
Ws, Wv = run(False)
gs, gv = torch.randn_like(Ws), torch.randn_like(Wv)
res = {}
for name, bwd in (("fwd", False), ("fwd+bwd", True)):
    for _ in range(3): run(bwd)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(args.reps): run(bwd)
    torch.cuda.synchronize(); res[name + "_ms"] = round((time.perf_counter() - t0) / args.reps * 1e3, 3)
res["graphs_per_s_fwd_bwd"] = round(args.graphs / (res["fwd+bwd_ms"] * 1e-3))
# the same through the both-tracks call (first-order banks from one launch per direction, orders per track)
def run2(bwd, grad):
    a, v = (xs, xv) if grad else (xs.detach(), xv.detach())
    Ws, Wv = scatter_both(x0_scalar=a, P=opP, x0_vector=v, Q=opQ, kwargs=kw)
    if bwd:
        xs.grad = None; xv.grad = None
        torch.autograd.backward([Ws, Wv], [gs, gv])
for name, bwd, grad in (("both_fwd_nograd", False, False), ("both_fwd+bwd", True, True)):
    for _ in range(3): run2(bwd, grad)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(args.reps): run2(bwd, grad)
    torch.cuda.synchronize(); res[name + "_ms"] = round((time.perf_counter() - t0) / args.reps * 1e3, 3)
def run3():
    scatter(track="scalar", x0=xs.detach(), P_or_Q=opP, kwargs=kw); scatter(track="vector", x0=xv.detach(), P_or_Q=opQ, kwargs=kw)
for _ in range(3): run3()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(args.reps): run3()
torch.cuda.synchronize(); res["fwd_nograd_ms"] = round((time.perf_counter() - t0) / args.reps * 1e3, 3)
print(json.dumps(res))
