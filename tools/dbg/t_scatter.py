import sys, os, json, time
sys.path.insert(0, "/root/repo")
from esc_gnn_amd import synthetic
b = synthetic.ellipsoid_batch(4096, 500, 10, seed=457893, workers=8)
import torch
from esc_gnn_amd import scatter, ScatterOperator
dev = torch.device("cuda:0")
t = lambda a: torch.from_numpy(a).to(dev)
opP = ScatterOperator.from_csr(t(b["rowptr"]), t(b["col"]), t(b["pval"]), b["N"])
opQ = ScatterOperator.from_bsr3(t(b["rowptr"]), t(b["col"]), t(b["blk"]), b["N"])
kw = {"diffusion_kwargs": {"scales_type": "dyadic", "J": 4}}
xs = t(b["pos"]).clone(); xv = t(b["pos"]).clone()
def tm(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) / reps, 4)
res = {}
res["scalar_nograd"] = tm(lambda: scatter(track="scalar", x0=xs, P_or_Q=opP, kwargs=kw))
res["vector_nograd"] = tm(lambda: scatter(track="vector", x0=xv, P_or_Q=opQ, kwargs=kw))
xs.requires_grad_(True); xv.requires_grad_(True)
res["scalar_grad_fwd"] = tm(lambda: scatter(track="scalar", x0=xs, P_or_Q=opP, kwargs=kw))
res["vector_grad_fwd"] = tm(lambda: scatter(track="vector", x0=xv, P_or_Q=opQ, kwargs=kw))
print(json.dumps(res))
