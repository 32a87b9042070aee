#!/usr/bin/env python3
"""round 3: where the once-per-batch cost of the drop-in goes at cfg2 (ms, synchronised after every stage)."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from esc_gnn_amd import synthetic
b = synthetic.ellipsoid_batch(4096, 500, 10, seed=457893, workers=int(os.environ.get("ESCAT_GEN_WORKERS", min(16, os.cpu_count()))))
import torch
from esc_gnn_amd import ScatterOperator
dev = torch.device("cuda:0")
t = lambda a: torch.from_numpy(a).to(dev)
rowptr, col, pval, blk = t(b["rowptr"]), t(b["col"]), t(b["pval"]), t(b["blk"])
N = b["N"]
def stage(name, fn, res):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize()
    res[name] = round((time.perf_counter() - t0) * 1e3, 3); return r
for rep in range(3):
    res = {}
    p = stage("P from_csr (parts)", lambda: ScatterOperator.from_csr(rowptr, col, pval, N), res)
    stage("P pack (plan + fill)", lambda: p.csr().sell(), res)
    stage("P transpose", lambda: p.csr_t(), res)
    stage("P^T pack (refill)", lambda: p.csr_t().sell(), res)
    q = stage("Q from_bsr3 (parts)", lambda: ScatterOperator.from_bsr3(rowptr, col, blk, N), res)
    f = stage("Q recover factored", lambda: q.factored(), res)
    if f is not None:
        stage("Pt pack", lambda: f.signed.csr().sell(), res)
        stage("Pt transpose", lambda: f.signed.csr_t(), res)
        stage("Pt^T pack", lambda: f.signed.csr_t().sell(), res)
    res["total"] = round(sum(res.values()), 3)
print(json.dumps(res, indent=1))
