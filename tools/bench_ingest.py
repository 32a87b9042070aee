#!/usr/bin/env python3
"""once-per-batch costs around the chain: COO ingest (coalesce -> CSR / BSR-3), transposes, SELL pack + slot plan."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("--graphs", type=int, default=1024); ap.add_argument("--nodes", type=int, default=500); ap.add_argument("--k", type=int, default=10)
args = ap.parse_args()
import numpy as np
from esc_gnn_amd import synthetic
b = synthetic.ellipsoid_batch(args.graphs, args.nodes, args.k, seed=457893, workers=8)
import torch
from esc_gnn_amd import ScatterOperator
dev = torch.device("cuda:0")
t = lambda a: torch.from_numpy(a).to(dev)
N, nnz = b["N"], b["nnz"]
rows = torch.repeat_interleave(torch.arange(N, device=dev), t(np.diff(b["rowptr"]).astype(np.int64)))
cols = t(b["col"]).long()
P = torch.sparse_coo_tensor(torch.stack([rows, cols]), t(b["pval"]), (N, N))
ii = (3 * rows[:, None, None] + torch.arange(3, device=dev)[None, :, None]).expand(-1, 3, 3).reshape(-1)
jj = (3 * cols[:, None, None] + torch.arange(3, device=dev)[None, None, :]).expand(-1, 3, 3).reshape(-1)
Q = torch.sparse_coo_tensor(torch.stack([ii, jj]), t(b["blk"]).reshape(-1), (3 * N, 3 * N))
def timed(f):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = f(); torch.cuda.synchronize(); return r, (time.perf_counter() - t0) * 1e3
res = {"graphs": args.graphs, "N": N, "nnz": nnz}
for rep in range(2):
    opP, res["P from_torch+csr ms"] = timed(lambda: (lambda o: (o.csr(), o)[1])(ScatterOperator.from_torch(P)))
    opQ, res["Q from_torch+bsr3 ms"] = timed(lambda: (lambda o: (o.bsr3(), o)[1])(ScatterOperator.from_torch(Q)))
    _, res["P transpose ms"] = timed(lambda: opP.csr_t())
    _, res["Q transpose ms"] = timed(lambda: opQ.bsr3_t())
    _, res["P sell (fwd+bwd) ms"] = timed(lambda: (opP.csr().sell(), opP.csr_t().sell()))
    _, res["Q sell (fwd+bwd) ms"] = timed(lambda: (opQ.bsr3().sell(), opQ.bsr3_t().sell()))
print(json.dumps(res, indent=1))
