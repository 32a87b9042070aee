#!/usr/bin/env python3
"""host-side profile (cProfile) of the fresh-COO drop-in loop: where the Python time and the host round trips go"""
import os, sys, time, json, cProfile, pstats, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from esc_gnn_amd import synthetic
b = synthetic.ellipsoid_batch(4096, 500, 10, seed=457893, workers=int(os.environ.get("ESCAT_GEN_WORKERS", min(16, os.cpu_count()))))
import torch
from esc_gnn_amd import get_Batch_Wjxs, operators
dev = torch.device("cuda:0")
t = lambda a: torch.from_numpy(a).to(dev)
N = b["N"]
rp = t(b["rowptr"]).long()
rows = torch.repeat_interleave(torch.arange(N, device=dev), rp[1:] - rp[:-1]); cols = t(b["col"]).long()
Pi = torch.stack([rows, cols]).contiguous()
a = torch.arange(3, device=dev)
Qi = torch.stack([((3 * rows)[:, None, None] + a[None, :, None] + 0 * a[None, None, :]).reshape(-1),
                  ((3 * cols)[:, None, None] + 0 * a[None, :, None] + a[None, None, :]).reshape(-1)]).contiguous()
Pv, Qv = t(b["pval"]), t(b["blk"]).reshape(-1)
xs = t(b["pos"]); xv = t(b["pos"]).reshape(-1, 1)
def step():
    operators._PLAN_CACHE.clear(); operators._PARTS_CACHE.clear()
    P = torch.sparse_coo_tensor(Pi, Pv, (N, N), check_invariants=False)
    Q = torch.sparse_coo_tensor(Qi, Qv, (3 * N, 3 * N), check_invariants=False)
    with torch.no_grad():
        Ws = get_Batch_Wjxs(xs, P, scales_type="dyadic", J=4)
        Wv = get_Batch_Wjxs(xv, Q, scales_type="dyadic", J=4)
    torch.cuda.synchronize()
for _ in range(3): step()
pr = cProfile.Profile()
pr.enable()
for _ in range(5): step()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28)
print(s.getvalue())
