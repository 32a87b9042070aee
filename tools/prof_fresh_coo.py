#!/usr/bin/env python3
"""round 3: the fresh-COO function-level drop-in at cfg2, forward only, one policy (ESCAT_INGEST), for rocprofv3 --kernel-trace."""
import os, sys, time, json
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
if os.environ.get("ESCAT_FP16_VALUES"):
    Qv = Qv.half().float(); Pv = Pv.half().float()
xs = t(b["pos"]); xv = t(b["pos"]).reshape(-1, 1)
ts = []
for r in range(6):
    operators._PLAN_CACHE.clear(); operators._PARTS_CACHE.clear()      # as bench.py: the next batch has other graphs
    torch.cuda.synchronize(); t0 = time.perf_counter()
    P = torch.sparse_coo_tensor(Pi, Pv, (N, N), check_invariants=False)
    Q = torch.sparse_coo_tensor(Qi, Qv, (3 * N, 3 * N), check_invariants=False)
    with torch.no_grad():
        torch.cuda.synchronize(); t1 = time.perf_counter()
        Ws = get_Batch_Wjxs(xs, P, scales_type="dyadic", J=4)
        torch.cuda.synchronize(); t2 = time.perf_counter()
        Wv = get_Batch_Wjxs(xv, Q, scales_type="dyadic", J=4)
    torch.cuda.synchronize(); t3 = time.perf_counter()
    ts.append([round((x - y) * 1e3, 3) for x, y in ((t2, t1), (t3, t2), (t3, t0))])
print(json.dumps({"policy": operators.INGEST_POLICY, "ms [P call, Q call, total]": ts[1:]}))
