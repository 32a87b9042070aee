import os, sys, time, json
sys.path.insert(0, '/root/repo' if os.path.isdir('/root/repo/esc_gnn_amd') else os.getcwd())
from esc_gnn_amd import synthetic
b = synthetic.ellipsoid_batch(4096, 500, 10, seed=457893, workers=16)
import torch
from esc_gnn_amd import ScatterOperator
from esc_gnn_amd import operators
dev = torch.device("cuda:0")
t = lambda a: torch.from_numpy(a).to(dev)
rowptr, col, pval, blk = t(b["rowptr"]), t(b["col"]), t(b["pval"]), t(b["blk"])
N = b["N"]
def timed(f):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = f(); torch.cuda.synchronize(); return r, (time.perf_counter() - t0) * 1e3
res = {}
for rep in range(3):
    p_, res["from_csr"] = timed(lambda: ScatterOperator.from_csr(rowptr, col, pval, N))
    q_, res["from_bsr3"] = timed(lambda: ScatterOperator.from_bsr3(rowptr, col, blk, N))
    _, res["validate"] = timed(lambda: operators._validate(rowptr, col, N))
    _, res["closed_ranges"] = timed(lambda: operators._closed_row_ranges(rowptr, col, N, 256))
    _, res["csr_t"] = timed(lambda: p_.csr_t())
    _, res["bsr3_t"] = timed(lambda: q_.bsr3_t())
    _, res["sell_P"] = timed(lambda: p_.csr().sell())
    _, res["sell_Pt"] = timed(lambda: p_.csr_t().sell())
    _, res["sell_Q"] = timed(lambda: q_.bsr3().sell())
    _, res["sell_Qt"] = timed(lambda: q_.bsr3_t().sell())
print(json.dumps(res, indent=1))
