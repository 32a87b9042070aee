"""cost of using an operator ONCE (a loader that hands a fresh torch sparse tensor every step, as the reference's does):
ingest + first chain call, with the SELL pack built at first use vs without a pack (lane-group kernels)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from esc_gnn_amd import synthetic, get_Batch_Wjxs
from esc_gnn_amd.process_pyg_data import build_PQ_batch, coo_parts

dev = torch.device("cuda:0")
B, n = 1024, 500
pos = torch.from_numpy(synthetic.ellipsoid_clouds(B, n, seed=3).reshape(-1, 3)).to(dev)
gp = torch.arange(B + 1, dtype=torch.int32, device=dev) * n
b = build_PQ_batch(pos, gp, 10, {"kernel": "gaussian", "gaussian_eps": 0.05})
pc = b["P"].csr(); rp = pc.rowptr.long()
rows = torch.repeat_interleave(torch.arange(pc.n, device=dev), rp[1:] - rp[:-1]); cols = pc.col.long()
a = torch.arange(3, device=dev)
qi = torch.stack([((3 * rows)[:, None, None] + a[None, :, None] + 0 * a[None, None, :]).reshape(-1),
                  ((3 * cols)[:, None, None] + 0 * a[None, :, None] + a[None, None, :]).reshape(-1)])
qv = b["blk"].reshape(-1)
res = {"graphs": B}
for mode in ("default", "stream"):
    os.environ["ESCAT_CHAIN_PATH"] = "" if mode == "default" else "stream"
    ts = []
    for it in range(4):
        P = torch.sparse_coo_tensor(torch.stack([rows, cols]), pc.val, (pc.n, pc.n))
        Q = torch.sparse_coo_tensor(qi, qv, (3 * pc.n, 3 * pc.n))
        torch.cuda.synchronize(); t0 = time.perf_counter()
        Ws = get_Batch_Wjxs(pos, P, J=4); Wv = get_Batch_Wjxs(pos.reshape(-1, 1), Q, J=4)
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    res[mode + "_first_use_ms"] = round(min(ts[1:]), 2)
print(json.dumps(res))
