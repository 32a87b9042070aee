import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from esc_gnn_amd import synthetic, ScatterOperator, operators
B, n, k = [int(a) for a in sys.argv[1:4]]
b = synthetic.ellipsoid_batch(B, n, k, seed=2000 + n)
dev = torch.device("cuda:0")
t = lambda a: torch.from_numpy(a).to(dev)
operators.FACTOR_TOL = 1e9
op = ScatterOperator.from_bsr3(t(b["rowptr"]), t(b["col"]), t(b["blk"]), b["N"])
f = op.factored()
print("factor_err", op.factor_err, "parts", op.bsr3().n_parts, "max rows", op.bsr3().max_part_rows)
bs = op.bsr3()
G = f.frames.double(); v3 = f.signed.csr().val.double()
rp = bs.rowptr.long(); rows = torch.repeat_interleave(torch.arange(bs.n, device=dev), rp[1:] - rp[:-1])
rec = torch.einsum("eac,ec,ebc->eab", G[rows], v3, G[bs.col.long()])
err = (rec - bs.val.double()).abs().amax(dim=(1, 2))
gerr = torch.zeros(B, dtype=torch.float64, device=dev).scatter_reduce(0, rows // n, err, "amax")
bad = torch.nonzero(gerr > 1e-6).flatten().cpu().tolist()
print("graphs with err > 1e-6:", len(bad), bad[:40])
print("errs:", [float(gerr[g]) for g in bad[:40]])
pp = bs.part_ptr.cpu().numpy()
print("part of first bad:", [int(np.searchsorted(pp, g * n, side="right") - 1) for g in bad[:40]], "part_ptr[:6]", pp[:6])
