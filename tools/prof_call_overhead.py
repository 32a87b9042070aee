import sys, os, cProfile, pstats
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from esc_gnn_amd import synthetic
b = synthetic.ellipsoid_batch(64, 128, 5, seed=1)
import torch
from esc_gnn_amd import get_Batch_Wjxs, ScatterOperator
dev = torch.device("cuda:0"); t = lambda a: torch.from_numpy(a).to(dev)
opP = ScatterOperator.from_csr(t(b["rowptr"]), t(b["col"]), t(b["pval"]), b["N"])
x = t(b["pos"]).clone().requires_grad_(True)
g = None
def run(n):
    global g
    for _ in range(n):
        W = get_Batch_Wjxs(x, opP, J=4)
        if g is None: g = torch.ones_like(W)
        W.backward(g)
run(20); torch.cuda.synchronize()
cProfile.run("run(500); torch.cuda.synchronize()", "/tmp/p.out")
pstats.Stats("/tmp/p.out").sort_stats("tottime").print_stats(18)
