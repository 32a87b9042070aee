import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from esc_gnn_amd import synthetic
b = synthetic.single_cloud(2_000_000, 16, workers=16)
import torch
from esc_gnn_amd import ScatterOperator
dev = torch.device("cuda:0"); t = lambda a: torch.from_numpy(a).to(dev)
rp, col, blk = t(b["rowptr"]), t(b["col"]), t(b["blk"])
for flag in ("0", "auto", "auto"):
    os.environ["ESCAT_REORDER"] = flag
    torch.cuda.synchronize(); t0 = time.perf_counter()
    op = ScatterOperator.from_bsr3(rp, col, blk, b["N"], parts=(torch.zeros(1, dtype=torch.int32, device=dev), 0, 0))
    torch.cuda.synchronize(); print("ESCAT_REORDER=%s: from_bsr3 %.3f s" % (flag, time.perf_counter() - t0))
