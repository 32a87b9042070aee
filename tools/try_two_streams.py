#!/usr/bin/env python3
"""experiment: the two tracks' chains on two HIP streams (tails and barriers of one fill with the other) vs back to back"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from esc_gnn_amd import synthetic
b = synthetic.ellipsoid_batch(4096, 500, 10, seed=457893, workers=16)
import torch
from esc_gnn_amd import get_Batch_Wjxs, ScatterOperator
dev = torch.device("cuda:0")
t = lambda a: torch.from_numpy(a).to(dev)
opP = ScatterOperator.from_csr(t(b["rowptr"]), t(b["col"]), t(b["pval"]), b["N"])
opQ = ScatterOperator.from_bsr3(t(b["rowptr"]), t(b["col"]), t(b["blk"]), b["N"])
xs = t(b["pos"]).clone(); xv = t(b["pos"]).reshape(-1, 1).clone()
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def seq():
    with torch.no_grad():
        get_Batch_Wjxs(xs, opP, J=4); get_Batch_Wjxs(xv, opQ, J=4)
def par():
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur); s2.wait_stream(cur)
    with torch.no_grad():
        with torch.cuda.stream(s1): get_Batch_Wjxs(xs, opP, J=4)
        with torch.cuda.stream(s2): get_Batch_Wjxs(xv, opQ, J=4)
    cur.wait_stream(s1); cur.wait_stream(s2)
res = {}
for name, f in (("sequential", seq), ("two_streams", par), ("sequential2", seq), ("two_streams2", par)):
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    res[name] = round(e0.elapsed_time(e1) / 20, 4)
print(json.dumps(res))
