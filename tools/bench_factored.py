"""cfg2 with the vector operator in factored form (FactoredVectorOperator) next to the BSR-3 chain: fwd and fwd+bwd of
the Q track, and the whole P+Q step.   python tools/bench_factored.py [--graphs 4096 --nodes 500 --k 10]"""
import argparse
import json
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--graphs", type=int, default=4096)
    ap.add_argument("--nodes", type=int, default=500)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--J", type=int, default=4)
    ap.add_argument("--iters", type=int, default=20)
    a = ap.parse_args()
    from esc_gnn_amd import synthetic, get_Batch_Wjxs
    from esc_gnn_amd.process_pyg_data import build_PQ_batch
    dev = torch.device("cuda:0")
    B, n = a.graphs, a.nodes
    pos = torch.from_numpy(synthetic.ellipsoid_clouds(B, n, seed=457893).reshape(-1, 3)).to(dev)
    gp = torch.arange(B + 1, dtype=torch.int32, device=dev) * n
    b = build_PQ_batch(pos, gp, a.k, {"kernel": "gaussian", "gaussian_eps": 0.05}, factored=True)
    xs = pos.clone().requires_grad_(True)
    xv = pos.reshape(-1, 1).clone().requires_grad_(True)
    F = a.J + 2
    gs = torch.ones(B * n, 3, F, device=dev)
    gv = torch.ones(3 * B * n, 1, F, device=dev)

    def step(opP, opQ, bwd=True, p=True, q=True):
        if p:
            W = get_Batch_Wjxs(xs, opP, J=a.J)
            if bwd:
                W.backward(gs)
        if q:
            W = get_Batch_Wjxs(xv, opQ, J=a.J)
            if bwd:
                W.backward(gv)

    def timed(fn):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / a.iters

    res = {"graphs": B, "nodes": n, "k": a.k, "J": a.J, "nnz": b["nnz"]}
    for name, Q in (("bsr3", b["Q"]), ("factored", b["Q_factored"])):
        res[name] = {"Q_fwd_ms": round(timed(lambda: step(b["P"], Q, False, False, True)), 4),
                     "Q_fwd_bwd_ms": round(timed(lambda: step(b["P"], Q, True, False, True)), 4),
                     "PQ_fwd_bwd_ms": round(timed(lambda: step(b["P"], Q)), 4)}
        res[name]["graphs_per_s"] = round(B / res[name]["PQ_fwd_bwd_ms"] * 1e3, 1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
