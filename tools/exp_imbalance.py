#!/usr/bin/env python3
"""Experiment: does the spread of row lengths inside a part cost the P chain time?  The same chain on (a) the cfg2
kNN batch and (b) batches of the same shape whose rows all have the same length (random columns inside the graph).
Forward only, J = 0 and J = 4 (slope per step = (t16 - t1) / 15)."""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def regular_batch(B, n, deg, seed=1):
    rng = np.random.RandomState(seed)
    # every row: itself + (deg - 1) distinct random columns of its graph, sorted
    cols = np.empty((B * n, deg), dtype=np.int64)
    for g in range(B):
        r = rng.rand(n, n)
        r[np.arange(n), np.arange(n)] = -1.0          # the diagonal always
        idx = np.argpartition(r, deg, axis=1)[:, :deg]
        cols[g * n:(g + 1) * n] = np.sort(idx, axis=1) + g * n
    rowptr = (np.arange(B * n + 1) * deg).astype(np.int32)
    val = np.full(B * n * deg, 1.0 / deg, dtype=np.float32)
    return rowptr, cols.reshape(-1).astype(np.int32), val


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--graphs", type=int, default=2048)
    ap.add_argument("--nodes", type=int, default=500)
    ap.add_argument("--reps", type=int, default=10)
    args = ap.parse_args()
    from esc_gnn_amd import synthetic
    b = synthetic.ellipsoid_batch(args.graphs, args.nodes, 10, seed=457893, with_q=False, workers=16)
    cases = {"knn": (b["rowptr"], b["col"], b["pval"])}
    for deg in (12, 13, 14, 20):
        cases[f"regular{deg}"] = regular_batch(args.graphs, args.nodes, deg)
    import torch
    from esc_gnn_amd import get_Batch_Wjxs, ScatterOperator, plan_chain
    dev = torch.device("cuda:0")
    t = lambda a: torch.from_numpy(a).to(dev)
    N = args.graphs * args.nodes
    x = torch.randn(N, 3, device=dev)
    res = {}
    for name, (rp, ci, v) in cases.items():
        op = ScatterOperator.from_csr(t(rp), t(ci), t(v), N)
        lens = np.diff(rp)
        out = {"nnz_per_row": float(lens.mean()), "max": int(lens.max())}
        for J in (0, 4):
            for r in range(args.reps + 2):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                if r == 2: tot = 0.0
                e0.record(); W = get_Batch_Wjxs(x, op, J=J); e1.record(); torch.cuda.synchronize()
                if r >= 2: tot += e0.elapsed_time(e1)
            out[f"T{plan_chain(J=J).T}_ms"] = round(tot / args.reps, 4)
        out["us_per_step"] = round((out["T16_ms"] - out["T1_ms"]) / 15 * 1e3, 2)
        out["ns_per_step_per_nnz_row"] = round(out["us_per_step"] * 1e3 / out["nnz_per_row"], 1)
        res[name] = out
        print(name, json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
