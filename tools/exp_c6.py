#!/usr/bin/env python3
"""Experiment: the P chain on 3 and on 6 channels (the same kernel family; C = 6 gathers two 16-byte pieces per slot and
holds a six-channel slab: one workgroup per CU).  Slope per step tells what doubling the work per wave and step costs."""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--graphs", type=int, default=4096)
    ap.add_argument("--nodes", type=int, default=500)
    ap.add_argument("--reps", type=int, default=10)
    args = ap.parse_args()
    from esc_gnn_amd import synthetic
    b = synthetic.ellipsoid_batch(args.graphs, args.nodes, 10, seed=457893, with_q=False, workers=16)
    import torch
    from esc_gnn_amd import get_Batch_Wjxs, ScatterOperator, plan_chain
    dev = torch.device("cuda:0")
    t = lambda a: torch.from_numpy(a).to(dev)
    N = b["N"]
    op = ScatterOperator.from_csr(t(b["rowptr"]), t(b["col"]), t(b["pval"]), N)
    for C in (3, 4, 6, 8):
        x = torch.randn(N, C, device=dev)
        out = {}
        for J in (0, 2, 4):
            for r in range(args.reps + 2):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                if r == 2: tot = 0.0
                e0.record(); W = get_Batch_Wjxs(x, op, J=J); e1.record(); torch.cuda.synchronize()
                if r >= 2: tot += e0.elapsed_time(e1)
            out[f"T{plan_chain(J=J).T}_ms"] = round(tot / args.reps, 4)
        out["us_per_step"] = round((out["T16_ms"] - out["T1_ms"]) / 15 * 1e3, 2)
        print("C =", C, json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
