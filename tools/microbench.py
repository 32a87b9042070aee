#!/usr/bin/env python3
"""per-kernel timing sweep: chain time vs T (J = 0..Jmax) for the P and Q chains, fwd and bwd.
time(T) = a + b*T separates the per-graph fixed cost (operator load, slab I/O) from the per-step cost."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--graphs", type=int, default=4096)
    ap.add_argument("--nodes", type=int, default=500)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--Jmax", type=int, default=4)
    ap.add_argument("--Jmin", type=int, default=0)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--half", action="store_true", help="round Q's blocks to fp16 first (half-valued pack, escat_sell_bsr3h.hip)")
    ap.add_argument("--gen-workers", type=int, default=min(16, os.cpu_count()))
    args = ap.parse_args()
    from esc_gnn_amd import synthetic
    b = synthetic.ellipsoid_batch(args.graphs, args.nodes, args.k, seed=457893, workers=args.gen_workers)
    if args.half:
        import numpy as np
        b["blk"] = b["blk"].astype(np.float16).astype(np.float32)
    import torch
    from esc_gnn_amd import get_Batch_Wjxs, ScatterOperator, plan_chain
    dev = torch.device("cuda:0")
    t = lambda a: torch.from_numpy(a).to(dev)
    opP = ScatterOperator.from_csr(t(b["rowptr"]), t(b["col"]), t(b["pval"]), b["N"])
    opQ = ScatterOperator.from_bsr3(t(b["rowptr"]), t(b["col"]), t(b["blk"]), b["N"])
    opP.csr_t(); opQ.bsr3_t()
    xs = t(b["pos"]).clone().requires_grad_(True)
    xv = t(b["pos"]).reshape(-1, 1).clone().requires_grad_(True)
    sq = opQ.bsr3().sell()
    res = {"half_pack": sq.half_val is not None, "dmax": sq.dmax, "max_row_nnz": opQ.bsr3().max_row_nnz, "nnz": b["nnz"], "N": b["N"], "path": os.environ.get("ESCAT_CHAIN_PATH", "auto")}
    for J in range(args.Jmin, args.Jmax + 1):
        plan = plan_chain(J=J)
        for name, x, op in (("P", xs, opP), ("Q", xv, opQ)):
            g = torch.randn(x.shape[0], x.shape[1], plan.F, device=dev)
            tf = tb = 0.0
            for r in range(args.reps + 2):
                x.grad = None
                e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
                e[0].record(); W = get_Batch_Wjxs(x, op, J=J); e[1].record(); W.backward(g); e[2].record()
                torch.cuda.synchronize()
                if r >= 2:
                    tf += e[0].elapsed_time(e[1]); tb += e[1].elapsed_time(e[2])
            res[f"{name}_T{plan.T}"] = {"fwd_ms": round(tf / args.reps, 4), "bwd_ms": round(tb / args.reps, 4)}
    print(json.dumps(res))

if __name__ == "__main__":
    main()
