#!/usr/bin/env python3
"""cfg3 (one large cloud, multi-launch path): the vector chain on the BSR-3 operator against the same chain on the factored
operator Q = D Pt D^T (two rotation launches around the 3-channel scalar chain on Pt).  Times and agreement."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nodes", type=int, default=2_000_000)
    ap.add_argument("--k", type=int, default=16)
    ap.add_argument("--J", type=int, default=6)
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--gen-workers", type=int, default=16)
    args = ap.parse_args()
    from esc_gnn_amd import synthetic
    c = synthetic.single_cloud(args.nodes, args.k, workers=args.gen_workers)
    import torch
    from esc_gnn_amd import get_Batch_Wjxs, ScatterOperator, FactoredVectorOperator, plan_chain
    dev = torch.device("cuda:0")
    t = lambda a: torch.from_numpy(a).to(dev)
    none = (torch.zeros(1, dtype=torch.int32, device=dev), 0, 0)
    n = c["N"]
    opQ = ScatterOperator.from_bsr3(t(c["rowptr"]), t(c["col"]), t(c["blk"]), n, parts=none)
    opF = FactoredVectorOperator.from_frames(t(c["rowptr"]), t(c["col"]), t(c["pval"]), t(c["frames"]), parts=none)
    plan = plan_chain(J=args.J)
    res = {"nodes": n, "nnz": c["nnz"], "T": plan.T, "F": plan.F, "fits_register_path": opF.fits_register_path(plan.F),
           "signed_relabelled": opF.signed.row_perm is not None}
    g = torch.randn(3 * n, 1, plan.F, device=dev)
    outs = {}
    for name, op in (("bsr3", opQ), ("factored", opF)):
        xv = t(c["pos"]).reshape(-1, 1).clone().requires_grad_(True)
        tf = tb = 0.0
        for r in range(args.reps + 1):
            e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
            xv.grad = None
            e[0].record(); W = get_Batch_Wjxs(xv, op, J=args.J); e[1].record(); W.backward(g); e[2].record()
            torch.cuda.synchronize()
            if r >= 1:
                tf += e[0].elapsed_time(e[1]) / args.reps; tb += e[1].elapsed_time(e[2]) / args.reps
        res[name] = {"fwd_ms": round(tf, 3), "bwd_ms": round(tb, 3), "fwd_ms_per_step": round(tf / plan.T, 4)}
        outs[name] = (W.detach(), xv.grad.clone())
        del W
    res["max_rel_diff_W"] = float((outs["bsr3"][0] - outs["factored"][0]).abs().max() / outs["bsr3"][0].abs().max())
    res["max_rel_diff_grad"] = float((outs["bsr3"][1] - outs["factored"][1]).abs().max() / outs["bsr3"][1].abs().max())
    print(json.dumps(res))


if __name__ == "__main__":
    main()
