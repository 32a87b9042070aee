#!/usr/bin/env python3
"""cfg2 step with both tracks in one launch per direction (escat_chain_both_f32) against the two launches per direction."""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--graphs", type=int, default=4096)
    ap.add_argument("--nodes", type=int, default=500)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--J", type=int, default=4)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--gen-workers", type=int, default=16)
    args = ap.parse_args()
    from esc_gnn_amd import synthetic
    b = synthetic.ellipsoid_batch(args.graphs, args.nodes, args.k, seed=457893, workers=args.gen_workers)
    import torch
    from esc_gnn_amd import ScatterOperator, FactoredVectorOperator, BothTracksOperator, plan_chain
    from esc_gnn_amd.both import _both_launch
    from esc_gnn_amd.pyg_utilities import _chain_launch
    dev = torch.device("cuda:0")
    t = lambda a: torch.from_numpy(a).to(dev)
    N = b["N"]
    P = ScatterOperator.from_csr(t(b["rowptr"]), t(b["col"]), t(b["pval"]), N)
    Qf = FactoredVectorOperator.from_frames(P.csr().rowptr, P.csr().col, P.csr().val, t(b["frames"]),
                                            parts=(P.csr().part_ptr, P.csr().n_parts, P.csr().max_part_rows))
    bop = BothTracksOperator.pair(P, Qf)
    assert bop is not None
    plan = plan_chain(J=args.J)
    xs = torch.randn(N, 3, device=dev); xv = torch.randn(N, 3, device=dev)
    Gs = torch.randn(N, 3, plan.F, device=dev); Gv = torch.randn(N, 3, plan.F, device=dev)

    def timed(fn):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.reps): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / args.reps
    res = {"T": plan.T, "F": plan.F, "dmax": bop.fop.signed.csr().sell().dmax}
    res["both_fwd_ms"] = timed(lambda: _both_launch(bop, plan, xs, xv, False))
    res["both_bwd_ms"] = timed(lambda: _both_launch(bop, plan, Gs, Gv, True))
    res["P_fwd_ms"] = timed(lambda: _chain_launch(P, plan.events, plan.T, plan.F, xs, 3, False))
    res["Q_fwd_ms"] = timed(lambda: _chain_launch(Qf.signed, plan.events, plan.T, plan.F, xv, 3, False, frames=Qf.frames))
    res["P_bwd_ms"] = timed(lambda: _chain_launch(P, plan.events, plan.T, plan.F, Gs, 3, True))
    res["Q_bwd_ms"] = timed(lambda: _chain_launch(Qf.signed, plan.events, plan.T, plan.F, Gv, 3, True, frames=Qf.frames))
    res["step_both_ms"] = res["both_fwd_ms"] + res["both_bwd_ms"]
    res["step_separate_ms"] = res["P_fwd_ms"] + res["Q_fwd_ms"] + res["P_bwd_ms"] + res["Q_bwd_ms"]
    print(json.dumps({k: (round(v, 4) if isinstance(v, float) else v) for k, v in res.items()}))


if __name__ == "__main__":
    main()
