#!/usr/bin/env python3
"""BASELINE cfg3: one large point cloud (default 2M nodes, kNN k=16), vector-track Q chain J=6 on one GPU.
The graph does not fit LDS, so this exercises the multi-launch step kernels (operator streamed from HBM each step).
Prints one JSON line: ms per chain, algorithmic GB/s (T(4(N+1)+40nnz)+12N+12NF) and per-step figures."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=2_000_000)
    ap.add_argument("--k", type=int, default=16)
    ap.add_argument("--J", type=int, default=6)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--bwd", action="store_true")
    ap.add_argument("--sort-axis", type=int, default=None)
    ap.add_argument("--check", action="store_true", help="telescoping-sum property check on the result")
    ap.add_argument("--gen-workers", type=int, default=-1, help="processes for the kNN of the cloud (-1: up to 16; 1 under rocprofv3)")
    ap.add_argument("--factored", action="store_true",
                    help="also run the vector operator in factored form (frames + signed scalar operator, 8 B per nonzero)")
    args = ap.parse_args()
    from esc_gnn_amd import synthetic
    t0 = time.time()
    b = synthetic.single_cloud(args.n, args.k, workers=min(16, os.cpu_count()) if args.gen_workers < 0 else args.gen_workers,
                               sort_axis=args.sort_axis)
    gen_s = time.time() - t0
    import torch
    from esc_gnn_amd import get_Batch_Wjxs, ScatterOperator, plan_chain
    dev = torch.device("cuda:0")
    t = lambda a: torch.from_numpy(a).to(dev)
    op = ScatterOperator.from_bsr3(t(b["rowptr"]), t(b["col"]), t(b["blk"]), b["N"], validate=True,
                                   parts=(torch.zeros(1, dtype=torch.int32, device=dev), 0, 0))
    plan = plan_chain(J=args.J)
    xv = t(b["pos"]).reshape(-1, 1).clone().requires_grad_(args.bwd)
    if args.bwd:
        op.bsr3_t()
    N, nnz, T, F = b["N"], b["nnz"], plan.T, plan.F
    g = torch.randn(3 * N, 1, F, device=dev) if args.bwd else None
    tf = tb = 0.0
    for r in range(args.reps + 1):
        e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        e[0].record()
        W = get_Batch_Wjxs(xv, op, J=args.J)
        e[1].record()
        if args.bwd:
            xv.grad = None
            W.backward(g)
        e[2].record()
        torch.cuda.synchronize()
        if r >= 1:
            tf += e[0].elapsed_time(e[1]); tb += e[1].elapsed_time(e[2])
    tf /= args.reps; tb /= args.reps
    alg = T * (4 * (N + 1) + 40 * nnz) + 12 * N + 12 * N * F
    route = "factored form recovered from the blocks (multi-launch k_csr_step)" if op.factored() is not None else "blocks (k_bsr3_step_staged)"
    out = {"workload": f"cfg3: single cloud n={N}, k={args.k}, Q given as BSR-3 blocks, chain J={args.J} (T={T}, F={F})", "route": route,
           "note": "ESCAT_Q_FACTOR_LARGE=0 keeps the blocks; fwd_alg_GBs / frac_of_8TBs use the BLOCK chain's algorithmic bytes",
           "nnz_blocks": nnz,
           "mean_row_blocks": nnz / N, "fwd_ms": tf, "fwd_alg_GBs": alg / (tf * 1e-3) / 1e9, "fwd_ms_per_step": tf / T,
           "frac_of_8TBs": alg / (tf * 1e-3) / 1e9 / 8000.0, "gen_s": gen_s}
    if args.bwd:
        out.update({"bwd_ms": tb, "bwd_alg_GBs": alg / (tb * 1e-3) / 1e9})
    if args.factored:
        from esc_gnn_amd import FactoredVectorOperator
        fop = FactoredVectorOperator.from_frames(t(b["rowptr"]), t(b["col"]), t(b["pval"]), t(b["frames"]),
                                                 parts=(torch.zeros(1, dtype=torch.int32, device=dev), 0, 0))
        if args.bwd:
            fop.signed.csr_t()
        tff = tfb = 0.0
        for r in range(args.reps + 1):
            e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
            e[0].record()
            Wf = get_Batch_Wjxs(xv, fop, J=args.J)
            e[1].record()
            if args.bwd:
                xv.grad = None
                Wf.backward(g)
            e[2].record()
            torch.cuda.synchronize()
            if r >= 1:
                tff += e[0].elapsed_time(e[1]); tfb += e[1].elapsed_time(e[2])
        tff /= args.reps; tfb /= args.reps
        bpn = 8 if fop.signed.csr().signed_pack() is not None else 16
        algf = T * (4 * (N + 1) + bpn * nnz) + 12 * N + 12 * N * F + 2 * 36 * N
        out["factored"] = {"fwd_ms": tff, "fwd_ms_per_step": tff / T, "bytes_per_nonzero": bpn,
                           "fwd_alg_GBs_at_that_many_bytes": algf / (tff * 1e-3) / 1e9,
                           "max_rel_diff_vs_bsr3": float(((Wf.detach() - W.detach()).abs().amax(dim=(0, 1)) / W.detach().abs().amax(dim=(0, 1))).max())}
        if args.bwd:
            out["factored"]["bwd_ms"] = tfb
    if args.check:
        out["telescoping_max_abs_err"] = float((W.detach().sum(-1).reshape(-1) - xv.detach().reshape(-1)).abs().max())
    print(json.dumps(out))

if __name__ == "__main__":
    main()
