#!/usr/bin/env python3
"""Experiment: does the spread of row lengths inside a part cost the BOTH-TRACKS kernel time?  The fused step on the cfg2
kNN batch and on batches of the same shape whose rows all have the same length (symmetric ring lattices with random node
labels: the pattern must be symmetric for the backward pack).  Forward and backward, J = 4 and J = 0."""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def lattice_batch(B, n, half, seed=1):
    """every node: itself + `half` neighbours on either side of a ring, node labels shuffled per graph -> rows of 2*half+1"""
    rng = np.random.RandomState(seed)
    deg = 2 * half + 1
    cols = np.empty((B * n, deg), dtype=np.int64)
    offs = np.arange(-half, half + 1)
    for g in range(B):
        perm = rng.permutation(n)                      # position on the ring -> node label
        inv = np.empty(n, dtype=np.int64); inv[perm] = np.arange(n)
        ring = perm[(inv[:, None] + offs[None, :]) % n]    # neighbours of node `label` = labels at ring positions +-k
        cols[g * n:(g + 1) * n] = np.sort(ring, axis=1) + g * n
    rowptr = (np.arange(B * n + 1) * deg).astype(np.int32)
    col = cols.reshape(-1).astype(np.int32)
    rows = np.repeat(np.arange(B * n), deg)
    val = np.where(rows == col, 0.5, 0.5 / (deg - 1)).astype(np.float32)
    return rowptr, col, val


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--graphs", type=int, default=4096)
    ap.add_argument("--nodes", type=int, default=500)
    ap.add_argument("--reps", type=int, default=10)
    args = ap.parse_args()
    from esc_gnn_amd import synthetic
    b = synthetic.ellipsoid_batch(args.graphs, args.nodes, 10, seed=457893, with_q=False, workers=16)
    cases = {"knn": (b["rowptr"], b["col"], b["pval"])}
    for half in (6, 7, 10):
        cases[f"lattice{2 * half + 1}"] = lattice_batch(args.graphs, args.nodes, half)
    frames = synthetic.random_frames(args.graphs * args.nodes, 12345)
    import torch
    from esc_gnn_amd import ScatterOperator, FactoredVectorOperator, BothTracksOperator, plan_chain
    from esc_gnn_amd.both import _both_launch
    dev = torch.device("cuda:0")
    t = lambda a: torch.from_numpy(a).to(dev)
    N = args.graphs * args.nodes
    fr = t(frames)
    xs = torch.randn(N, 3, device=dev); xv = torch.randn(N, 3, device=dev)
    for name, (rp, ci, v) in cases.items():
        P = ScatterOperator.from_csr(t(rp), t(ci), t(v), N)
        Qf = FactoredVectorOperator.from_frames(P.csr().rowptr, P.csr().col, P.csr().val, fr,
                                                parts=(P.csr().part_ptr, P.csr().n_parts, P.csr().max_part_rows))
        bop = BothTracksOperator.pair(P, Qf)
        assert bop is not None
        out = {"nnz_per_row": float(np.diff(rp).mean()), "max": int(np.diff(rp).max()), "dmax": bop.fop.signed.csr().sell().dmax}
        for J in (0, 4):
            plan = plan_chain(J=J)
            G = torch.randn(N, 3, plan.F, device=dev)
            for tag, fn in (("fwd", lambda: _both_launch(bop, plan, xs, xv, False)), ("bwd", lambda: _both_launch(bop, plan, G, G, True))):
                for _ in range(3): fn()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(args.reps): fn()
                e1.record(); torch.cuda.synchronize()
                out[f"{tag}_T{plan.T}_ms"] = round(e0.elapsed_time(e1) / args.reps, 4)
        out["fwd_us_per_step"] = round((out["fwd_T16_ms"] - out["fwd_T1_ms"]) / 15 * 1e3, 2)
        print(name, json.dumps(out), flush=True)
        del P, Qf, bop


if __name__ == "__main__":
    main()
