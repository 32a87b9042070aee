#!/usr/bin/env python3
"""randomised comparison of the both-tracks launch with the two separate chains (forward and gradients): ragged batches,
random neighbour counts, random custom scale sets, views of stored operators.  Prints one line per failure and a summary."""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=60)
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args()
    import numpy as np
    import torch
    from esc_gnn_amd import synthetic, get_Batch_Wjxs, get_Batch_Wjxs_both, ScatterOperator, BothTracksOperator
    from esc_gnn_amd.process_pyg_data import build_PQ_batch
    dev = torch.device("cuda:0")
    rng = np.random.RandomState(args.seed)
    fails, fused, worst_f, worst_g = 0, 0, 0.0, 0.0
    for case in range(args.cases):
        nmax = int(rng.choice([8, 40, 128, 300, 512]))
        B = int(rng.randint(1, 40))
        k = int(rng.randint(2, 13))
        sizes = [int(rng.randint(max(k + 2, 4), max(nmax, k + 3) + 1)) for _ in range(B)]
        clouds = [synthetic.ellipsoid_clouds(1, n, seed=int(rng.randint(1 << 30)))[0] for n in sizes]
        pos = torch.from_numpy(np.concatenate(clouds)).to(dev)
        gp = torch.from_numpy(np.cumsum([0] + sizes).astype(np.int32)).to(dev)
        b = build_PQ_batch(pos, gp, k, {"kernel": "gaussian", "gaussian_eps": 0.1})
        N = pos.shape[0]
        if rng.rand() < 0.5:
            kw = dict(scales_type="dyadic", J=int(rng.randint(1, 6)), include_lowpass=bool(rng.rand() < 0.7))
        else:
            T = int(rng.randint(1, 20))
            m = int(rng.randint(2, min(T + 1, 8) + 1))
            sc = np.sort(rng.choice(np.arange(0, T + 1), size=min(m, T + 1), replace=False))
            kw = dict(scales_type="custom", diffusion_scales=torch.from_numpy(sc.astype(np.int64)), include_lowpass=bool(rng.rand() < 0.7))
        P, Q = b["P"], b["Q"]
        use_view = rng.rand() < 0.3 and B > 2
        if use_view:
            parts = ScatterOperator.parts_from_graph_ptr(gp)
            P = ScatterOperator.from_csr(b["rowptr"], b["col"], b["pval"], N, parts=parts)
            Q = ScatterOperator.from_bsr3(b["rowptr"], b["col"], b["blk"], N, parts=parts)
            ids = torch.from_numpy(rng.randint(0, B, size=int(rng.randint(1, B + 1))))
            Pv, Qv = P.select(ids), Q.select(ids)
            rows = Pv.row_index()
            n = rows.numel()
            Pu, Qu = Pv, Qv
        else:
            n = N
            Pu, Qu = P, Q
        g = torch.Generator(device=dev).manual_seed(case)
        xs = torch.randn(n, 3, device=dev, generator=g)
        xv = torch.randn(3 * n, 1, device=dev, generator=g)
        try:
            a1, a2 = xs.clone().requires_grad_(True), xv.clone().requires_grad_(True)
            Ws, Wv = get_Batch_Wjxs_both(a1, Pu, a2, Qu, **kw)
            b1, b2 = xs.clone().requires_grad_(True), xv.clone().requires_grad_(True)
            Rs, Rv = get_Batch_Wjxs(b1, Pu, **kw), get_Batch_Wjxs(b2, Qu, **kw)
            Gs, Gv = torch.randn(Ws.shape, device=dev, generator=g), torch.randn(Wv.shape, device=dev, generator=g)
            torch.autograd.backward([Ws, Wv], [Gs, Gv]); torch.autograd.backward([Rs, Rv], [Gs, Gv])
            rel = lambda x, y: float((x - y).abs().max() / y.abs().max().clamp_min(1e-30))
            ef = max(rel(Ws.detach(), Rs.detach()), rel(Wv.detach(), Rv.detach()))
            eg = max(rel(a1.grad, b1.grad), rel(a2.grad, b2.grad))
            worst_f, worst_g = max(worst_f, ef), max(worst_g, eg)
            base = P if not use_view else P
            paired = BothTracksOperator.pair(base, Q) is not None
            fused += int(paired)
            if not (ef <= 3e-6 and eg <= 3e-6) or not all(torch.isfinite(t_).all() for t_ in (Ws, Wv, a1.grad, a2.grad)):
                fails += 1
                print("FAIL", json.dumps({"case": case, "sizes": sizes[:6], "B": B, "k": k, "kw": str(kw), "view": bool(use_view), "ef": ef, "eg": eg}), flush=True)
        except Exception as e:
            fails += 1
            print("EXC", case, B, k, str(kw), bool(use_view), repr(e)[:300], flush=True)
        if case % 10 == 9:
            print(f"... {case + 1} cases, {fails} failures", flush=True)
    print(json.dumps({"cases": args.cases, "failures": fails, "paired": fused, "worst_rel_forward": worst_f, "worst_rel_grad": worst_g}))


if __name__ == "__main__":
    main()
