#!/usr/bin/env python3
"""cProfile of the host side of a small-batch step (32 x 512-node graphs drawn from a stored dataset): where the 0.5-0.6 ms
per step go when the GPU work is ~0.1 ms."""
import cProfile, pstats, os, sys, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from esc_gnn_amd import synthetic
    G, n = 512, 512
    b = synthetic.ellipsoid_batch(G, n, 10, seed=11, workers=16)
    import torch
    from esc_gnn_amd import ScatterOperator, scatter, scatter_both
    dev = torch.device("cuda:0")
    t = lambda a: torch.from_numpy(a).to(dev)
    gp = torch.arange(G + 1, dtype=torch.int32, device=dev) * n
    parts = ScatterOperator.parts_from_graph_ptr(gp)
    P = ScatterOperator.from_csr(t(b["rowptr"]), t(b["col"]), t(b["pval"]), b["N"], parts=parts)
    Q = ScatterOperator.from_bsr3(t(b["rowptr"]), t(b["col"]), t(b["blk"]), b["N"], parts=parts)
    pos = t(b["pos"])
    kw = {"diffusion_kwargs": dict(scales_type="dyadic", J=4)}
    g = torch.Generator().manual_seed(0)
    batches = [torch.randperm(G, generator=g)[:32].to(dev) for _ in range(200)]

    def step(ids, both):
        Pv, Qv = P.select(ids), Q.select(ids)
        x = pos[Pv.row_index()]
        with torch.no_grad():
            if both:
                return scatter_both(x0_scalar=x, P=Pv, x0_vector=x, Q=Qv, kwargs=kw)
            return scatter(track="scalar", x0=x, P_or_Q=Pv, kwargs=kw), scatter(track="vector", x0=x, P_or_Q=Qv, kwargs=kw)
    for both in (False, True):
        for ids in batches[:10]:
            step(ids, both)
        torch.cuda.synchronize()
        pr = cProfile.Profile()
        pr.enable()
        for ids in batches:
            step(ids, both)
        torch.cuda.synchronize()
        pr.disable()
        s = io.StringIO()
        pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28)
        print("=== both" if both else "=== two calls")
        print("\n".join(l for l in s.getvalue().splitlines() if l.strip())[:6000])


if __name__ == "__main__":
    main()
