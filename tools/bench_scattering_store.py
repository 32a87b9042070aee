#!/usr/bin/env python3
"""the reference's own training shape (config/yaml_files/ellipsoids/experiment.yaml: batch_size 32, 512-node graphs, J = 4):
per-step cost of the two _scatter calls on a 32-graph batch drawn from a stored dataset -- computed on views of the stored
operators, and gathered from a ScatteringStore -- plus what the store costs to fill."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--graphs", type=int, default=4096)
    ap.add_argument("--nodes", type=int, default=512)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--gen-workers", type=int, default=16)
    args = ap.parse_args()
    from esc_gnn_amd import synthetic
    b = synthetic.ellipsoid_batch(args.graphs, args.nodes, args.k, seed=11, workers=args.gen_workers)
    import torch
    from esc_gnn_amd import ScatterOperator, ScatteringStore, scatter, scatter_both
    dev = torch.device("cuda:0")
    t = lambda a: torch.from_numpy(a).to(dev)
    gp = torch.arange(args.graphs + 1, dtype=torch.int32, device=dev) * args.nodes
    parts = ScatterOperator.parts_from_graph_ptr(gp)
    ops = {"P": ScatterOperator.from_csr(t(b["rowptr"]), t(b["col"]), t(b["pval"]), b["N"], parts=parts),
           "Q": ScatterOperator.from_bsr3(t(b["rowptr"]), t(b["col"]), t(b["blk"]), b["N"], parts=parts), "graph_ptr": gp}
    pos = t(b["pos"])
    kw = {"diffusion_kwargs": dict(scales_type="dyadic", J=4)}
    res = {"graphs": args.graphs, "nodes": args.nodes, "batch": args.batch}
    for name in ("fill_first_s", "fill_s"):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        sstore = ScatteringStore(ops, pos, pos, kw)
        torch.cuda.synchronize(); res[name] = round(time.perf_counter() - t0, 4)
    res["fill_graphs_per_s"] = round(args.graphs / res["fill_s"])
    res["store_GB"] = round((sstore.Ws.numel() + sstore.Wv.numel()) * 4 / 1e9, 3)
    g = torch.Generator().manual_seed(0)
    batches = [torch.randperm(args.graphs, generator=g)[: args.batch].to(dev) for _ in range(50)]

    def timed(fn):
        for ids in batches[:5]:
            fn(ids)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for ids in batches:
            fn(ids)
        torch.cuda.synchronize()
        return round((time.perf_counter() - t0) / len(batches) * 1e3, 4)

    def two_calls(ids):
        Pv, Qv = ops["P"].select(ids), ops["Q"].select(ids)
        x = pos[sstore.rows_of(ids)]
        with torch.no_grad():
            return scatter(track="scalar", x0=x, P_or_Q=Pv, kwargs=kw), scatter(track="vector", x0=x, P_or_Q=Qv, kwargs=kw)

    def both(ids):
        Pv, Qv = ops["P"].select(ids), ops["Q"].select(ids)
        x = pos[sstore.rows_of(ids)]
        with torch.no_grad():
            return scatter_both(x0_scalar=x, P=Pv, x0_vector=x, Q=Qv, kwargs=kw)
    res["two_scatter_calls_on_views_ms"] = timed(two_calls)
    res["scatter_both_on_views_ms"] = timed(both)
    res["store_batch_ms"] = timed(lambda ids: sstore.batch(ids))
    print(json.dumps(res))


if __name__ == "__main__":
    main()
