#!/usr/bin/env python3
"""experiment: cfg3's multi-launch chains under a TILED row order (8 breadth-first levels x 8 ring neighbours per tile)
instead of the plain breadth-first order -- does a compacter footprint per workgroup trip help the gather of x?"""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nodes", type=int, default=2_000_000)
    ap.add_argument("--k", type=int, default=16)
    ap.add_argument("--J", type=int, default=6)
    ap.add_argument("--A", type=int, default=8, help="levels per tile")
    ap.add_argument("--Bw", type=int, default=8, help="ring neighbours per level and tile")
    args = ap.parse_args()
    from esc_gnn_amd import synthetic
    c = synthetic.single_cloud(args.nodes, args.k, workers=16)
    import torch
    from esc_gnn_amd import get_Batch_Wjxs, ScatterOperator, FactoredVectorOperator, plan_chain, operators
    dev = torch.device("cuda:0")
    t = lambda a: torch.from_numpy(a).to(dev)
    none = (torch.zeros(1, dtype=torch.int32, device=dev), 0, 0)
    n = c["N"]
    plan = plan_chain(J=args.J)
    bfs = operators.locality_order

    def tiled(rowptr, col, n_, max_components=64):
        perm = bfs(rowptr, col, n_, max_components)              # new -> old, breadth-first
        inv = torch.empty_like(perm); inv[perm] = torch.arange(n_, device=perm.device)
        rp = rowptr.long()
        rows = torch.repeat_interleave(torch.arange(n_, device=perm.device), rp[1:] - rp[:-1])
        r2, c2 = inv[rows], inv[col.long()]
        # level of every node in the breadth-first labelling: parent = smallest neighbour label
        cand = torch.where(c2 < r2, c2, torch.full_like(c2, n_))
        parent = torch.full((n_,), n_, dtype=torch.int64, device=perm.device).scatter_reduce(0, r2, cand, reduce="amin", include_self=True)
        root = parent >= n_
        ptr = torch.where(root, torch.arange(n_, device=perm.device), parent)
        depth = (~root).long()
        for _ in range(64):
            nxt = ptr[ptr]
            depth = depth + depth[ptr]
            if bool((nxt == ptr).all()):
                break
            ptr = nxt
        nlev = int(depth.max()) + 1
        start = torch.searchsorted(depth, torch.arange(nlev + 1, device=perm.device))       # depth is sorted in BFS order
        size = (start[1:] - start[:-1])
        lab = torch.arange(n_, device=perm.device)
        rho = lab - start[depth]
        blk = depth // args.A
        ref = size[(blk * args.A).clamp_max(nlev - 1)]                                     # ring size of the block's first level
        tile = (rho.double() / size[depth].double() * (ref.double() / args.Bw)).floor().long()
        key = (blk * (int(ref.max()) // args.Bw + 2) + tile) * n_ + lab
        order = torch.argsort(key)                                                          # BFS labels in tiled order
        return perm[order]

    res = {"nodes": n, "A": args.A, "Bw": args.Bw}
    for name, fn in (("bfs", bfs), ("tiled", tiled)):
        operators.locality_order = fn
        opQ = ScatterOperator.from_bsr3(t(c["rowptr"]), t(c["col"]), t(c["blk"]), n, parts=none)
        opF = FactoredVectorOperator.from_frames(t(c["rowptr"]), t(c["col"]), t(c["pval"]), t(c["frames"]), parts=none)
        xv = t(c["pos"]).reshape(-1, 1).clone()
        os.environ["ESCAT_CHAIN_PATH"] = "multi"
        out = {}
        for nm, op in (("blocks", opQ), ("factored", opF)):
            if nm == "factored":
                os.environ.pop("ESCAT_CHAIN_PATH", None)
            tf = 0.0
            for r in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); W = get_Batch_Wjxs(xv, op, J=args.J); e1.record(); torch.cuda.synchronize()
                if r: tf += e0.elapsed_time(e1) / 2
            out[nm + "_fwd_ms"] = round(tf, 3)
            out[nm + "_checksum"] = float(W.double().abs().sum())
            del W
        res[name] = out
        del opQ, opF
    print(json.dumps(res))


if __name__ == "__main__":
    main()
