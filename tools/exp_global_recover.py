#!/usr/bin/env python3
"""experiment: frames of one large connected cloud recovered from the blocks alone, in torch on the GPU
(tree = min-index neighbour of the breadth-first relabelled operator, pointer doubling, common eigenbasis of the
holonomies, relaxation sweeps), checked like ScatterOperator._factored_from_hint does."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def inv_t3(X):
    """inverse transpose of a batch of 3x3 matrices = cofactor matrix / determinant"""
    import torch
    a, b, c = X[:, 0], X[:, 1], X[:, 2]
    C = torch.stack([torch.linalg.cross(b, c), torch.linalg.cross(c, a), torch.linalg.cross(a, b)], dim=1)
    det = (a * C[:, 0]).sum(dim=1)
    return C / det[:, None, None]


def polar(X, iters=6):
    for _ in range(iters):
        X = 0.5 * (X + inv_t3(X))
    return X


_T0 = [None]


def say(*a):
    import torch
    torch.cuda.synchronize()
    now = time.perf_counter()
    print(f"[{0.0 if _T0[0] is None else now - _T0[0]:8.3f} s]", *a, flush=True)
    _T0[0] = now


def recover(rowptr, col, blk, sweeps, log, newton=3, gs=True):
    import torch
    dev = blk.device
    n = rowptr.numel() - 1
    rp = rowptr.long(); cl = col.long()
    deg = rp[1:] - rp[:-1]
    rows = torch.repeat_interleave(torch.arange(n, device=dev), deg)
    B = blk.double()
    p = B[:, 0, :].norm(dim=1)
    R = B / p[:, None, None]
    # tree: parent = the smallest neighbour index below the row's own
    cand = torch.where(cl < rows, cl, torch.full_like(cl, n))
    parent = torch.full((n,), n, dtype=torch.int64, device=dev).scatter_reduce(0, rows, cand, reduce="amin", include_self=True)
    is_root = parent >= n
    say("parents")
    te = torch.nonzero(cl == parent[rows]).flatten()           # the tree entry of every non-root row
    A = torch.eye(3, dtype=torch.float64, device=dev).repeat(n, 1, 1)
    A[rows[te]] = R[te]                                        # G_i = R_e G_parent
    ptr = torch.where(is_root, torch.arange(n, device=dev), parent)
    depth = (~is_root).long()
    rounds = 0
    while True:
        nxt = ptr[ptr]
        A = A @ A[ptr]
        depth = depth + depth[ptr]
        rounds += 1
        if bool((nxt == ptr).all()):
            break
        ptr = nxt
    log["doubling_rounds"] = rounds; log["roots"] = int(is_root.sum()); log["levels"] = int(depth.max()) + 1
    say("tree walked", log)
    G = A
    root = ptr
    order = torch.argsort(root, stable=True)                   # rows grouped by component
    rid, cnt = torch.unique_consecutive(root[order], return_counts=True)
    comp_of_row = torch.empty(n, dtype=torch.int64, device=dev)
    comp_of_row[order] = torch.repeat_interleave(torch.arange(rid.numel(), device=dev), cnt)

    def comp_sum(X):                                           # (nnz, ...) -> (components, ...), fixed order
        Xr = torch.segment_reduce(X, "sum", lengths=deg, axis=0)
        return torch.segment_reduce(Xr[order], "sum", lengths=cnt, axis=0)

    H = G[rows].transpose(1, 2) @ R @ G[cl]                    # holonomies in the tree gauge
    Hs = 0.5 * (H + H.transpose(1, 2))
    w = ((torch.arange(H.shape[0], device=dev) * 2654435761 % 1000003).double() / 1000003.0) + 0.5
    _, V = torch.linalg.eigh(comp_sum(Hs * w[:, None, None]))
    say("eigh")
    for it in range(newton):                                   # joint diagonalisation with the signs known: V <- V exp(Theta)
        Ve = V[comp_of_row][rows]
        Hp = Ve.transpose(1, 2) @ Hs @ Ve
        d = torch.sign(torch.diagonal(Hp, dim1=1, dim2=2))
        num = torch.stack([(d[:, a] - d[:, b]) * Hp[:, a, b] for a, b in ((0, 1), (0, 2), (1, 2))], dim=1)
        den = torch.stack([(d[:, a] - d[:, b]) ** 2 for a, b in ((0, 1), (0, 2), (1, 2))], dim=1)
        nu, de = comp_sum(num), comp_sum(den)
        th = torch.where(de > 0, -nu / de.clamp_min(1.0), torch.zeros_like(nu))
        Th = torch.zeros(rid.numel(), 3, 3, dtype=torch.float64, device=dev)
        Th[:, 0, 1], Th[:, 0, 2], Th[:, 1, 2] = th[:, 0], th[:, 1], th[:, 2]
        Th = Th - Th.transpose(1, 2)
        V = V @ torch.linalg.matrix_exp(Th)
        log.setdefault("newton_theta", []).append(float(th.abs().max()))
        log.setdefault("distinguishing", []).append([float(x) for x in (de[0] / 4).tolist()])
    say("newton", log.get("newton_theta"))
    G = G @ V[comp_of_row]
    del H, Hs

    def resid(G):
        Mx = G[rows].transpose(1, 2) @ B @ G[cl]
        d = torch.diagonal(Mx, dim1=1, dim2=2)
        v = torch.where(d >= 0, p[:, None], -p[:, None])
        return float((Mx - torch.diag_embed(v)).abs().amax() / B.abs().max())
    log["err_after_tree"] = resid(G)
    say("after tree + eigenbasis", log["err_after_tree"])
    sg = torch.sign((G[rows] * (B @ G[cl])).sum(dim=1))       # the signs S_ij (columns), fixed from here on
    BS = B * 1.0
    if gs:
        # level by level from the roots: every row from ALL its neighbours of the levels before (not its tree parent alone)
        lv_sorted, lv_order = torch.sort(depth, stable=True)
        nlev = int(lv_sorted[-1]) + 1
        bounds = torch.searchsorted(lv_sorted, torch.arange(nlev + 1, device=dev)).tolist()
        contiguous = bool((lv_order == torch.arange(n, device=dev)).all())
        log["levels_contiguous"] = contiguous
        if contiguous:
            rpl = rp.tolist() if n <= 4_000_000 else None
            for l in range(1, nlev):
                a, b_ = bounds[l], bounds[l + 1]
                e0, e1 = rpl[a], rpl[b_]
                cj = cl[e0:e1]
                use = (depth[cj] < l).double()
                RG = (B[e0:e1] @ G[cj]) * sg[e0:e1, None, :] * use[:, None, None]
                acc = torch.segment_reduce(RG, "sum", lengths=deg[a:b_], axis=0)
                G[a:b_] = polar(acc)
            log["err_after_gs"] = resid(G)
            say("level walk", log["err_after_gs"])
    errs = []
    for s in range(sweeps):
        RG = (B @ G[cl]) * sg[:, None, :]
        acc = torch.segment_reduce(RG, "sum", lengths=deg, axis=0)
        G = polar(acc)
        errs.append(resid(G))
        say("sweep", s, errs[-1])
    log["err_after_sweeps"] = errs
    return G


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nodes", type=int, default=200_000)
    ap.add_argument("--k", type=int, default=16)
    ap.add_argument("--sweeps", type=int, default=12)
    ap.add_argument("--builder", action="store_true", help="frames from the product's builder (local PCA) instead of random ones")
    ap.add_argument("--gen-workers", type=int, default=16)
    args = ap.parse_args()
    from esc_gnn_amd import synthetic
    c = synthetic.single_cloud(args.nodes, args.k, workers=args.gen_workers)
    import torch
    from esc_gnn_amd import ScatterOperator
    dev = torch.device("cuda:0")
    t = lambda a: torch.from_numpy(a).to(dev)
    none = (torch.zeros(1, dtype=torch.int32, device=dev), 0, 0)
    os.environ["ESCAT_REORDER"] = "1"
    if args.builder:
        from esc_gnn_amd.process_pyg_data import build_PQ_batch
        bb = build_PQ_batch(t(c["pos"]), torch.tensor([0, c["N"]], dtype=torch.int32), args.k, {'kernel': 'epanechnikov', 'r_cut': 5.0})
        op = ScatterOperator.from_bsr3(bb["rowptr"], bb["col"], bb["blk"], c["N"], parts=none)
    else:
        op = ScatterOperator.from_bsr3(t(c["rowptr"]), t(c["col"]), t(c["blk"]), c["N"], parts=none)
    b = op.bsr3()
    say("operator ready")
    log = {"nodes": c["N"], "nnz": b.nnz, "relabelled": op.row_perm is not None}
    torch.cuda.synchronize(); t0 = time.perf_counter()
    G = recover(b.rowptr, b.col, b.val, args.sweeps, log)
    torch.cuda.synchronize(); log["seconds"] = time.perf_counter() - t0
    print(json.dumps(log))


if __name__ == "__main__":
    main()
