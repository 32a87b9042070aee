"""
Frames of ONE LARGE connected cloud recovered from the blocks of its vector operator alone (BASELINE cfg3's shape).

escat_recover_factored (csrc/escat_factor.hip) works per part inside a workgroup's LDS (parts of <= 512 rows).  A cloud of
millions of nodes is one connected component; this is the same algorithm across the whole device, as batched torch
operations in fp64 (ingest-time work, once per operator; no HIP kernel of its own):

  blocks Q_ij = p_ij O_i S_ij O_j^T (process_pyg_data.py:306-313, :492), R_ij = Q_ij / p_ij, wanted: G_i with
  G_i^T R_ij G_j = diag(+-1) on every edge.

  1. tree: the operator was relabelled breadth-first at ingest (operators.locality_order), so the smallest neighbour
     index of a row is its breadth-first parent and the levels are contiguous row ranges (several components: the same
     levels of all of them side by side, as index lists); G_i = R_i,parent G_parent down the
     tree by POINTER DOUBLING (log2(depth) rounds of batched 3x3 products instead of ~600 dependent levels);
  2. gauge: in the tree gauge every edge's holonomy H_e = G_i^T R_ij G_j is V diag(+-1) V^T for one orthonormal V per
     component: V = eigenvectors of a pseudo-randomly weighted sum of the symmetrised H_e, then Newton steps of the joint
     diagonalisation with the signs known (only the cycles that tell two columns apart enter: with local-PCA frames two
     of the three columns flip together on all but a few thousand of 37 M edges, and the plain eigenvectors of the sum
     are then off by 4e-3 rad);
  3. the signs S_ij are now known exactly; the tree walk's drift (one parent per row, ~600 levels of fp32-rounded blocks:
     1e-5) is taken out by a LEVEL WALK -- every row from all its neighbours of the levels before, G_i = polar(sum_j p_ij
     R_ij G_j S_ij) -- and one relaxation sweep over all neighbours: measured 2.6e-7 .. 2.9e-7 x max|Q| at 2 M nodes
     (the tree alone: 1e-5; tree + twelve sweeps without the level walk: 4e-6 and converging by 15 % per sweep).

The caller (ScatterOperator.factored) checks every block in the returned frames and keeps the factored form only if all
are reproduced to its tolerance -- fp16-rounded, 'procrustes' or arbitrary blocks fail that check and stay BSR-3.
"""
import torch


def _inv_t3(X):
    """inverse transpose of a batch of 3x3 matrices (cofactors / determinant)"""
    a, b, c = X[:, 0], X[:, 1], X[:, 2]
    C = torch.stack([torch.linalg.cross(b, c), torch.linalg.cross(c, a), torch.linalg.cross(a, b)], dim=1)
    det = (a * C[:, 0]).sum(dim=1)
    return C / det[:, None, None]


def _polar(X, iters=7):
    """orthogonal polar factor (Newton, X <- (X + X^-T) / 2) of a batch of 3x3 matrices scaled to unit size first"""
    X = X / (X.reshape(-1, 9).norm(dim=1) / 3.0 ** 0.5).clamp_min(1e-300)[:, None, None]
    for _ in range(iters):
        X = 0.5 * (X + _inv_t3(X))
    return X


def workspace_bytes(nnz: int, n: int) -> int:
    """rough peak of the fp64 temporaries below (several (nnz, 3, 3) arrays alive at once)"""
    return 8 * 9 * (6 * nnz + 8 * n)


def recover_frames_large(rowptr: torch.Tensor, col: torch.Tensor, blk: torch.Tensor, newton_steps: int = 3):
    """rowptr / col / blk (nnz, 3, 3): a BSR-3 operator whose rows are in breadth-first order.  Returns frames (n, 3, 3)
    fp32 in that row order, or None when a block is degenerate.  The frames are a CANDIDATE: the caller verifies them
    against every block (a row order that is not breadth-first gives a forest that does not span its components, and
    frames that fail that check)."""
    dev = blk.device
    n = rowptr.numel() - 1
    rp, cl = rowptr.long(), col.long()
    deg = rp[1:] - rp[:-1]
    idx = torch.arange(n, device=dev)
    rows = torch.repeat_interleave(idx, deg)
    B = blk.double()
    p = B[:, 0, :].norm(dim=1)
    if not bool((p > 0).all()):
        return None
    R = B / p[:, None, None]
    # 1. the tree and the walk down it
    cand = torch.where(cl < rows, cl, torch.full_like(cl, n))
    parent = torch.full((n,), n, dtype=torch.int64, device=dev).scatter_reduce(0, rows, cand, reduce="amin", include_self=True)
    is_root = parent >= n
    te = torch.nonzero(cl == parent[rows]).flatten()            # the tree entry of every non-root row
    G = torch.eye(3, dtype=torch.float64, device=dev).repeat(n, 1, 1)
    G[rows[te]] = R[te]
    ptr = torch.where(is_root, idx, parent)
    depth = (~is_root).long()
    for _ in range(64):
        nxt = ptr[ptr]
        G = G @ G[ptr]
        depth = depth + depth[ptr]
        if bool((nxt == ptr).all()):
            break
        ptr = nxt
    root = ptr
    lv_sorted, lv_order = torch.sort(depth, stable=True)        # rows by level (several components: their levels side by side)
    order = torch.argsort(root, stable=True)                    # rows grouped by component
    rid, cnt = torch.unique_consecutive(root[order], return_counts=True)
    comp_of_row = torch.empty(n, dtype=torch.int64, device=dev)
    comp_of_row[order] = torch.repeat_interleave(torch.arange(rid.numel(), device=dev), cnt)

    def comp_sum(X):                                            # (nnz, ...) -> (components, ...), a fixed order of additions
        Xr = torch.segment_reduce(X, "sum", lengths=deg, axis=0)
        return torch.segment_reduce(Xr[order], "sum", lengths=cnt, axis=0)

    # 2. the gauge of every component
    H = G[rows].transpose(1, 2) @ R @ G[cl]
    H = 0.5 * (H + H.transpose(1, 2))
    del R
    w = ((torch.arange(H.shape[0], device=dev) * 2654435761 % 1000003).double() / 1000003.0) + 0.5
    _, V = torch.linalg.eigh(comp_sum(H * w[:, None, None]))
    pairs = ((0, 1), (0, 2), (1, 2))
    for _ in range(newton_steps):
        Ve = V[comp_of_row][rows]
        Hp = Ve.transpose(1, 2) @ H @ Ve
        d = torch.sign(torch.diagonal(Hp, dim1=1, dim2=2))
        num = comp_sum(torch.stack([(d[:, a] - d[:, b]) * Hp[:, a, b] for a, b in pairs], dim=1))
        den = comp_sum(torch.stack([(d[:, a] - d[:, b]) ** 2 for a, b in pairs], dim=1))
        th = torch.where(den > 0, -num / den.clamp_min(1.0), torch.zeros_like(num))
        Th = torch.zeros(rid.numel(), 3, 3, dtype=torch.float64, device=dev)
        Th[:, 0, 1], Th[:, 0, 2], Th[:, 1, 2] = th[:, 0], th[:, 1], th[:, 2]
        V = V @ torch.linalg.matrix_exp(Th - Th.transpose(1, 2))
        del Ve, Hp
        if float(th.abs().max()) < 1e-6:          # (quadratic: what is left after this correction is its square)
            break
    del H
    G = G @ V[comp_of_row]
    # 3. signs, level walk, one sweep
    sg = torch.sign((G[rows] * (B @ G[cl])).sum(dim=1))
    nlev = int(lv_sorted[-1]) + 1
    bounds = torch.searchsorted(lv_sorted, torch.arange(nlev + 1, device=dev)).tolist()
    contiguous = bool((lv_order == idx).all())                  # one component in breadth-first order: levels are row ranges
    ebounds = rp[torch.tensor(bounds, device=dev)].tolist() if contiguous else None
    for l in range(1, nlev):
        a, b_ = bounds[l], bounds[l + 1]
        if contiguous:
            rows_l, cnt_l = slice(a, b_), deg[a:b_]
            ent = slice(ebounds[l], ebounds[l + 1])
        else:
            rows_l = lv_order[a:b_]
            cnt_l = deg[rows_l]
            first = torch.cumsum(cnt_l, 0) - cnt_l
            owner = torch.repeat_interleave(torch.arange(b_ - a, device=dev), cnt_l)
            ent = rp[rows_l][owner] + (torch.arange(owner.numel(), device=dev) - first[owner])
        cj = cl[ent]
        use = (depth[cj] < l).double()
        acc = torch.segment_reduce((B[ent] @ G[cj]) * sg[ent][:, None, :] * use[:, None, None], "sum", lengths=cnt_l, axis=0)
        G[rows_l] = _polar(acc)
    G = _polar(torch.segment_reduce((B @ G[cl]) * sg[:, None, :], "sum", lengths=deg, axis=0))
    if not bool(torch.isfinite(G).all()):
        return None
    return G.float().contiguous()
