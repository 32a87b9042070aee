"""
Seeded synthetic inputs of the reference's shape (host side, numpy): ellipsoid
point clouds (recipe of data_processing/ellipsoid_data_classes.py:82-134),
symmetrised kNN graphs (:209-211), the lazy random-walk operator P = 1/2(I + D^-1 A)
(process_pyg_data.py:575-598) in CSR, and Q blocks p_ij * O_i O_j^T on P's
pattern (process_pyg_data.py:823-857) from given or random orthonormal frames.

Used by bench.py, __graft_entry__.smoke() and the size-scaled GPU tests to make
inputs; it is input generation, not part of the measured path.
"""
import numpy as np


def ellipsoid_clouds(B, n, seed=457892, mu=(3.0, 1.0, 1.0), sd=(0.5, 0.2, 0.2), dtype=np.float32):
    """(B, n, 3); one RandomState stream, graph after graph: the draws AND the arithmetic order of the reference's
    EllipsoidPointCloudGenerator.generate_ellipsoid_points_with_params (ellipsoid_data_classes.py:112-131: semi-axes
    |N(mu, sd)|, phi, theta, unit-sphere point, then the scaling), computed in float64 like the reference and cast
    once -- bit-identical to it (tests/test_host.py, golden generated from the reference class)."""
    rng = np.random.RandomState(seed)
    out = np.empty((B, n, 3), dtype=dtype)
    for g in range(B):
        a = abs(rng.normal(mu[0], sd[0])); b = abs(rng.normal(mu[1], sd[1])); c = abs(rng.normal(mu[2], sd[2]))
        phi = rng.uniform(0, 2 * np.pi, n)
        theta = np.arccos(2 * rng.uniform(0, 1, n) - 1)
        out[g, :, 0] = a * (np.sin(theta) * np.cos(phi))
        out[g, :, 1] = b * (np.sin(theta) * np.sin(phi))
        out[g, :, 2] = c * np.cos(theta)
    return out


def knn_sym_csr(pos, k, workers=1):
    """one graph: symmetrised kNN pattern INCLUDING the diagonal, as CSR (rowptr int64, col int64), cols sorted."""
    from scipy.spatial import cKDTree
    n = pos.shape[0]
    p64 = pos.astype(np.float64)
    _, nb = cKDTree(p64).query(p64, k=k + 1, workers=workers)
    src = np.repeat(np.arange(n, dtype=np.int64), k)
    dst = nb[:, 1:].reshape(-1).astype(np.int64)
    key = np.unique(np.concatenate([src * n + dst, dst * n + src, np.arange(n, dtype=np.int64) * (n + 1)]))
    rows, cols = key // n, key % n
    rowptr = np.zeros(n + 1, dtype=np.int64)
    np.add.at(rowptr, rows + 1, 1)
    return np.cumsum(rowptr), cols


def lazy_walk_values(rowptr, col):
    """P = 1/2 (I + D^-1 A) on a pattern that includes the diagonal; unit edge weights, fp32 like the reference."""
    n = rowptr.shape[0] - 1
    deg = (np.diff(rowptr) - 1).astype(np.float32)            # neighbours, diagonal excluded
    rows = np.repeat(np.arange(n), np.diff(rowptr))
    dinv = np.where(deg > 0, np.float32(1.0) / np.maximum(deg, 1), np.float32(0.0)).astype(np.float32)
    val = (np.float32(0.5) * dinv[rows]).astype(np.float32)
    val[rows == col] = np.float32(0.5)
    return val


def random_frames(N, seed):
    rng = np.random.RandomState(seed)
    q, r = np.linalg.qr(rng.standard_normal((N, 3, 3)))
    q = q * np.sign(np.diagonal(r, axis1=1, axis2=2))[:, None, :]
    return q.astype(np.float32)


def q_blocks_from_frames(rowptr, col, pval, frames):
    """blk[e] = p_ij * O_i @ O_j'^T with the column-dot sign alignment of process_pyg_data.py:290-312."""
    n = rowptr.shape[0] - 1
    rows = np.repeat(np.arange(n), np.diff(rowptr))
    Oi, Oj = frames[rows], frames[col]
    s = np.where(np.einsum("eac,eac->ec", Oi, Oj) >= 0, 1.0, -1.0).astype(np.float32)
    return (pval[:, None, None] * np.einsum("eac,ec,ebc->eab", Oi, s, Oj)).astype(np.float32)


def _knn_chunk(args):
    clouds, k = args
    return [knn_sym_csr(c, k) for c in clouds]


def ellipsoid_batch(B, n, k, seed=457892, with_q=True, frames_seed=12345, workers=1):
    """
    Block-diagonal batch of B ellipsoid graphs with n nodes each.
    Returns dict: pos (N,3) f32, rowptr (N+1) int32, col int32, pval f32, blk (nnz,3,3) f32 or None,
    graph_ptr (B+1) int32, nnz.  workers > 1 spreads the per-graph kNN over processes
    (call it before the process touches the GPU: it forks).
    """
    clouds = ellipsoid_clouds(B, n, seed)
    if workers > 1 and B >= 4 * workers:
        import multiprocessing as mp
        chunks = np.array_split(np.arange(B), workers * 4)
        with mp.get_context("fork").Pool(workers) as pool:
            parts = pool.map(_knn_chunk, [(clouds[c], k) for c in chunks if len(c)])
        graphs = [g for part in parts for g in part]
    else:
        graphs = _knn_chunk((clouds, k))
    rps, cols, off = [np.zeros(1, dtype=np.int64)], [], 0
    for g, (rp, c) in enumerate(graphs):
        rps.append(rp[1:] + off)
        cols.append(c + g * n)
        off += c.shape[0]
    rowptr = np.concatenate(rps)
    col = np.concatenate(cols)
    pval = lazy_walk_values(rowptr, col)
    pos = clouds.reshape(B * n, 3)
    frames = random_frames(B * n, frames_seed) if with_q else None
    blk = q_blocks_from_frames(rowptr, col, pval, frames) if with_q else None
    return dict(pos=pos, rowptr=rowptr.astype(np.int32), col=col.astype(np.int32), pval=pval, blk=blk, frames=frames,
                graph_ptr=(np.arange(B + 1) * n).astype(np.int32), nnz=int(col.shape[0]), N=B * n)


def single_cloud(n, k, seed=457892, workers=-1, frames_seed=12345, sort_axis=None):
    """cfg3: one ellipsoid cloud (a,b,c)=(3,1,1) with n points, kNN k; Q in BSR-3.
    sort_axis: None keeps the sampler's random node order (the reference's); 0/1/2 sorts the nodes along that
    coordinate first (a locality experiment: neighbours then have nearby indices)."""
    rng = np.random.RandomState(seed)
    phi = rng.uniform(0, 2 * np.pi, n)
    theta = np.arccos(2 * rng.uniform(0, 1, n) - 1)
    pos = np.column_stack([3.0 * np.sin(theta) * np.cos(phi), np.sin(theta) * np.sin(phi), np.cos(theta)]).astype(np.float32)
    if sort_axis is not None:
        pos = pos[np.argsort(pos[:, sort_axis], kind="stable")]
    rowptr, col = knn_sym_csr(pos, k, workers=workers)
    pval = lazy_walk_values(rowptr, col)
    frames = random_frames(n, frames_seed)
    blk = q_blocks_from_frames(rowptr, col, pval, frames)
    return dict(pos=pos, rowptr=rowptr.astype(np.int32), col=col.astype(np.int32), pval=pval, blk=blk, frames=frames,
                graph_ptr=np.array([0, n], dtype=np.int32), nnz=int(col.shape[0]), N=n)
