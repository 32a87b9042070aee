"""
oracle/builder_oracle.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

CPU restatement of the reference's P / Q operator build
(/root/reference/data_processing/process_pyg_data.py).  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg import it.

Follows, step by step:
  get_P                         process_pyg_data.py:499-600   P = 1/2 (I + D^-1 A)
  edge_index_to_dict            :48-73    neighbours of i = dst of edges with src == i, edge order
  get_C_i_dict                  :330-451  V_i = v_j - v_i (or minus neighbourhood mean), w = kernel(|V_i|), C_i = (w * V_i)^T
  get_local_pca_kernel_weights  :79-138   gaussian / cosine_cutoff / epanechnikov, sqrt
  get_l_singular_vecs           :238-286  U[:, :d] of the full SVD (torch.linalg.svd, the reference's own call)
  calc_O_ij                     :454-493  column-dot sign alignment (or the literal Procrustes formula :315-327)
  process_pyg_data              :773-898  per P-nnz block p_ij * O_ij, indices (d*i+a, d*j+b), dtype cast
and the kNN + symmetrise stand-in used by the goldens (PyG's knn_graph /
to_undirected are not in this image: tie order is "parity unpinned").

Pinned by tests/test_oracle.py against tests/golden/builder_goldens.npz.
"""
import math
import numpy as np
import torch


def knn_sym_edges(pos, k):
    """kNN (k nearest, no self loop) -> both directions -> unique, row-major sorted (int64 (2,E))."""
    from scipy.spatial import cKDTree
    p = np.asarray(pos, dtype=np.float64)
    n = p.shape[0]
    _, nb = cKDTree(p).query(p, k=k + 1)
    src = np.repeat(np.arange(n), k)
    dst = nb[:, 1:].reshape(-1)
    key = np.unique(np.concatenate([src * n + dst, dst * n + src]))
    return np.stack([key // n, key % n]).astype(np.int64)


def kernel_weights(r, kernel, r_cut=None, gaussian_eps=None, take_sqrt=True):
    if kernel == "gaussian":
        w = torch.exp(-r ** 2 / gaussian_eps) if gaussian_eps is not None else torch.exp(-r ** 2)
        if r_cut is not None:
            w = torch.where(r < r_cut, w, torch.zeros_like(w))
    elif kernel == "cosine_cutoff":
        w = 0.5 * (torch.cos(math.pi * r / r_cut) + 1.0)
        w = torch.clamp(torch.where(r < r_cut, w, torch.zeros_like(w)), min=0.0)
    elif kernel == "epanechnikov":
        w = 1.0 - (r / r_cut) ** 2
        w = torch.clamp(torch.where(r < r_cut, w, torch.zeros_like(w)), min=0.0)
    else:
        raise ValueError(f"Unsupported kernel type: {kernel}")
    return torch.sqrt(w) if take_sqrt else w


def get_P_coo(edge_index, n, edge_weight=None, row_normalize=True):
    """returns coalesced (indices int64 (2,nnz), values f32)."""
    ei = torch.as_tensor(edge_index, dtype=torch.int64)
    w = torch.ones(ei.shape[1]) if edge_weight is None else torch.as_tensor(edge_weight, dtype=torch.float32)
    A = torch.sparse_coo_tensor(ei, w, (n, n)).coalesce()
    D = torch.zeros(n).index_add_(0, A.indices()[0], A.values())
    Dinv = torch.where(D > 0, 1.0 / D, torch.zeros_like(D))
    v = Dinv[A.indices()[0]] * A.values()
    idx = A.indices() if row_normalize else A.indices().flip(0)
    eye = torch.arange(n).unsqueeze(0).repeat(2, 1)
    P = torch.sparse_coo_tensor(torch.cat([eye, idx], 1), torch.cat([torch.ones(n), v]), (n, n)).coalesce()
    return P.indices(), 0.5 * P.values()


def local_frames(pos, edge_index, kernel_kwargs, use_mean_recentering=False):
    v = torch.as_tensor(pos, dtype=torch.float32)
    ei = np.asarray(edge_index)
    n, d = v.shape
    nbrs = {}
    for s, t in zip(ei[0].tolist(), ei[1].tolist()):
        nbrs.setdefault(s, []).append(t)
    O = {}
    for i, nb in nbrs.items():
        nb_t = torch.tensor(nb, dtype=torch.long)
        if use_mean_recentering:
            mu = torch.mean(v[torch.cat((torch.tensor([i]), nb_t))], dim=0)
            V = v[nb_t] - mu
        else:
            V = v[nb_t] - v[i]
        r = torch.norm(V, p=2, dim=1)
        w = kernel_weights(r, **kernel_kwargs)
        C_i = (w[:, None] * V).T.contiguous()
        O[i] = torch.linalg.svd(C_i, full_matrices=True)[0][:, :d]
    return O


def calc_O_ij(O_i, O_j, enforce_sign=True, method="column_dot"):
    if enforce_sign:
        if method == "procrustes":
            U, _, Vt = torch.linalg.svd(O_i.T @ O_j, full_matrices=False)
            O_j = O_j @ (U @ Vt)
        else:
            s = torch.where((O_i * O_j).sum(0) >= 0.0, 1.0, -1.0)
            O_j = O_j * s[None, :]
    return O_i @ O_j.T


def build_PQ(pos, edge_index, kernel_kwargs, row_normalize=True, use_mean_recentering=False,
             enforce_sign=True, method="column_dot", edge_weight=None, dtype="float32"):
    """dict {P:{indices,values,size}, Q:{...}} of numpy arrays, like process_pyg_data(return_data_object=False)."""
    pos = np.asarray(pos, dtype=np.float32)
    n, d = pos.shape
    Pi, Pv = get_P_coo(edge_index, n, edge_weight, row_normalize)
    O = local_frames(pos, edge_index, kernel_kwargs, use_mean_recentering)
    nnz = Pi.shape[1]
    Qv = torch.empty(nnz, d, d)
    for e, (i, j) in enumerate(Pi.T.tolist()):
        if i not in O or j not in O:
            raise KeyError(i if i not in O else j)          # isolated node: unsupported, like the reference
        Qv[e] = float(Pv[e]) * calc_O_ij(O[i], O[j], enforce_sign, method)
    a = torch.arange(d)
    ri = (d * Pi[0])[:, None, None] + a[None, :, None] + torch.zeros(1, 1, d, dtype=torch.long)
    ci = (d * Pi[1])[:, None, None] + a[None, None, :] + torch.zeros(1, d, 1, dtype=torch.long)
    Qi = torch.stack([ri.reshape(-1), ci.reshape(-1)])
    tdt = getattr(torch, dtype)
    return {
        "P": {"indices": Pi.numpy(), "values": Pv.to(tdt).numpy(), "size": np.array([n, n], dtype=np.int64)},
        "Q": {"indices": Qi.numpy(), "values": Qv.reshape(-1).to(tdt).numpy(),
              "size": np.array([d * n, d * n], dtype=np.int64)},
        "frames": O,
    }
