"""
GPU build of the diffusion operators P and Q -- drop-in for
data_processing/process_pyg_data.py::process_pyg_data (:640-919) plus a batched
sibling for whole datasets (the reference loops over graphs in Python,
scripts/python/generate_ellipsoid_dataset.py:1165-1183).

  build_PQ_batch(pos, graph_ptr, k, ...)   many graphs per call: kNN -> symmetrise -> P (CSR) -> local-PCA
                                           frames -> Q (BSR-3), all on the GPU (esc_gnn_amd/csrc/escat_build.hip);
                                           the result feeds get_Batch_Wjxs directly (no ingest) and splits into the
                                           reference's per-graph COO dicts / HDF5 groups.
  process_pyg_data(data, ...)              the reference's signature for ONE graph; returns the same nested dict
                                           (return_data_object=False) or an object carrying sparse P and Q.

Kernel steps and their reference lines are listed in include/escat.h.  Graph construction: 'k-nn' (the default, with the
Gaussian RBF x cosine-cutoff edge weights of build_weighted_radius_adjacency, :173-232), 'radius' (all neighbours inside
r_cutoff among the max_num_neighbors <= 32 nearest -- torch-cluster's choice of WHICH neighbours survive its cap is
unpinned; identical whenever the cap is not reached), 'reweight_existing_edges' (with the reference's pairing of
coalesced weights and un-coalesced edge list, see process_pyg_data below), or None (a given edge_index).
rank_deficiency_strategy='Tikhonov' is supported (:262-279).  Not reproduced: Bessel edge features and Dirac selection
(QM9 path, out of scope) and custom kernel callables: these raise NotImplementedError.
"""
import ctypes
import math
from typing import Any, Dict, Optional
import torch

from . import _lib
from .operators import ScatterOperator, FactoredVectorOperator
from .pyg_utilities import _ptr, _stream

_KERNELS = {"gaussian": 0, "cosine_cutoff": 1, "epanechnikov": 2}


def _kernel_args(kw: Dict[str, Any]):
    kw = dict(kw or {})
    name = kw.get("kernel", "gaussian")
    if name not in _KERNELS:
        raise ValueError(f"Unsupported kernel type: {name}")
    eps, rc = kw.get("gaussian_eps"), kw.get("r_cut")
    if name != "gaussian" and rc is None:
        raise ValueError(f"r_cut must be provided for {name} kernel.")
    return (_KERNELS[name], float(eps) if eps is not None else 0.0, int(eps is not None),
            float(rc) if rc is not None else 0.0, int(rc is not None), int(bool(kw.get("take_sqrt", True))))


def frames_and_Q(rowptr, col, pval, pos, kernel_kwargs, use_mean_recentering=False, enforce_sign=True,
                 sing_vect_align_method="column_dot", tikhonov_eps: Optional[float] = None):
    """B2 + B3 on a given CSR pattern (diagonal included) and its P values; returns (frames (N,3,3), blk (nnz,3,3))."""
    lib = _lib.load()
    dev = pos.device
    n = pos.shape[0]
    pos = pos.to(torch.float32).contiguous()
    frames = torch.empty(n, 3, 3, dtype=torch.float32, device=dev)
    blk = torch.empty(col.numel(), 3, 3, dtype=torch.float32, device=dev)
    iso = torch.zeros(1, dtype=torch.int32, device=dev)
    kid, eps, has_eps, rc, has_rc, sq = _kernel_args(kernel_kwargs)
    with torch.cuda.device(dev):
        st = _stream(dev)
        _lib.check(lib.escat_build_frames_ex(_ptr(rowptr), _ptr(col), _ptr(pos), n, kid, eps, has_eps, rc, has_rc, sq,
                                             int(bool(use_mean_recentering)), int(tikhonov_eps is not None),
                                             float(tikhonov_eps or 0.0), _ptr(frames), _ptr(iso), st), "escat_build_frames_ex")
        _lib.check(lib.escat_build_Q_bsr3(_ptr(rowptr), _ptr(col), _ptr(pval), _ptr(frames), n, int(bool(enforce_sign)),
                                          int(sing_vect_align_method == "procrustes"), _ptr(blk), st), "escat_build_Q_bsr3")
    if int(iso.item()):
        raise KeyError(f"{int(iso.item())} node(s) have no neighbours: the reference cannot build a frame for them "
                       "(process_pyg_data.py:62-73, :823-826)")
    return frames, blk


def _kernel_weights_torch(r: torch.Tensor, kw: Dict[str, Any]) -> torch.Tensor:
    """get_local_pca_kernel_weights (process_pyg_data.py:79-138) as torch ops, for the generic-dimension path below"""
    kw = dict(kw or {})
    name, rc, eps = kw.get("kernel", "gaussian"), kw.get("r_cut"), kw.get("gaussian_eps")
    _kernel_args(kw)                                  # the same argument errors as the d = 3 path
    if name == "gaussian":
        w = torch.exp(-r ** 2 / eps) if eps is not None else torch.exp(-r ** 2)
        if rc is not None:
            w = torch.where(r < rc, w, torch.zeros_like(w))
    elif name == "cosine_cutoff":
        w = torch.clamp(torch.where(r < rc, 0.5 * (torch.cos(math.pi * r / rc) + 1.0), torch.zeros_like(r)), min=0.0)
    else:
        w = torch.clamp(torch.where(r < rc, 1.0 - (r / rc) ** 2, torch.zeros_like(r)), min=0.0)
    return torch.sqrt(w) if kw.get("take_sqrt", True) else w


def frames_and_Q_any_dim(rowptr, col, pval, v, kernel_kwargs, use_mean_recentering=False, enforce_sign=True,
                         sing_vect_align_method="column_dot", tikhonov_eps: Optional[float] = None):
    """B2 + B3 for vector features of any dimension d (process_pyg_data.py:330-451 is dimension-generic; the theory
    scripts use d = 2, theory_verification/vector_diffusion_equivariance_viz.py:322-395).  Not the hot path -- d = 3 has
    its own kernels (frames_and_Q) -- so this is batched torch on the GPU: the left singular vectors of every node's
    weighted neighbour matrix C_i (d x deg) as the eigenvectors of C_i C_i^T (fp64 eigh, descending; with the
    reference's 'Tikhonov' branch, :262-279, the regularised covariance has the same eigenvectors), then
    Q_ij = p_ij O_i S_ij O_j^T.  Column signs of a frame differ from LAPACK's SVD, which enforce_sign=True cancels
    (both alignments are invariant to them); with enforce_sign=False the blocks depend on LAPACK's sign choice: unpinned.
    Returns (frames (n, d, d), blk (nnz, d, d))."""
    dev = v.device
    n, d = v.shape
    rp = rowptr.long()
    rows = torch.repeat_interleave(torch.arange(n, device=dev), rp[1:] - rp[:-1])
    cols = col.long()
    off = rows != cols
    er, ec = rows[off], cols[off]
    deg = torch.bincount(er, minlength=n)
    if bool((deg == 0).any()):
        raise KeyError(f"{int((deg == 0).sum())} node(s) have no neighbours: the reference cannot build a frame for them "
                       "(process_pyg_data.py:62-73, :823-826)")
    if use_mean_recentering:                          # mean over the node and its neighbours (:64-67)
        mu = (torch.zeros(n, d, device=dev).index_add_(0, er, v[ec]) + v) / (deg + 1).to(v.dtype)[:, None]
        V = v[ec] - mu[er]
    else:
        V = v[ec] - v[er]
    w = _kernel_weights_torch(V.norm(dim=1), kernel_kwargs)
    WV = (w[:, None] * V).double()
    cov = torch.zeros(n, d, d, dtype=torch.float64, device=dev).index_add_(0, er, WV[:, :, None] * WV[:, None, :])
    if tikhonov_eps is not None:
        cov = cov + float(tikhonov_eps) * torch.eye(d, dtype=torch.float64, device=dev)
    O = torch.linalg.eigh(cov)[1].flip(-1).to(torch.float32).contiguous()          # columns by descending singular value
    Oi, Oj = O[rows], O[cols]
    if enforce_sign:
        if sing_vect_align_method == "procrustes":                                # :296-304
            U, _, Vt = torch.linalg.svd(Oi.transpose(1, 2) @ Oj, full_matrices=False)
            Oj = Oj @ (U @ Vt)
        else:                                                                      # column_dot (:306-313)
            sgn = torch.where((Oi * Oj).sum(1) >= 0.0, 1.0, -1.0)
            Oj = Oj * sgn[:, None, :]
    blk = pval[:, None, None] * (Oi @ Oj.transpose(1, 2))
    return O, blk.contiguous()


def _knn_P_grid_batch(pos, gp, k, lib, dev):
    """escat_build_knn_P_grid per cloud (any size), results joined with global column ids."""
    rps, cols, vals, base = [], [], [], 0
    for g in range(len(gp) - 1):
        n = gp[g + 1] - gp[g]
        p = pos[gp[g]:gp[g + 1]].contiguous()
        cap = n * (2 * k + 1)
        rp = torch.empty(n + 1, dtype=torch.int32, device=dev)
        co = torch.empty(cap, dtype=torch.int32, device=dev)
        va = torch.empty(cap, dtype=torch.float32, device=dev)
        nnz_d = torch.zeros(1, dtype=torch.int32, device=dev)
        ws = torch.empty(lib.escat_build_grid_workspace_bytes(n, k) // 4, dtype=torch.int32, device=dev)
        with torch.cuda.device(dev):
            _lib.check(lib.escat_build_knn_P_grid(_ptr(p), n, k, _ptr(rp), _ptr(co), _ptr(va), cap, _ptr(nnz_d), _ptr(ws),
                                                  ws.numel() * 4, _stream(dev)), "escat_build_knn_P_grid")
        nnz = int(nnz_d.item())
        if nnz < 0 or base + nnz >= 2 ** 31:
            raise OverflowError("escat: more than 2^31 nonzeros")
        rps.append(rp[:-1] + base if g + 1 < len(gp) - 1 else rp + base)
        cols.append(co[:nnz] + gp[g]); vals.append(va[:nnz])
        base += nnz
        del ws
    return torch.cat(rps).contiguous(), torch.cat(cols).contiguous(), torch.cat(vals).contiguous()


def _knn_pattern(pos, graph_ptr, k):
    """symmetrised kNN pattern (diagonal included) of a batch as CSR: the first third of build_PQ_batch."""
    lib = _lib.load()
    dev = pos.device
    N, B = pos.shape[0], graph_ptr.numel() - 1
    m = int((graph_ptr[1:] - graph_ptr[:-1]).max())
    if m * ((m + 31) // 32) * 4 > 150 * 1024:
        rowptr, col, _ = _knn_P_grid_batch(pos, graph_ptr.tolist(), k, lib, dev)
        return rowptr, col
    cap = N * (2 * k + 1)
    rowptr = torch.empty(N + 1, dtype=torch.int32, device=dev)
    col = torch.empty(cap, dtype=torch.int32, device=dev)
    pval = torch.empty(cap, dtype=torch.float32, device=dev)
    nnz_d = torch.zeros(1, dtype=torch.int32, device=dev)
    ws = torch.empty(lib.escat_build_workspace_bytes(N, B, k) // 4, dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        _lib.check(lib.escat_build_knn_P(_ptr(pos), _ptr(graph_ptr), B, N, m, k, _ptr(rowptr), _ptr(col), _ptr(pval), cap,
                                         _ptr(nnz_d), _ptr(ws), ws.numel() * 4, _stream(dev)), "escat_build_knn_P")
    return rowptr, col[:int(nnz_d.item())]


def build_PQ_batch(pos: torch.Tensor, graph_ptr: torch.Tensor, k: int, local_pca_kernel_fn_kwargs: Dict[str, Any],
                   use_mean_recentering: bool = False, enforce_sign: bool = True,
                   sing_vect_align_method: str = "column_dot", max_graph_nodes: Optional[int] = None,
                   factored: bool = False):
    """
    pos (N,3) fp32 on the GPU, graph_ptr int32 (B+1,) node ranges.  Returns a dict with the canonical operators:
    rowptr (N+1), col (nnz), pval (nnz), blk (nnz,3,3), frames (N,3,3), nnz, and ScatterOperators 'P', 'Q'
    (+ 'Q_factored', the same vector operator as frames + signed scalar operator, with factored=True).
    """
    if not pos.is_cuda:
        raise RuntimeError("escat: build_PQ_batch needs GPU tensors (no CPU fallback)")
    lib = _lib.load()
    dev = pos.device
    pos = pos.to(torch.float32).contiguous()
    graph_ptr = graph_ptr.to(device=dev, dtype=torch.int32).contiguous()
    N, B = pos.shape[0], graph_ptr.numel() - 1
    true_max = int((graph_ptr[1:] - graph_ptr[:-1]).max()) if B > 0 else 0
    if max_graph_nodes is None:
        max_graph_nodes = true_max
    elif max_graph_nodes < true_max:                   # it sizes the kernels' LDS bitmap: too small would overflow LDS
        raise ValueError(f"escat: max_graph_nodes={max_graph_nodes} but the largest graph has {true_max} nodes")
    if max_graph_nodes * ((max_graph_nodes + 31) // 32) * 4 > 150 * 1024:
        # beyond the LDS adjacency bitmap of escat_build_knn_P (~1090 nodes): the grid builder, one cloud at a time
        rowptr, col, pval = _knn_P_grid_batch(pos, graph_ptr.tolist(), k, lib, dev)
        nnz = col.numel()
    else:
        cap = N * (2 * k + 1)
        rowptr = torch.empty(N + 1, dtype=torch.int32, device=dev)
        col = torch.empty(cap, dtype=torch.int32, device=dev)
        pval = torch.empty(cap, dtype=torch.float32, device=dev)
        nnz_d = torch.zeros(1, dtype=torch.int32, device=dev)
        ws = torch.empty(lib.escat_build_workspace_bytes(N, B, k) // 4, dtype=torch.int32, device=dev)
        with torch.cuda.device(dev):
            _lib.check(lib.escat_build_knn_P(_ptr(pos), _ptr(graph_ptr), B, N, max_graph_nodes, k, _ptr(rowptr), _ptr(col),
                                             _ptr(pval), cap, _ptr(nnz_d), _ptr(ws), ws.numel() * 4, _stream(dev)),
                       "escat_build_knn_P")
        nnz = int(nnz_d.item())
        col, pval = col[:nnz], pval[:nnz]
    frames, blk = frames_and_Q(rowptr, col, pval, pos, local_pca_kernel_fn_kwargs, use_mean_recentering, enforce_sign,
                               sing_vect_align_method)
    out = {"rowptr": rowptr, "col": col, "pval": pval, "blk": blk, "frames": frames, "nnz": nnz, "graph_ptr": graph_ptr,
           "P": ScatterOperator.from_csr(rowptr, col, pval, N, validate=False),
           "Q": ScatterOperator.from_bsr3(rowptr, col, blk, N, validate=False)}
    if sing_vect_align_method == "column_dot" or not enforce_sign:
        out["Q"].hint_frames(frames, pval)      # Q.factored() then needs no recovery (and serves clouds of any size)
    if factored:
        if sing_vect_align_method != "column_dot" and enforce_sign:
            raise ValueError("escat: only the column_dot alignment (or no sign enforcement) has the factored form")
        # the same operator in factored form (frames + signed scalar operator), see FactoredVectorOperator
        out["Q_factored"] = FactoredVectorOperator.from_frames(rowptr, col, pval, frames, enforce_sign)
    return out


def coo_parts(batch: Dict[str, Any], g: int, hdf5_tensor_dtype: str = "float16"):
    """graph g of a build_PQ_batch result as the reference's nested dict (process_pyg_data.py:884-898):
    P coalesced row-major, Q in block-emission order (one 3x3 block per P-nonzero, row-major inside)."""
    gp = batch["graph_ptr"]
    r0, r1 = int(gp[g]), int(gp[g + 1])
    rp = batch["rowptr"][r0:r1 + 1].long()
    e0, e1 = int(rp[0]), int(rp[-1])
    n = r1 - r0
    rows = torch.repeat_interleave(torch.arange(n, device=rp.device), rp[1:] - rp[:-1])
    cols = batch["col"][e0:e1].long() - r0
    tdt = getattr(torch, hdf5_tensor_dtype)
    d = int(batch["blk"].shape[1])                      # 3 on the hot path; any d from frames_and_Q_any_dim
    a = torch.arange(d, device=rp.device)
    qi = torch.stack([((d * rows)[:, None, None] + a[None, :, None] + 0 * a[None, None, :]).reshape(-1),
                      ((d * cols)[:, None, None] + 0 * a[None, :, None] + a[None, None, :]).reshape(-1)])
    return {"original_idx": g,
            "P": {"indices": torch.stack([rows, cols]), "values": batch["pval"][e0:e1].to(tdt), "size": torch.Size([n, n])},
            "Q": {"indices": qi, "values": batch["blk"][e0:e1].reshape(-1).to(tdt), "size": torch.Size([d * n, d * n])}}


def _get_P_csr(edge_index, n, edge_weight, row_normalize, dev):
    """get_P (process_pyg_data.py:499-600) for an arbitrary edge list: A = coalesce(edges), P = 1/2 (I + D^-1 A)."""
    ei = edge_index.to(dev).long()
    w = torch.ones(ei.shape[1], device=dev) if edge_weight is None else edge_weight.to(dev, torch.float32)
    A = torch.sparse_coo_tensor(ei, w, (n, n)).coalesce()
    D = torch.zeros(n, device=dev).index_add_(0, A.indices()[0], A.values())
    Dinv = torch.where(D > 0, 1.0 / D, torch.zeros_like(D))
    v = Dinv[A.indices()[0]] * A.values()
    idx = A.indices() if row_normalize else A.indices().flip(0)
    eye = torch.arange(n, device=dev).unsqueeze(0).repeat(2, 1)
    P = torch.sparse_coo_tensor(torch.cat([eye, idx], 1), torch.cat([torch.ones(n, device=dev), v]), (n, n)).coalesce()
    rows, cols, vals = P.indices()[0], P.indices()[1], 0.5 * P.values()
    cnt = torch.bincount(rows, minlength=n)
    rowptr = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    torch.cumsum(cnt, 0, out=rowptr[1:])
    return rowptr.to(torch.int32), cols.to(torch.int32).contiguous(), vals.contiguous(), A


def process_pyg_data(
    data,
    data_i: Optional[int] = None,
    row_normalize: bool = True,
    use_mean_recentering: bool = False,
    vector_feat_key: str = 'pos',
    local_pca_kernel_fn=None,
    local_pca_kernel_fn_kwargs: Dict[str, Any] = {'kernel': 'epanechnikov', 'r_cut': 5.0},
    enforce_sign: bool = True,
    device: Optional[str] = None,
    *,
    graph_construction: Optional[str] = 'k-nn',
    graph_construction_kwargs: Dict[str, Any] = {'r_cutoff': 5.0, 'gaussian_eps': 4.0, 'max_num_neighbors': 4},
    return_data_object: bool = True,
    rank_deficiency_strategy: Optional[str] = None,
    tikhonov_eps: float = 1e-3,
    adjacency_rbf_eps: Optional[float] = None,
    num_edge_features: int = 0,
    include_diracs: bool = False,
    dirac_fn=None,
    dirac_kwargs: Dict[str, Any] = {},
    hdf5_tensor_dtype: str = 'float16',
    edge_feature_type: str = 'bessel',
    sing_vect_align_method: str = 'column_dot',
):
    """Same arguments and return structure as the reference's process_pyg_data (see module docstring for the
    options that raise).  `data` is any object with item/attribute access to `pos` (N,3) and `edge_index` (2,E)."""
    if local_pca_kernel_fn is not None:
        raise NotImplementedError("escat: custom local_pca_kernel_fn callables cannot run inside the kernel; "
                                  "use local_pca_kernel_fn_kwargs with kernel = gaussian | cosine_cutoff | epanechnikov")
    if rank_deficiency_strategy is not None and str(rank_deficiency_strategy).lower() != 'tikhonov':
        rank_deficiency_strategy = None                 # the reference only acts on 'Tikhonov' (:262-263) and ignores the rest
    if num_edge_features or include_diracs:
        raise NotImplementedError("escat: Bessel edge features / Dirac selection are outside the scattering hot path")
    if not torch.cuda.is_available():
        raise RuntimeError("escat: process_pyg_data needs a GPU (no CPU fallback); use the reference on CPU")
    dev = torch.device("cuda", torch.cuda.current_device())
    get = (lambda k: data[k]) if hasattr(data, "__getitem__") else (lambda k: getattr(data, k))
    v = get(vector_feat_key).to(dev, torch.float32).contiguous()
    n, d = v.shape
    if d < 1:
        raise ValueError("escat: vector features need at least one component")
    edge_weight = getattr(data, "edge_weight", None)
    A_vals = None
    if graph_construction is not None:
        gck = graph_construction_kwargs
        if graph_construction in ('k-nn', 'radius'):
            k = int(gck.get('max_num_neighbors', 32))
            if k > 32:
                raise NotImplementedError("escat: max_num_neighbors > 32 is beyond the kNN kernel")
            gp = torch.tensor([0, n], dtype=torch.int32, device=dev)
            rp32, col32 = _knn_pattern(v, gp, min(k, max(n - 1, 1)))         # pattern only; values come below, with weights
            rp = rp32.long()
            rows = torch.repeat_interleave(torch.arange(n, device=dev), rp[1:] - rp[:-1])
            cols = col32.long()
            off = rows != cols
            edge_index = torch.stack([rows[off], cols[off]])        # symmetrised, row-major sorted: to_undirected's order
            if graph_construction == 'radius':
                # radius_graph(pos, r_cutoff, max_num_neighbors) + to_undirected (:207-214): of the (at most)
                # max_num_neighbors nearest, those inside the cutoff; an edge survives if either end keeps it
                dd = (v[edge_index[0]] - v[edge_index[1]]).norm(dim=1)
                edge_index = edge_index[:, dd <= float(gck['r_cutoff'])]
        elif graph_construction == 'reweight_existing_edges':
            edge_index = get('edge_index').to(dev)
        else:
            raise NotImplementedError(f"escat: graph_construction={graph_construction!r}")
        # Gaussian RBF x cosine cutoff weights (build_weighted_radius_adjacency, :217-223)
        dists = (v[edge_index[0]] - v[edge_index[1]]).norm(dim=1)
        gauss = torch.exp(-((dists / gck['gaussian_eps']) ** 2))
        cosine_cut = 0.5 * (torch.cos(math.pi * dists / gck['r_cutoff']) + 1.0)
        edge_weight = gauss * cosine_cut
        if graph_construction == 'reweight_existing_edges':
            # The reference keeps the weights of the COALESCED adjacency (A_sparse._values(), :758: row-major order,
            # duplicates summed) and get_P pairs them with the edge list in its ORIGINAL order (:563-579): on an edge
            # list that is not row-major sorted the weights land on other edges.  Reproduced as it is (golden
            # tests/golden/round3b_goldens.npz); duplicate edges make the two lengths differ, where the reference fails.
            av = torch.sparse_coo_tensor(edge_index, edge_weight, (n, n)).coalesce().values()
            if av.numel() != edge_weight.numel():
                raise ValueError("escat: 'reweight_existing_edges' with duplicate edges: the coalesced weights "
                                 f"({av.numel()}) no longer match the edge list ({edge_weight.numel()}); the reference "
                                 "fails here too (to_torch_coo_tensor, process_pyg_data.py:575)")
            edge_weight = av
        try:
            data.edge_index, data.edge_weight = edge_index, edge_weight
        except Exception:
            try:
                data['edge_index'], data['edge_weight'] = edge_index, edge_weight
            except Exception:
                pass
    else:
        edge_index = get('edge_index')
    rowptr, col, pval, A = _get_P_csr(edge_index, n, edge_weight, row_normalize, dev)
    if graph_construction is not None:
        A_vals = edge_weight                # what the reference leaves in data.edge_weight (:758): the coalesced weights
    build = frames_and_Q if d == 3 else frames_and_Q_any_dim       # d = 3: the HIP kernels; any other d: batched torch on the GPU
    frames, blk = build(rowptr, col, pval, v, local_pca_kernel_fn_kwargs, use_mean_recentering, enforce_sign,
                        sing_vect_align_method, tikhonov_eps if rank_deficiency_strategy is not None else None)
    batch = {"rowptr": rowptr, "col": col, "pval": pval, "blk": blk,
             "graph_ptr": torch.tensor([0, n], dtype=torch.int32)}
    parts = coo_parts(batch, 0, hdf5_tensor_dtype if not return_data_object else "float32")
    if return_data_object:
        P = torch.sparse_coo_tensor(parts["P"]["indices"], parts["P"]["values"], (n, n)).coalesce()
        Q = torch.sparse_coo_tensor(parts["Q"]["indices"], parts["Q"]["values"], (d * n, d * n))
        # an ESCGNNData (:870-882) carrying every attribute of `data` plus the operators; P and Q are the reference's
        # torch sparse tensors, with the canonical device form already attached (get_Batch_Wjxs skips the ingest)
        from .data_classes import ESCGNNData
        keys = list(data.keys()) if hasattr(data, "keys") else [k_ for k_ in vars(data) if not k_.startswith("_")]
        out = ESCGNNData(**{k_: get(k_) for k_ in keys if k_ not in ("P", "Q")})
        if graph_construction is not None:
            out.edge_index, out.edge_weight = edge_index, A_vals
        P._escat_operator = ScatterOperator.from_csr(rowptr, col, pval, n, validate=False)
        if d == 3:                                      # (any other d: get_Batch_Wjxs ingests the COO as a scalar operator)
            Q._escat_operator = ScatterOperator.from_bsr3(rowptr, col, blk, n, validate=False)
            if sing_vect_align_method == "column_dot" or not enforce_sign:
                Q._escat_operator.hint_frames(frames, pval)     # the frames the blocks were made from: no recovery needed
        out.P, out.Q = P, Q
        return out
    parts["original_idx"] = data_i
    if graph_construction is not None:
        parts["edge_index"] = edge_index.cpu().to(torch.int64)
        parts["edge_weight"] = A_vals.to(getattr(torch, hdf5_tensor_dtype)).cpu()
    return parts


def write_dataset_hdf5(path: str, pos: torch.Tensor, graph_ptr: torch.Tensor, k: int,
                       local_pca_kernel_fn_kwargs: Dict[str, Any], hdf5_tensor_dtype: str = "float16",
                       chunk_graphs: int = 16384, store_frames: bool = False, index_offset: int = 0,
                       compression: Optional[str] = "gzip", **build_kwargs) -> Dict[str, Any]:
    """
    compression: 'gzip' (default) writes indices / values as the reference does (create_dataset(..., compression='gzip'),
    generate_ellipsoid_dataset.py:1192-1200; chunks deflated by a thread pool), None writes them contiguous (2-3x the
    bytes, several times the write rate: for scratch files that only esc_gnn_amd.store reads back).
    The P/Q -> HDF5 loop of scripts/python/generate_ellipsoid_dataset.py:1165-1213 for a whole dataset: graphs are
    built on the GPU in chunks (build_PQ_batch), the per-graph COO arrays of the reference layout are formed on the
    GPU for the whole chunk, copied to the host once and written by esc_gnn_amd.hdf5_min (atomic rename at the end).
    index_offset: sample index of the first graph (a shard of a larger dataset: write_dataset_hdf5_sharded).
    store_frames=True adds /frames/{idx}/values, the (n, 3, 3) fp32 local frames -- an extension of the layout (the
    reference's reader never looks at it) that lets esc_gnn_amd.store load the vector operator in factored form.
    """
    import numpy as np
    from .hdf5_min import H5Writer
    dev = pos.device
    gp_all = graph_ptr.to(torch.int64).cpu()
    B = gp_all.numel() - 1
    tdt = getattr(torch, hdf5_tensor_dtype)
    total_nnz = 0
    with H5Writer(path) as f:
        for g0 in range(0, B, chunk_graphs):
            g1 = min(B, g0 + chunk_graphs)
            n0, n1 = int(gp_all[g0]), int(gp_all[g1])
            gp = (gp_all[g0:g1 + 1] - n0).to(torch.int32)
            b = build_PQ_batch(pos[n0:n1], gp.to(dev), k, local_pca_kernel_fn_kwargs, **build_kwargs)
            total_nnz += b["nnz"]
            rp = b["rowptr"].long()
            n = n1 - n0
            rows = torch.repeat_interleave(torch.arange(n, device=dev), rp[1:] - rp[:-1])
            gid = torch.repeat_interleave(torch.arange(g1 - g0, device=dev), (gp[1:] - gp[:-1]).to(dev).long())
            base = gp.to(dev).long()[gid]                                   # first node of each row's graph
            lrow, lcol = rows - base[rows], b["col"].long() - base[rows]
            eoff = rp[gp.to(dev).long()].cpu().numpy()                      # nnz offset of each graph
            P_idx = torch.stack([lrow, lcol]).cpu().numpy()
            P_val = b["pval"].to(tdt).cpu().numpy()
            Q_val = b["blk"].reshape(-1).to(tdt).cpu().numpy()
            a = torch.arange(3, device=dev)
            Q_r = ((3 * lrow)[:, None, None] + a[None, :, None] + 0 * a[None, None, :]).reshape(-1).cpu().numpy()
            Q_c = ((3 * lcol)[:, None, None] + 0 * a[None, :, None] + a[None, None, :]).reshape(-1).cpu().numpy()
            gpn = gp.numpy()
            fr = b["frames"].cpu().numpy() if store_frames else None
            items = []
            for g in range(g1 - g0):
                e0, e1 = int(eoff[g]), int(eoff[g + 1])
                ng = int(gpn[g + 1] - gpn[g])
                idx = index_offset + g0 + g
                items.append((f"P/{idx}/indices", P_idx[:, e0:e1]))
                items.append((f"P/{idx}/values", P_val[e0:e1]))
                items.append((f"P/{idx}/size", np.array([ng, ng], dtype=np.int64), False))       # plain, as the reference's
                items.append((f"Q/{idx}/indices", np.stack([Q_r[9 * e0:9 * e1], Q_c[9 * e0:9 * e1]])))
                items.append((f"Q/{idx}/values", Q_val[9 * e0:9 * e1]))
                items.append((f"Q/{idx}/size", np.array([3 * ng, 3 * ng], dtype=np.int64), False))
                if store_frames:
                    items.append((f"frames/{idx}/values", fr[int(gpn[g]):int(gpn[g + 1])]))
                if len(items) >= 2048 or g == g1 - g0 - 1:
                    f.write_many(items, compression=compression)
                    items = []
    return {"graphs": B, "nnz": total_nnz, "path": path}


def write_dataset_hdf5_sharded(path: str, pos: torch.Tensor, graph_ptr: torch.Tensor, k: int,
                               local_pca_kernel_fn_kwargs: Dict[str, Any], rank: Optional[int] = None,
                               world: Optional[int] = None, merge: bool = True, **kwargs) -> Dict[str, Any]:
    """
    The dataset build over several GPUs (scripts/python/parallel_process_dataset.py:239-330 does it with a
    process pool and temp files): graphs are independent, so rank r builds the contiguous range dist_utils.graph_shard
    gives it and writes `path + f'.shard{r}'` with GLOBAL sample indices -- no collective on the data path; after a
    barrier rank 0 merges the shards into `path` (hdf5_min.merge_h5_files) and removes them.  rank / world default to
    the torch.distributed environment; every rank passes the same pos / graph_ptr (or at least its own range of pos
    rows at their global positions).
    """
    from .dist_utils import env_rank_world, graph_shard
    from .hdf5_min import merge_h5_files
    if rank is None or world is None:
        rank, _, world = env_rank_world()
    gp_all = graph_ptr.to(torch.int64).cpu()
    B = gp_all.numel() - 1
    g0, g1 = graph_shard(B, rank, world)
    shard = f"{path}.shard{rank}"
    n0, n1 = int(gp_all[g0]), int(gp_all[g1])
    info = {"graphs": 0, "nnz": 0, "path": shard}
    if g1 > g0:
        info = write_dataset_hdf5(shard, pos[n0:n1], gp_all[g0:g1 + 1] - n0, k, local_pca_kernel_fn_kwargs,
                                  index_offset=g0, **kwargs)
    else:
        from .hdf5_min import H5Writer
        with H5Writer(shard):
            pass
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()
    info.update(rank=rank, world=world, shard_range=(g0, g1))
    if merge and rank == 0:
        info["merged_datasets"] = merge_h5_files(path, [f"{path}.shard{r}" for r in range(world)])
        info["path"] = path
    return info
