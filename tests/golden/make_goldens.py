#!/usr/bin/env python3
"""
Generate the golden fixtures under tests/golden/ by IMPORTING the real
reference (dj408/esc-gnn, mounted read-only at /root/reference) in the build
container and running its own functions on small seeded inputs.

Runs only in the build container (the reference never travels to the GPU box);
the resulting .npz files are data (inputs + expected outputs) and are committed.

Reference functions exercised (file:line under /root/reference):
  pyg_utilities.get_Batch_Wjxs                              pyg_utilities.py:283-560
  models.escgnn_modular.ESCGNNModular._scatter              models/escgnn_modular.py:309-362
  data_processing.process_pyg_data.process_pyg_data         data_processing/process_pyg_data.py:640-919
  data_processing.process_pyg_data.get_local_pca_kernel_weights / calc_O_ij   :79-138 / :454-493

`torch_geometric`, `torch_scatter`, `torchmetrics`, `h5py` are absent from this
image.  They are replaced by NAMES-ONLY stubs (empty base classes; functions
the exercised paths never call).  The only stand-ins that compute anything are
`to_torch_coo_tensor` (= torch.sparse_coo_tensor(...).coalesce(), its documented
meaning) and the kNN graph construction, which is done HERE (scipy cKDTree +
symmetrise + row-major sort) and handed to the reference as a precomputed
`edge_index` with graph_construction=None, exactly as
scripts/python/generate_ellipsoid_dataset.py:1167-1183 calls it.  Parity for
kNN tie order is therefore "unpinned" (SURVEY.md §8c).

Usage:  python tests/golden/make_goldens.py   (from the repo root)
"""
import os
import sys
import types
import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


# ---------------------------------------------------------------- stubs
def _install_stubs():
    class _DictData:
        """dict-like stand-in for torch_geometric.data.Data (names only)."""
        def __init__(self, **kw):
            self.__dict__.update(kw)

        def __getitem__(self, k):
            return self.__dict__[k]

        def __setitem__(self, k, v):
            self.__dict__[k] = v

        def keys(self):
            return [k for k in self.__dict__.keys()]

        def __iter__(self):
            return iter(self.__dict__.items())

        def to(self, *a, **k):
            return self

    class _Dataset:
        def __init__(self, *a, **k):
            pass

    def _nope(*a, **k):
        raise RuntimeError("stub: this function is not part of the exercised path")

    def to_torch_coo_tensor(edge_index, edge_attr=None, size=None):
        if edge_attr is None:
            edge_attr = torch.ones(edge_index.shape[1])
        return torch.sparse_coo_tensor(edge_index, edge_attr, size).coalesce()

    tg = types.ModuleType("torch_geometric")
    tgd = types.ModuleType("torch_geometric.data")
    tgu = types.ModuleType("torch_geometric.utils")
    tgn = types.ModuleType("torch_geometric.nn")
    tgl = types.ModuleType("torch_geometric.loader")
    for name in ("Data", "Batch"):
        setattr(tgd, name, _DictData)
    for name in ("Dataset", "InMemoryDataset"):
        setattr(tgd, name, _Dataset)
    tgu.to_torch_coo_tensor = to_torch_coo_tensor
    tgu.to_undirected = _nope
    tgn.radius_graph = _nope
    tgn.knn_graph = _nope
    tgl.DataLoader = _Dataset
    tg.data, tg.utils, tg.nn, tg.loader = tgd, tgu, tgn, tgl
    sys.modules.update({
        "torch_geometric": tg, "torch_geometric.data": tgd,
        "torch_geometric.utils": tgu, "torch_geometric.nn": tgn,
        "torch_geometric.loader": tgl,
    })
    ts = types.ModuleType("torch_scatter")
    ts.scatter = _nope
    sys.modules["torch_scatter"] = ts
    sys.modules["h5py"] = types.ModuleType("h5py")

    class _AnyAttr(types.ModuleType):
        def __getattr__(self, name):
            if name.startswith("__"):
                raise AttributeError(name)
            return type(name, (), {"__init__": lambda self, *a, **k: None})
    for m in ("torchmetrics", "torchmetrics.regression", "torchmetrics.classification"):
        sys.modules[m] = _AnyAttr(m)
    return _DictData


Data = _install_stubs()
sys.path.insert(0, REF)
import pyg_utilities as ref_pu                                    # noqa: E402
from data_processing import process_pyg_data as ref_ppd            # noqa: E402
from models import escgnn_modular as ref_mod                       # noqa: E402


# ---------------------------------------------------------------- inputs
def ellipsoid_points(rng, n, mu=(3.0, 1.0, 1.0), sd=(0.5, 0.2, 0.2)):
    """recipe of data_processing/ellipsoid_data_classes.py:112-131 (our own draw)."""
    a = abs(rng.normal(mu[0], sd[0])); b = abs(rng.normal(mu[1], sd[1])); c = abs(rng.normal(mu[2], sd[2]))
    phi = rng.uniform(0, 2 * np.pi, n)
    theta = np.arccos(2 * rng.uniform(0, 1, n) - 1)
    return np.column_stack([a * np.sin(theta) * np.cos(phi),
                            b * np.sin(theta) * np.sin(phi),
                            c * np.cos(theta)]).astype(np.float32)


def knn_sym_edges(pos, k):
    """kNN (no self loops) -> symmetrise -> unique, row-major sorted. OUR stand-in."""
    from scipy.spatial import cKDTree
    n = pos.shape[0]
    _, nb = cKDTree(pos.astype(np.float64)).query(pos.astype(np.float64), k=k + 1)
    src = np.repeat(np.arange(n), k)
    dst = nb[:, 1:].reshape(-1)
    key = np.unique(np.concatenate([src * n + dst, dst * n + src]))
    return np.stack([key // n, key % n]).astype(np.int64)


def run_ref_builder(pos, ei, eps, align="column_dot", recenter=False, dtype="float32", row_normalize=True,
                    kernel="gaussian", r_cut=None):
    data = Data(pos=torch.from_numpy(pos), edge_index=torch.from_numpy(ei), num_nodes=pos.shape[0])
    kw = {"kernel": kernel, "gaussian_eps": eps}
    if r_cut is not None:
        kw = {"kernel": kernel, "r_cut": r_cut}
    d = ref_ppd.process_pyg_data(
        data, data_i=0, row_normalize=row_normalize, return_data_object=False, vector_feat_key="pos",
        device="cpu", graph_construction=None, use_mean_recentering=recenter,
        local_pca_kernel_fn_kwargs=kw, sing_vect_align_method=align, hdf5_tensor_dtype=dtype)
    return d


def mean_edge_len(pos, ei):
    return float(np.linalg.norm(pos[ei[0]] - pos[ei[1]], axis=1).mean())


def block_diag_coo(parts):
    """stand-in for PyG's collate of sparse operators with __cat_dim__=(0,1)
    (models/escgnn_data_classes.py:20-27): concatenate with cumulative offsets."""
    idx, val, off = [], [], 0
    for (i, v, n) in parts:
        idx.append(i + off); val.append(v); off += n
    return np.concatenate(idx, axis=1), np.concatenate(val), off


# ---------------------------------------------------------------- G3/G4: builder
def make_builder_goldens():
    rng = np.random.RandomState(457892)
    out = {}
    cases = [(32, 5, "column_dot", False), (64, 5, "column_dot", False), (128, 5, "column_dot", False),
             (64, 10, "column_dot", False), (128, 10, "column_dot", True), (48, 5, "procrustes", False)]
    meta = []
    for ci, (n, k, align, recenter) in enumerate(cases):
        pos = ellipsoid_points(rng, n)
        ei = knn_sym_edges(pos, k)
        eps = mean_edge_len(pos, ei) ** 2
        d32 = run_ref_builder(pos, ei, eps, align, recenter, "float32")
        d16 = run_ref_builder(pos, ei, eps, align, recenter, "float16")
        p = f"g{ci}_"
        out[p + "pos"] = pos
        out[p + "edge_index"] = ei
        out[p + "eps"] = np.float64(eps)
        for op in ("P", "Q"):
            out[p + op + "_indices"] = d32[op]["indices"].numpy()
            out[p + op + "_values_f32"] = d32[op]["values"].numpy()
            out[p + op + "_values_f16"] = d16[op]["values"].numpy()
            out[p + op + "_size"] = np.array(tuple(d32[op]["size"]), dtype=np.int64)
        meta.append(f"{n},{k},{align},{int(recenter)}")
    out["cases"] = np.array(meta)
    # column-normalised P and an epanechnikov-kernel case on graph 0's points
    pos, ei = out["g0_pos"], out["g0_edge_index"]
    dcol = run_ref_builder(pos, ei, float(out["g0_eps"]), row_normalize=False)
    out["g0_colnorm_P_indices"] = dcol["P"]["indices"].numpy()
    out["g0_colnorm_P_values_f32"] = dcol["P"]["values"].numpy()
    out["g0_colnorm_Q_values_f32"] = dcol["Q"]["values"].numpy()
    dep = run_ref_builder(pos, ei, None, kernel="epanechnikov", r_cut=5.0)
    out["g0_epan_Q_values_f32"] = dep["Q"]["values"].numpy()
    # G4: kernel-weight spot values and calc_O_ij spot values
    r = torch.linspace(0.0, 3.0, 13)
    out["kw_r"] = r.numpy()
    out["kw_gauss_eps0p5_sqrt"] = ref_ppd.get_local_pca_kernel_weights(r, "gaussian", gaussian_eps=0.5).numpy()
    out["kw_gauss_none_nosqrt"] = ref_ppd.get_local_pca_kernel_weights(r, "gaussian", take_sqrt=False).numpy()
    out["kw_cos_rc2"] = ref_ppd.get_local_pca_kernel_weights(r, "cosine_cutoff", r_cut=2.0).numpy()
    out["kw_epan_rc2"] = ref_ppd.get_local_pca_kernel_weights(r, "epanechnikov", r_cut=2.0).numpy()
    g = torch.Generator().manual_seed(11)
    Oi = torch.linalg.qr(torch.randn(3, 3, generator=g))[0]
    Oj = torch.linalg.qr(torch.randn(3, 3, generator=g))[0]
    out["oij_Oi"], out["oij_Oj"] = Oi.numpy(), Oj.numpy()
    out["oij_column_dot"] = ref_ppd.calc_O_ij(Oi, Oj, True, "column_dot").numpy()
    out["oij_procrustes"] = ref_ppd.calc_O_ij(Oi, Oj, True, "procrustes").numpy()
    out["oij_nosign"] = ref_ppd.calc_O_ij(Oi, Oj, False).numpy()
    out["sq_idx_3"] = ref_ppd.get_square_matrix_coo_indices(3).numpy()
    np.savez_compressed(os.path.join(OUT, "builder_goldens.npz"), **out)
    return out


# ---------------------------------------------------------------- G1/G2/G5: scattering
def make_scatter_goldens(bg):
    out = {}
    # batch = block-diagonal of builder graphs 0 (n=32), 1 (n=64), 5 (n=48)
    use = [0, 1, 5]
    Pp = [(bg[f"g{g}_P_indices"], bg[f"g{g}_P_values_f32"], int(bg[f"g{g}_P_size"][0])) for g in use]
    Qp = [(bg[f"g{g}_Q_indices"], bg[f"g{g}_Q_values_f32"], int(bg[f"g{g}_Q_size"][0])) for g in use]
    Pi, Pv, N = block_diag_coo(Pp)
    Qi, Qv, N3 = block_diag_coo(Qp)
    assert N3 == 3 * N
    out["P_indices"], out["P_values"], out["N"] = Pi, Pv, np.int64(N)
    out["Q_indices"], out["Q_values"] = Qi, Qv
    out["graph_ptr"] = np.cumsum([0] + [p[2] for p in Pp]).astype(np.int64)
    # operators exactly as models/escgnn_data_classes.py:137-142 builds them (un-coalesced flag)
    P = torch.sparse_coo_tensor(torch.from_numpy(Pi), torch.from_numpy(Pv), (N, N), dtype=torch.float32)
    Q = torch.sparse_coo_tensor(torch.from_numpy(Qi), torch.from_numpy(Qv), (N3, N3), dtype=torch.float32)
    g = torch.Generator().manual_seed(2024)
    xs = {C: torch.randn(N, C, generator=g) for C in (1, 3, 5)}
    xv = torch.cat([torch.from_numpy(bg[f"g{g_}_pos"]) for g_ in use], 0)  # (N,3)
    for C, x in xs.items():
        out[f"x_C{C}"] = x.numpy()
    out["x_v"] = xv.numpy()

    modes = {
        "dyadic_J2": dict(scales_type="dyadic", J=2),
        "dyadic_J4": dict(scales_type="dyadic", J=4),
        "dyadic_J4_nolp": dict(scales_type="dyadic", J=4, include_lowpass=False),
        "dyadic_str_J3": dict(scales_type="dyadic", diffusion_scales="dyadic", J=3),
        "custom_0_16": dict(scales_type="custom", diffusion_scales=torch.tensor([0, 1, 2, 4, 6, 8, 16])),
        "custom_0_16_nolp": dict(scales_type="custom", diffusion_scales=torch.tensor([0, 1, 2, 4, 6, 8, 16]),
                                 include_lowpass=False),
        "custom_1_2_4": dict(scales_type="custom", diffusion_scales=torch.tensor([1, 2, 4])),
        "custom_dup": dict(scales_type="custom", diffusion_scales=torch.tensor([0, 1, 1, 3, 5])),
        "p1": dict(scales_type=None, diffusion_scales=None),
    }
    out["mode_names"] = np.array(list(modes.keys()))

    def run(x, op, kw, tag):
        xr = x.clone().requires_grad_(True)
        W = ref_pu.get_Batch_Wjxs(x=xr, P_sparse=op, **kw)
        gen = torch.Generator().manual_seed(7)
        gup = torch.randn(W.shape, generator=gen)
        (W * gup).sum().backward()
        out[tag + "_W"] = W.detach().numpy()
        out[tag + "_gup"] = gup.numpy()
        out[tag + "_gx"] = xr.grad.numpy()

    for name, kw in modes.items():
        for C, x in xs.items():
            run(x, P, kw, f"P_{name}_C{C}")
        run(xv.reshape(N3, 1), Q, kw, f"Q_{name}")
    # 2-D per-channel scales (C=3), J=3 -> runs all t=1..8 (pyg_utilities.py:392-444)
    ds2 = torch.tensor([[0, 1, 2, 4, 8], [0, 2, 3, 5, 8], [1, 2, 4, 6, 7]])
    out["perchan_scales"] = ds2.numpy()
    for lp in (True, False):
        run(xs[3], P, dict(scales_type="custom", diffusion_scales=ds2, J=3, include_lowpass=lp),
            f"P_perchan_lp{int(lp)}_C3")
    # coalesced COO and CSR inputs give the same answer (probe)
    Wc = ref_pu.get_Batch_Wjxs(x=xs[3], P_sparse=P.coalesce(), scales_type="dyadic", J=4)
    assert torch.equal(Wc, torch.from_numpy(out["P_dyadic_J4_C3_W"]))

    # G5: fp64 dense-power reference for dyadic J=4 (separates rounding from logic)
    Pd = P.to_dense().double(); Qd = Q.to_dense().double()

    def dense_bank(A, x, J):
        pw = [x.double()]
        cur = x.double()
        for t in range(1, 2 ** J + 1):
            cur = A @ cur
            if t & (t - 1) == 0:
                pw.append(cur)
        bands = [pw[j - 1] - pw[j] for j in range(1, J + 2)] + [pw[-1]]
        return torch.stack(bands, -1)
    out["f64_P_dyadic_J4_C3_W"] = dense_bank(Pd, xs[3], 4).numpy()
    out["f64_Q_dyadic_J4_W"] = dense_bank(Qd, xv.reshape(N3, 1), 4).numpy()

    # G2: _scatter (models/escgnn_modular.py:309-362) called unbound on a names-only self
    class _Self:
        ablate_second_order_wavelets = False
        _triu_indices_cache = {}
        _get_triu_pair_indices = ref_mod.ESCGNNModular._get_triu_pair_indices
    for abl in (False, True):
        s = _Self(); s.ablate_second_order_wavelets = abl; s._triu_indices_cache = {}
        kw = {"diffusion_kwargs": dict(scales_type="dyadic", J=4, include_lowpass=True, diffusion_scales="dyadic")}
        Ws = ref_mod.ESCGNNModular._scatter(s, track="scalar", x0=xs[3], P_or_Q=P, kwargs=kw)
        Wv = ref_mod.ESCGNNModular._scatter(s, track="vector", x0=xv, P_or_Q=Q, kwargs=kw)
        out[f"scatter_scalar_abl{int(abl)}"] = Ws.numpy()
        out[f"scatter_vector_abl{int(abl)}"] = Wv.contiguous().numpy()
    kw = {"diffusion_kwargs": dict(scales_type="custom", J=4, include_lowpass=True,
                                   diffusion_scales=torch.tensor([0, 1, 2, 4, 6, 8, 16]))}
    s = _Self(); s._triu_indices_cache = {}
    out["scatter_scalar_custom"] = ref_mod.ESCGNNModular._scatter(
        s, track="scalar", x0=xs[3], P_or_Q=P, kwargs=kw).numpy()
    np.savez_compressed(os.path.join(OUT, "scatter_goldens.npz"), **out)
    return out


if __name__ == "__main__":
    torch.manual_seed(0)
    torch.set_num_threads(1)
    bg = make_builder_goldens()
    sg = make_scatter_goldens(bg)
    for f in ("builder_goldens.npz", "scatter_goldens.npz"):
        print(f, os.path.getsize(os.path.join(OUT, f)) // 1024, "KiB")
    print("scalar W", sg["P_dyadic_J4_C3_W"].shape, "vector W", sg["Q_dyadic_J4_W"].shape,
          "scatter", sg["scatter_scalar_abl0"].shape, sg["scatter_vector_abl0"].shape)
