"""
GPU parity tests of the P / Q operator build (esc_gnn_amd/process_pyg_data.py -> escat_build_* in libescat.so)
against the goldens generated from the reference's process_pyg_data and against oracle/builder_oracle.py.

Index structure (P CSR pattern, Q block pattern and emission order) and P values are compared BIT-EXACT.
Q values: the reference takes O_i from a fp32 LAPACK SVD of C_i, the kernel from a fp64 Jacobi eigensolver of
C_i C_i^T; both are exact up to rounding, but where two singular values nearly coincide the frame is ill-conditioned
(SURVEY.md §7), so blocks are compared with a gap-aware tolerance and globally through the invariants.
"""
import os
import numpy as np
import pytest
import torch

from oracle import builder_oracle as bo

pytestmark = pytest.mark.gpu


def _dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _cases(bg):
    for ci, c in enumerate(bg["cases"]):
        n, k, align, rec = str(c).split(",")
        yield ci, int(n), int(k), align, bool(int(rec))


class _Data(dict):
    __getattr__ = dict.get


def _gaps(pos, ei, eps, rec):
    """smallest relative singular-value gap per node (conditioning of its frame)."""
    O = bo.local_frames(pos, ei, {"kernel": "gaussian", "gaussian_eps": eps}, rec)
    v = torch.from_numpy(pos)
    gaps = np.ones(pos.shape[0])
    nb = {}
    for s, t in zip(ei[0].tolist(), ei[1].tolist()):
        nb.setdefault(s, []).append(t)
    for i, l in nb.items():
        V = v[l] - (v[[i] + l].mean(0) if rec else v[i])
        w = bo.kernel_weights(V.norm(dim=1), "gaussian", gaussian_eps=eps)
        s = torch.linalg.svdvals((w[:, None] * V).T.double())
        gaps[i] = float(min(s[0] - s[1], s[1] - s[2]) / s[0])
    return gaps


def test_process_pyg_data_vs_goldens(bg):
    from esc_gnn_amd.process_pyg_data import process_pyg_data
    dev = _dev()
    worst = 0.0
    for ci, n, k, align, rec in _cases(bg):
        p = f"g{ci}_"
        eps = float(bg[p + "eps"])
        data = _Data(pos=torch.from_numpy(bg[p + "pos"]), edge_index=torch.from_numpy(bg[p + "edge_index"]), num_nodes=n)
        for dt in ("float32", "float16"):
            out = process_pyg_data(data, data_i=ci, return_data_object=False, vector_feat_key="pos", device="cuda",
                                   graph_construction=None, use_mean_recentering=rec,
                                   local_pca_kernel_fn_kwargs={"kernel": "gaussian", "gaussian_eps": eps},
                                   sing_vect_align_method=align, hdf5_tensor_dtype=dt)
            assert out["original_idx"] == ci
            assert np.array_equal(out["P"]["indices"].cpu().numpy(), bg[p + "P_indices"])            # bit-exact structure
            assert np.array_equal(out["Q"]["indices"].cpu().numpy(), bg[p + "Q_indices"])
            assert tuple(out["P"]["size"]) == tuple(bg[p + "P_size"]) and tuple(out["Q"]["size"]) == tuple(bg[p + "Q_size"])
            suffix = "f32" if dt == "float32" else "f16"
            assert np.array_equal(out["P"]["values"].cpu().numpy(), bg[p + "P_values_" + suffix])    # bit-exact values
            qv = out["Q"]["values"].cpu().numpy().astype(np.float32).reshape(-1, 3, 3)
            ref = bg[p + "Q_values_" + suffix].astype(np.float32).reshape(-1, 3, 3)
            if align == "procrustes":
                continue        # depends on the SVD's arbitrary column signs: checked in test_procrustes_formula
            if dt == "float32":
                gaps = _gaps(bg[p + "pos"], bg[p + "edge_index"], eps, rec)
                Pi = bg[p + "P_indices"]
                g = np.minimum(gaps[Pi[0]], gaps[Pi[1]])
                err = np.abs(qv - ref).max(axis=(1, 2))
                worst = max(worst, float((err * g).max()))
                assert (err <= 2e-6 / np.maximum(g, 1e-6) + 2e-6).all(), (ci, float((err * g).max()))
            else:
                assert np.abs(qv - ref).max() <= 2e-3
    print("worst gap-scaled Q error", worst)


def test_procrustes_formula(bg):
    """the reference's 'procrustes' alignment (process_pyg_data.py:315-327) is NOT invariant to the arbitrary column
    signs LAPACK picks for O_i, so goldens cannot pin it for a different eigensolver; pin the kernel's formula instead:
    with the kernel's own frames, blocks must equal the oracle's literal U V^T computation."""
    from esc_gnn_amd.process_pyg_data import frames_and_Q, _get_P_csr
    dev = _dev()
    pos = torch.from_numpy(bg["g5_pos"]).to(dev)
    rowptr, col, pval, _ = _get_P_csr(torch.from_numpy(bg["g5_edge_index"]), 48, None, True, dev)
    for method in ("procrustes", "column_dot"):
        frames, blk = frames_and_Q(rowptr, col, pval, pos, {"kernel": "gaussian", "gaussian_eps": float(bg["g5_eps"])},
                                   sing_vect_align_method=method)
        F, B = frames.cpu(), blk.cpu()
        rp = rowptr.cpu().numpy()
        for i in range(48):
            for e in range(rp[i], rp[i + 1]):
                j = int(col[e])
                ref = float(pval[e]) * bo.calc_O_ij(F[i], F[j], True, method)
                assert (B[e] - ref).abs().max() < 2e-5, (method, i, j)
        assert (torch.einsum("nac,nbc->nab", F, F) - torch.eye(3)).abs().max() < 1e-6       # frames orthonormal


def test_other_kernels_and_colnorm(bg):
    from esc_gnn_amd.process_pyg_data import process_pyg_data
    _dev()
    data = _Data(pos=torch.from_numpy(bg["g0_pos"]), edge_index=torch.from_numpy(bg["g0_edge_index"]), num_nodes=32)
    out = process_pyg_data(data, return_data_object=False, graph_construction=None, hdf5_tensor_dtype="float32",
                           local_pca_kernel_fn_kwargs={"kernel": "epanechnikov", "r_cut": 5.0})
    assert np.abs(out["Q"]["values"].cpu().numpy() - bg["g0_epan_Q_values_f32"]).max() <= 1e-4
    out = process_pyg_data(data, return_data_object=False, graph_construction=None, hdf5_tensor_dtype="float32",
                           row_normalize=False,
                           local_pca_kernel_fn_kwargs={"kernel": "gaussian", "gaussian_eps": float(bg["g0_eps"])})
    assert np.array_equal(out["P"]["indices"].cpu().numpy(), bg["g0_colnorm_P_indices"])
    assert np.array_equal(out["P"]["values"].cpu().numpy(), bg["g0_colnorm_P_values_f32"])
    assert np.abs(out["Q"]["values"].cpu().numpy() - bg["g0_colnorm_Q_values_f32"]).max() <= 1e-4
    with pytest.raises(NotImplementedError):
        process_pyg_data(data, graph_construction="delaunay")
    with pytest.raises(ValueError):
        process_pyg_data(data, graph_construction=None, local_pca_kernel_fn_kwargs={"kernel": "cosine_cutoff"})
    # isolated node -> KeyError like the reference
    ei = torch.from_numpy(bg["g0_edge_index"])
    keep = (ei[0] != 5) & (ei[1] != 5)
    with pytest.raises(KeyError):
        process_pyg_data(_Data(pos=data["pos"], edge_index=ei[:, keep], num_nodes=32), return_data_object=False,
                         graph_construction=None, local_pca_kernel_fn_kwargs={"kernel": "gaussian", "gaussian_eps": 1.0})


def test_batched_knn_builder(bg):
    """kNN + symmetrise + P on the GPU for a batch of ragged graphs == per-graph oracle (pattern and P bit-exact),
    Q blocks satisfy the invariants of SURVEY.md §4 and feed get_Batch_Wjxs directly."""
    from esc_gnn_amd.process_pyg_data import build_PQ_batch, coo_parts
    from esc_gnn_amd import get_Batch_Wjxs, synthetic
    dev = _dev()
    sizes = [32, 64, 128, 64, 128, 48, 500, 7, 200]
    rng = np.random.RandomState(5)
    clouds = [synthetic.ellipsoid_clouds(1, n, seed=int(rng.randint(1 << 30)))[0] for n in sizes]
    for k in (5, 10):
        use = [c for c in clouds if c.shape[0] > k]
        pos = np.concatenate(use)
        gp = np.cumsum([0] + [c.shape[0] for c in use]).astype(np.int32)
        b = build_PQ_batch(torch.from_numpy(pos).to(dev), torch.from_numpy(gp), k, {"kernel": "gaussian", "gaussian_eps": 0.05})
        rp, col, pv = b["rowptr"].cpu().numpy(), b["col"].cpu().numpy(), b["pval"].cpu().numpy()
        assert rp[-1] == b["nnz"] == col.shape[0]
        for g, c in enumerate(use):
            ei = bo.knn_sym_edges(c, k)
            Pi, Pv = bo.get_P_coo(ei, c.shape[0])
            parts = coo_parts(b, g, "float32")
            assert np.array_equal(parts["P"]["indices"].cpu().numpy(), Pi.numpy()), (k, g)
            assert np.array_equal(parts["P"]["values"].cpu().numpy(), Pv.numpy())
            Qv = parts["Q"]["values"].cpu().numpy().reshape(-1, 3, 3)
            R = Qv / Pv.numpy()[:, None, None]
            assert np.abs(np.einsum("eab,ecb->eac", R, R) - np.eye(3)).max() < 1e-4          # blocks / p_ij are orthogonal
            diag = (Pi[0] == Pi[1]).numpy()
            assert np.abs(R[diag] - np.eye(3)).max() < 1e-5                                    # Q_ii = 1/2 I
            # Q_ji / p_ji = (Q_ij / p_ij)^T
            n = c.shape[0]
            key = (Pi[0] * n + Pi[1]).numpy()
            tkey = (Pi[1] * n + Pi[0]).numpy()
            order = np.argsort(key)
            tpos = order[np.searchsorted(key[order], tkey)]
            assert np.abs(R[tpos] - R.transpose(0, 2, 1)).max() < 1e-4
        # the batch feeds the scattering path with no ingest step
        N = pos.shape[0]
        W = get_Batch_Wjxs(torch.from_numpy(pos).to(dev).reshape(-1, 1), b["Q"], J=3)
        assert W.shape == (3 * N, 1, 5) and torch.isfinite(W).all()
        assert (W.sum(-1).reshape(-1) - torch.from_numpy(pos).to(dev).reshape(-1)).abs().max() < 1e-5


def test_builder_vs_oracle_Q_blocks(bg):
    """full Q of a fresh seeded graph vs the numpy/torch restatement (gap-aware)."""
    from esc_gnn_amd.process_pyg_data import build_PQ_batch
    from esc_gnn_amd import synthetic
    dev = _dev()
    c = synthetic.ellipsoid_clouds(1, 150, seed=4242)[0]
    ei = bo.knn_sym_edges(c, 8)
    eps = float(np.linalg.norm(c[ei[0]] - c[ei[1]], axis=1).mean() ** 2)
    ref = bo.build_PQ(c, ei, {"kernel": "gaussian", "gaussian_eps": eps})
    b = build_PQ_batch(torch.from_numpy(c).to(dev), torch.tensor([0, 150], dtype=torch.int32), 8,
                       {"kernel": "gaussian", "gaussian_eps": eps})
    gaps = _gaps(c, ei, eps, False)
    Pi = ref["P"]["indices"]
    g = np.minimum(gaps[Pi[0]], gaps[Pi[1]])
    err = np.abs(b["blk"].cpu().numpy() - ref["Q"]["values"].reshape(-1, 3, 3)).max(axis=(1, 2))
    assert (err <= 2e-6 / np.maximum(g, 1e-6) + 2e-6).all(), float((err * g).max())


def test_dataset_to_hdf5_and_back(tmp_path, bg):
    """cfg5 in small: build a ragged dataset on the GPU, write the reference's HDF5 layout, read it back the way
    models/escgnn_data_classes.py:135-143 does and run the scattering chain on the loaded operators."""
    from esc_gnn_amd.process_pyg_data import write_dataset_hdf5, process_pyg_data
    from esc_gnn_amd.hdf5_min import H5Reader
    from esc_gnn_amd import get_Batch_Wjxs, synthetic
    dev = _dev()
    sizes = [32, 64, 48, 128, 20]
    clouds = [synthetic.ellipsoid_clouds(1, n, seed=100 + n)[0] for n in sizes]
    pos = torch.from_numpy(np.concatenate(clouds)).to(dev)
    gp = torch.from_numpy(np.cumsum([0] + sizes).astype(np.int32))
    kw = {"kernel": "gaussian", "gaussian_eps": 0.1}
    path = str(tmp_path / "pq.h5")
    info = write_dataset_hdf5(path, pos, gp, 5, kw, hdf5_tensor_dtype="float16", chunk_graphs=2)
    assert info["graphs"] == 5
    f = H5Reader(path)
    assert set(f.keys()) == {"P", "Q"} and set(f["P"].keys()) == {str(i) for i in range(5)}
    per_graph_W = []
    for g, c in enumerate(clouds):
        ei = bo.knn_sym_edges(c, 5)
        one = process_pyg_data(_Data(pos=torch.from_numpy(c), edge_index=torch.from_numpy(ei), num_nodes=c.shape[0]),
                               return_data_object=False, graph_construction=None, local_pca_kernel_fn_kwargs=kw,
                               hdf5_tensor_dtype="float16")
        for op in ("P", "Q"):
            grp = f[op][str(g)]
            assert np.array_equal(grp["indices"][...], one[op]["indices"].cpu().numpy())
            assert grp["values"].dtype == np.float16
            assert np.array_equal(grp["values"][...], one[op]["values"].cpu().numpy())
            assert tuple(grp["size"][...].tolist()) == tuple(one[op]["size"])
        # the reference loader's construction (fp32 sparse COO from fp16 storage), then scattering on the GPU
        grp = f["Q"][str(g)]
        Q = torch.sparse_coo_tensor(torch.from_numpy(grp["indices"][...].copy()), torch.from_numpy(grp["values"][...].copy()),
                                    tuple(grp["size"][...].tolist()), dtype=torch.float32).to(dev)
        xv = torch.from_numpy(c).to(dev).reshape(-1, 1)
        W = get_Batch_Wjxs(xv, Q, J=2)
        ref = __import__("oracle.chain_oracle", fromlist=["x"]).ref_get_Batch_Wjxs(xv.cpu(), Q.cpu(), J=2)
        assert (W.cpu() - ref).abs().max() <= 1e-5 * ref.abs().max()
        per_graph_W.append(W)
    f.close()
    # the same file as an operator store resident in HBM: shuffled mini-batches are views of it
    from esc_gnn_amd.store import load_operator_store
    store = load_operator_store(path, dev)
    assert store["graph_ptr"].tolist() == gp.tolist() and store["Q"].bsr3().n_parts == 5 and store["P"].csr().n_parts == 5
    order = [3, 0, 4, 1, 2]
    vQ, vP = store["Q"].select(order), store["P"].select(order)
    xv_all = pos.reshape(-1, 1)
    Wv = get_Batch_Wjxs(xv_all[vQ.row_index()], vQ, J=2)
    want = torch.cat([per_graph_W[g] for g in order])
    assert (Wv - want).abs().max() <= 1e-6 * want.abs().max()
    Wp = get_Batch_Wjxs(pos[vP.row_index()], vP, J=2)
    Wp_full = get_Batch_Wjxs(pos, store["P"], J=2)
    assert torch.equal(Wp, Wp_full[vP.row_index()])


def test_full_size_rotation_equivariance():
    """size-independent property at BASELINE cfg2's full size (1024 graphs x 128 nodes, k=10), through builder ->
    scattering -> invariants: building on a rotated/reflected cloud gives Q' = R Q R^T (frames pick up R and a column
    sign that the reference's column_dot alignment cancels, process_pyg_data.py:306-313), hence vector coefficients
    W' = R W and unchanged invariants for orders 0 and 1; P and the scalar track do not move at all.  (Second-order
    coefficients are component-wise products W1[i]*W1[j] in the reference, models/escgnn_modular.py:507-516, which no
    rotation commutes with, so they are only checked to extend the order-0/1 columns unchanged.)
    Frames are ill-conditioned where singular values nearly coincide, so the bound is on quantiles."""
    from esc_gnn_amd import synthetic
    from esc_gnn_amd.process_pyg_data import build_PQ_batch
    from esc_gnn_amd.scatter import scatter
    from esc_gnn_amd.invariants import get_vector_invariants
    dev = _dev()
    B, n, k = 1024, 128, 10
    clouds = synthetic.ellipsoid_clouds(B, n, seed=77)
    pos = torch.from_numpy(clouds.reshape(-1, 3)).to(dev)
    gp = torch.arange(B + 1, dtype=torch.int32, device=dev) * n
    g = torch.Generator().manual_seed(9)
    Rm, _ = torch.linalg.qr(torch.randn(3, 3, generator=g, dtype=torch.float64))
    Rm[:, 0] *= -torch.sign(torch.linalg.det(Rm))              # det = -1: a reflection, the harder O(3) case
    Rm = Rm.float().to(dev)
    kw = {"kernel": "gaussian", "gaussian_eps": 0.05}
    skw = {"diffusion_kwargs": dict(scales_type="dyadic", J=3, include_lowpass=True)}
    outs = []
    for p in (pos, pos @ Rm.T):
        b = build_PQ_batch(p, gp, k, kw)
        pc = b["P"].csr(); rp = pc.rowptr.long()
        rows = torch.repeat_interleave(torch.arange(pc.n, device=dev), rp[1:] - rp[:-1]); cols = pc.col.long()
        ei = torch.stack([cols[rows != cols], rows[rows != cols]])
        Ws = scatter(track="scalar", x0=pos.norm(dim=1, keepdim=True), P_or_Q=b["P"], kwargs=skw)
        Wv = scatter(track="vector", x0=p, P_or_Q=b["Q"], kwargs=skw, ablate_second_order_wavelets=True)   # (N, 1, 3, 1+F)
        W2 = scatter(track="vector", x0=p, P_or_Q=b["Q"], kwargs=skw)
        assert W2.shape[-1] == 16 and torch.equal(W2[..., :Wv.shape[-1]], Wv)
        inv = get_vector_invariants(Wv.permute(0, 3, 1, 2).contiguous(), ei, ("percentiles", "min", "max", "mean"), 0.2)
        outs.append((pc.col.clone(), pc.val.clone(), Ws, Wv, inv))
    (c0, v0, Ws0, Wv0, inv0), (c1, v1, Ws1, Wv1, inv1) = outs
    # kNN distances move by rounding under R: a handful of near-tie neighbours may swap, nothing more
    same = (c0.shape == c1.shape) and bool((c0 == c1).all())
    if same:
        assert (v0 - v1).abs().max() <= 1e-6
        assert (Ws0 - Ws1).abs().max() <= 1e-5 * Ws0.abs().max()
    scale = float(Wv0.abs().max())
    err = (torch.einsum("ab,ncbw->ncaw", Rm, Wv0) - Wv1).abs().amax(dim=(1, 2, 3)) / scale
    q = torch.quantile(err, torch.tensor([0.5, 0.99], device=dev)).tolist()
    ierr = (inv0 - inv1).abs().amax(dim=tuple(range(1, inv0.dim()))) / float(inv0.abs().max())
    qi = torch.quantile(ierr, torch.tensor([0.5, 0.99], device=dev)).tolist()
    print("pattern identical", same, "W err quantiles", q, "max", float(err.max()), "inv", qi, float(ierr.max()))
    assert q[0] <= 2e-6 and q[1] <= 1e-4
    assert qi[0] <= 1e-5 and qi[1] <= 1e-3


# ---------------------------------------------------------------------------------------------- round 2: A5 branches
def _r2():
    import os
    return np.load(os.path.join(os.path.dirname(__file__), "golden", "round2_goldens.npz"))


def _blocks(values):
    return values.astype(np.float32).reshape(-1, 3, 3)


def test_tikhonov_and_low_degree_nodes_vs_reference_goldens():
    """rank_deficiency_strategy='Tikhonov' (process_pyg_data.py:262-279) on a graph with degree-1 and degree-2 nodes;
    goldens from the reference (tests/golden/make_round2_goldens.py).  P bit-exact.  Q: a node with >= 2 neighbours has
    a frame that is determined up to column signs (which the column-dot alignment cancels), so blocks between two such
    nodes are compared with the gap-aware bound; a degree-1 node's frame is its neighbour direction completed by an
    ARBITRARY basis of the degenerate plane (LAPACK's choice in the reference, the Jacobi solver's here), so for blocks
    touching such a node the invariants are checked instead: Q_ij / p_ij orthogonal, and it maps the neighbour
    direction's frame component consistently (|first-column overlap| = 1)."""
    from esc_gnn_amd.process_pyg_data import process_pyg_data
    _dev()
    g = _r2()
    pos, ei, eps = g["low_pos"], g["low_ei"], float(g["low_eps"])
    n = pos.shape[0]
    deg = np.bincount(ei[0], minlength=n)
    assert deg.min() == 1 and (deg == 2).sum() >= 4
    for tag, strat in (("tik", "Tikhonov"), ("plain", None)):
        data = _Data(pos=torch.from_numpy(pos), edge_index=torch.from_numpy(ei), num_nodes=n)
        out = process_pyg_data(data, data_i=0, return_data_object=False, graph_construction=None,
                               local_pca_kernel_fn_kwargs={"kernel": "gaussian", "gaussian_eps": eps},
                               rank_deficiency_strategy=strat, tikhonov_eps=1e-3, hdf5_tensor_dtype="float32")
        assert np.array_equal(out["P"]["indices"].cpu().numpy(), g[f"low_{tag}_P_indices"])
        assert np.array_equal(out["P"]["values"].cpu().numpy(), g[f"low_{tag}_P_values"])
        assert np.array_equal(out["Q"]["indices"].cpu().numpy(), g[f"low_{tag}_Q_indices"])
        Pi = g[f"low_{tag}_P_indices"]
        pv = g[f"low_{tag}_P_values"]
        ours, ref = _blocks(out["Q"]["values"].cpu().numpy()), _blocks(g[f"low_{tag}_Q_values"])
        both = (deg[Pi[0]] >= 2) & (deg[Pi[1]] >= 2)
        assert both.sum() > 20
        # frames of degree >= 2 nodes: compare (degree-2 nodes have singular values s1 > s2 > 0 = s3: well separated here)
        assert np.abs(ours[both] - ref[both]).max() <= 5e-5, float(np.abs(ours[both] - ref[both]).max())
        R = ours / pv[:, None, None]
        assert np.abs(np.einsum("eab,ecb->eac", R, R) - np.eye(3)).max() <= 1e-5          # every block: p_ij * orthogonal
        Rr = ref / pv[:, None, None]
        assert np.abs(np.einsum("eab,ecb->eac", Rr, Rr) - np.eye(3)).max() <= 1e-5


@pytest.mark.parametrize("tag,gc,kw", [("knn", "k-nn", {"r_cutoff": 5.0, "gaussian_eps": 4.0, "max_num_neighbors": 4}),
                                       ("radius", "radius", {"r_cutoff": 1.1, "gaussian_eps": 4.0, "max_num_neighbors": 32})])
def test_graph_construction_paths_vs_reference_goldens(tag, gc, kw):
    """the DEFAULT graph_construction='k-nn' path with its Gaussian-RBF x cosine-cutoff edge weights
    (process_pyg_data.py:173-232) and 'radius'; goldens from the reference with scipy stand-ins for the absent
    torch_geometric graph constructors (neighbour ties / cap order: parity unpinned)."""
    from esc_gnn_amd.process_pyg_data import process_pyg_data
    _dev()
    g = _r2()
    pos = g["gc_pos"]
    data = _Data(pos=torch.from_numpy(pos), num_nodes=pos.shape[0])
    out = process_pyg_data(data, data_i=3, return_data_object=False, graph_construction=gc, graph_construction_kwargs=kw,
                           local_pca_kernel_fn_kwargs={"kernel": "epanechnikov", "r_cut": 5.0}, hdf5_tensor_dtype="float32")
    assert out["original_idx"] == 3
    assert np.array_equal(out["edge_index"].numpy(), g[f"gc_{tag}_edge_index"])
    assert np.abs(out["edge_weight"].numpy() - g[f"gc_{tag}_edge_weight"]).max() <= 2e-7
    assert np.array_equal(out["P"]["indices"].cpu().numpy(), g[f"gc_{tag}_P_indices"])
    assert np.array_equal(out["Q"]["indices"].cpu().numpy(), g[f"gc_{tag}_Q_indices"])
    assert np.abs(out["P"]["values"].cpu().numpy() - g[f"gc_{tag}_P_values"]).max() <= 2e-7      # weighted degrees: fp32 sum order
    ours, ref = _blocks(out["Q"]["values"].cpu().numpy()), _blocks(g[f"gc_{tag}_Q_values"])
    # gap-aware, as in test_process_pyg_data_vs_goldens: per-node frame conditioning from the oracle's singular values
    ei = g[f"gc_{tag}_edge_index"]
    v = torch.from_numpy(pos)
    gaps = np.ones(pos.shape[0])
    for i in range(pos.shape[0]):
        nb = ei[1][ei[0] == i]
        V = v[nb] - v[i]
        w = bo.kernel_weights(V.norm(dim=1), "epanechnikov", r_cut=5.0)
        s = torch.cat([torch.linalg.svdvals((w[:, None] * V).T.double()), torch.zeros(3, dtype=torch.float64)])[:3]
        # (a node with one neighbour has an arbitrary frame in the degenerate plane: gap 0, its blocks are skipped)
        gaps[i] = float(min(s[0] - s[1], s[1] - s[2]) / s[0]) if nb.size >= 2 else 0.0
    Pi = g[f"gc_{tag}_P_indices"]
    gg = np.minimum(gaps[Pi[0]], gaps[Pi[1]])
    err = np.abs(ours - ref).max(axis=(1, 2))
    ok = gg > 0
    assert ok.sum() > 0.8 * ok.size
    assert (err[ok] <= 4e-6 / np.maximum(gg[ok], 1e-6) + 4e-6).all(), float((err[ok] * gg[ok]).max())
    # and the data-object form: an ESCGNNData with sparse P / Q, the new edges and the canonical operators attached
    from esc_gnn_amd.data_classes import ESCGNNData
    from esc_gnn_amd import get_Batch_Wjxs
    obj = process_pyg_data(_Data(pos=torch.from_numpy(pos), num_nodes=pos.shape[0]), graph_construction=gc,
                           graph_construction_kwargs=kw, local_pca_kernel_fn_kwargs={"kernel": "epanechnikov", "r_cut": 5.0})
    assert isinstance(obj, ESCGNNData) and obj.__cat_dim__("Q", obj.Q) == (0, 1)
    assert obj.P.is_sparse and obj.Q.is_sparse and tuple(obj.Q.shape) == (3 * pos.shape[0],) * 2
    assert np.array_equal(obj.edge_index.cpu().numpy(), g[f"gc_{tag}_edge_index"])
    W = get_Batch_Wjxs(torch.from_numpy(pos).cuda().reshape(-1, 1), obj.Q, J=2)
    assert W.shape == (3 * pos.shape[0], 1, 4) and torch.isfinite(W).all()


# ------------------------------------------------------------------ clouds beyond the LDS bitmap: the grid builder
def _grid_P(pos_np, k):
    from esc_gnn_amd import _lib
    from esc_gnn_amd.process_pyg_data import _knn_P_grid_batch
    dev = torch.device("cuda:0")
    pos = torch.from_numpy(np.ascontiguousarray(pos_np, dtype=np.float32)).to(dev)
    rp, co, va = _knn_P_grid_batch(pos, [0, pos.shape[0]], k, _lib.load(), dev)
    return rp.cpu().numpy(), co.cpu().numpy(), va.cpu().numpy()


@pytest.mark.parametrize("n,k", [(900, 10), (700, 16), (64, 5), (5, 8), (1, 3)])
def test_grid_builder_equals_bitmap_builder(n, k):
    """escat_build_knn_P_grid (uniform grid in HBM, any size) against escat_build_knn_P (brute force + LDS bitmap) on
    clouds both can take: rowptr, col and the lazy-walk values bit for bit (same fp64 distances, same tie rule)."""
    from esc_gnn_amd import synthetic
    from esc_gnn_amd.process_pyg_data import build_PQ_batch
    dev = torch.device("cuda:0")
    pos = synthetic.ellipsoid_clouds(1, n, seed=31 + n)[0] if n >= 5 else np.random.RandomState(n).randn(n, 3).astype(np.float32)
    pos = np.ascontiguousarray(pos, dtype=np.float32).reshape(n, 3)
    rp, co, va = _grid_P(pos, k)
    if n > k:
        b = build_PQ_batch(torch.from_numpy(pos).to(dev), torch.tensor([0, n], dtype=torch.int32), k, {'kernel': 'gaussian'})
        assert np.array_equal(rp, b["rowptr"].cpu().numpy()) and np.array_equal(co, b["col"].cpu().numpy())
        assert np.array_equal(va, b["pval"].cpu().numpy())
    if n <= k:                       # fewer than k other points: every point is linked to all others
        assert np.array_equal(np.diff(rp), np.full(n, n)) and np.array_equal(co, np.tile(np.arange(n), n))


@pytest.mark.parametrize("shape", ["ellipsoid", "cube", "line", "duplicates"])
def test_grid_builder_large_clouds_vs_kdtree(shape):
    """30 000-point clouds (beyond the bitmap builder) against scipy's cKDTree + the host symmetrisation of
    synthetic.knn_sym_csr: identical CSR structure and values -- on a surface, in a volume, on a line (two degenerate
    axes) and with repeated points (ties resolved by index: compared through distances there)."""
    from esc_gnn_amd import synthetic
    n, k = 30000, 16
    rng = np.random.RandomState(5)
    if shape == "ellipsoid":
        pos = synthetic.single_cloud(n, 2, workers=1)["pos"]
    elif shape == "cube":
        pos = rng.uniform(-1, 1, (n, 3)).astype(np.float32)
    elif shape == "line":
        pos = np.zeros((n, 3), np.float32); pos[:, 1] = rng.permutation(n).astype(np.float32) * 0.25 + 3.0
    else:
        pos = rng.uniform(-1, 1, (n // 2, 3)).astype(np.float32); pos = np.concatenate([pos, pos])[rng.permutation(n)]
    rp, co, va = _grid_P(pos, k)
    assert rp[0] == 0 and rp[-1] == co.shape[0] and np.all(np.diff(rp) >= k + 1)
    rows = np.repeat(np.arange(n), np.diff(rp))
    # sorted, unique, symmetric, diagonal present
    key = rows.astype(np.int64) * n + co
    assert np.all(np.diff(key) > 0)
    assert np.array_equal(np.sort(co.astype(np.int64) * n + rows), key)
    assert np.all(np.isin(np.arange(n, dtype=np.int64) * (n + 1), key))
    assert np.array_equal(va, synthetic.lazy_walk_values(rp.astype(np.int64), co.astype(np.int64)))
    if shape != "duplicates":
        rp_ref, co_ref = synthetic.knn_sym_csr(pos, k, workers=1)
        assert np.array_equal(rp, rp_ref) and np.array_equal(co, co_ref)
    else:
        # with exact ties the kd-tree's choice among equidistant points is its own; compare the k-th distances instead
        from scipy.spatial import cKDTree
        p64 = pos.astype(np.float64)
        dk, _ = cKDTree(p64).query(p64, k=k + 1)
        d = np.linalg.norm(p64[rows] - p64[co], axis=1)
        # every point's row holds at least its k nearest: the (k)-th smallest off-diagonal distance in the row <= kd-tree's
        order = np.lexsort((d, rows))
        dsorted = d[order]
        kth = dsorted[rp[:-1] + k]                    # position 0 is the diagonal (d = 0)... duplicates also have d = 0
        assert np.all(kth <= dk[:, k] + 1e-12)


def test_grid_builder_through_build_PQ_batch():
    """build_PQ_batch takes the grid route by itself for clouds beyond ~1090 nodes (two clouds here: global column
    ids, one rowptr) and the result runs through the scattering chain (row-stochastic P: the bands of a constant vanish)."""
    from esc_gnn_amd import synthetic, get_Batch_Wjxs
    from esc_gnn_amd.process_pyg_data import build_PQ_batch
    dev = torch.device("cuda:0")
    pos = np.concatenate([synthetic.single_cloud(1500, 2, workers=1)["pos"], synthetic.single_cloud(2100, 2, seed=3, workers=1)["pos"] + 10.0])
    gp = torch.tensor([0, 1500, 3600], dtype=torch.int32)
    b = build_PQ_batch(torch.from_numpy(pos).to(dev), gp, 12, {'kernel': 'gaussian'})
    rp, co = b["rowptr"].cpu().numpy(), b["col"].cpu().numpy()
    ref = [synthetic.knn_sym_csr(pos[a:z], 12, workers=1) for a, z in ((0, 1500), (1500, 3600))]
    assert np.array_equal(rp, np.concatenate([ref[0][0][:-1], ref[1][0] + ref[0][0][-1]]))
    assert np.array_equal(co, np.concatenate([ref[0][1], ref[1][1] + 1500]))
    x = torch.ones(3600, 2, device=dev)
    W = get_Batch_Wjxs(x, b["P"], J=3)
    assert W[..., :-1].abs().max() < 1e-5 and (W[..., -1] - 1).abs().max() < 1e-5
    assert b["blk"].shape == (co.shape[0], 3, 3) and torch.isfinite(b["blk"]).all()


def test_sharded_dataset_build_equals_single_file(tmp_path):
    """SURVEY 8e, operator build over ranks: two ranks' shards (built one after the other here, global sample indices,
    no exchange) merged by rank 0 hold exactly the datasets of the one-process build."""
    from esc_gnn_amd.process_pyg_data import write_dataset_hdf5, write_dataset_hdf5_sharded
    from esc_gnn_amd.hdf5_min import open_h5
    from esc_gnn_amd import synthetic
    dev = _dev()
    sizes = [40, 33, 64, 21, 57, 48, 30]
    pos = torch.from_numpy(np.concatenate([synthetic.ellipsoid_clouds(1, n, seed=7 + n)[0] for n in sizes])).to(dev)
    gp = torch.from_numpy(np.cumsum([0] + sizes).astype(np.int32))
    kw = {"kernel": "gaussian", "gaussian_eps": 0.1}
    one, many = str(tmp_path / "one.h5"), str(tmp_path / "many.h5")
    write_dataset_hdf5(one, pos, gp, 5, kw, chunk_graphs=3)
    i1 = write_dataset_hdf5_sharded(many, pos, gp, 5, kw, rank=1, world=2, chunk_graphs=3)
    i0 = write_dataset_hdf5_sharded(many, pos, gp, 5, kw, rank=0, world=2, chunk_graphs=3)
    assert i0["shard_range"] == (0, 4) and i1["shard_range"] == (4, 7) and i0["merged_datasets"] == 6 * 7
    assert not os.path.exists(many + ".shard0") and not os.path.exists(many + ".shard1")
    fa, fb = open_h5(one), open_h5(many)
    for op in ("P", "Q"):
        assert sorted(fa[op].keys(), key=int) == sorted(fb[op].keys(), key=int) == [str(i) for i in range(7)]
        for i in range(7):
            for d in ("indices", "values", "size"):
                x, y = fa[op][str(i)][d][...], fb[op][str(i)][d][...]
                assert x.dtype == y.dtype and np.array_equal(x, y), (op, i, d)
    fa.close(); fb.close()


def test_reweight_existing_edges_on_an_unsorted_edge_list_vs_reference_golden():
    """graph_construction='reweight_existing_edges' keeps the reference's quirk -- the COALESCED weights
    (row-major order) are what data.edge_weight receives, and get_P pairs them with the edge list in its original
    order (process_pyg_data.py:758, :563-579).  Golden from the reference on a shuffled edge list
    (tests/golden/make_round3b_goldens.py).  Duplicate edges: the reference fails, so does the drop-in."""
    from esc_gnn_amd.process_pyg_data import process_pyg_data
    _dev()
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "round3b_goldens.npz"))
    pos, ei = g["rw_pos"], g["rw_ei"]
    n = pos.shape[0]
    key = ei[0] * n + ei[1]
    assert not (key[1:] > key[:-1]).all()                                          # the point of the fixture
    kw = {"r_cutoff": 5.0, "gaussian_eps": 4.0, "max_num_neighbors": 32}
    data = _Data(pos=torch.from_numpy(pos), edge_index=torch.from_numpy(ei), num_nodes=n)
    out = process_pyg_data(data, data_i=0, return_data_object=False, graph_construction="reweight_existing_edges",
                           graph_construction_kwargs=kw, local_pca_kernel_fn_kwargs={"kernel": "epanechnikov", "r_cut": 5.0},
                           hdf5_tensor_dtype="float32")
    assert np.array_equal(out["edge_index"].numpy(), g["rw_edge_index"]) and np.array_equal(g["rw_edge_index"], ei)
    assert np.abs(out["edge_weight"].numpy() - g["rw_edge_weight"]).max() <= 2e-7
    assert np.array_equal(out["P"]["indices"].cpu().numpy(), g["rw_P_indices"])
    assert np.abs(out["P"]["values"].cpu().numpy() - g["rw_P_values"]).max() <= 2e-7
    assert np.array_equal(out["Q"]["indices"].cpu().numpy(), g["rw_Q_indices"])
    # the weights aligned with the edges as given (what one might expect) would NOT reproduce the golden
    d = np.linalg.norm(pos[ei[0]] - pos[ei[1]], axis=1)
    w_aligned = np.exp(-(d / 4.0) ** 2) * 0.5 * (np.cos(np.pi * d / 5.0) + 1.0)
    assert np.abs(w_aligned - g["rw_edge_weight"]).max() > 1e-3
    # Q: p_ij times an orthogonal block everywhere; equal to the golden where the frames are well conditioned
    ours, ref = _blocks(out["Q"]["values"].cpu().numpy()), _blocks(g["rw_Q_values"])
    pv = g["rw_P_values"]
    R = ours / pv[:, None, None]
    assert np.abs(np.einsum("eab,ecb->eac", R, R) - np.eye(3)).max() <= 1e-5
    err = np.abs(ours - ref).max(axis=(1, 2))
    assert np.median(err) <= 1e-6 and (err <= 1e-4).mean() >= 0.9, (float(np.median(err)), float((err <= 1e-4).mean()))
    dup = np.concatenate([ei, ei[:, :1]], axis=1)
    with pytest.raises(ValueError):
        process_pyg_data(_Data(pos=torch.from_numpy(pos), edge_index=torch.from_numpy(dup), num_nodes=n), data_i=0,
                         return_data_object=False, graph_construction="reweight_existing_edges", graph_construction_kwargs=kw,
                         local_pca_kernel_fn_kwargs={"kernel": "epanechnikov", "r_cut": 5.0}, hdf5_tensor_dtype="float32")


@pytest.mark.parametrize("tag,method", [("cd", "column_dot"), ("pr", "procrustes")])
def test_vector_features_of_dimension_two_vs_reference_goldens(tag, method):
    """VERDICT r2 "missing 4": process_pyg_data for d != 3 (process_pyg_data.py:330-451 is dimension-generic; the theory
    scripts use planar points, vector_diffusion_equivariance_viz.py:322-395).  Goldens from the reference
    (tests/golden/make_round3b_goldens.py): P bit-exact, Q's 2x2 blocks within the frame-conditioning bound, and the
    dyadic bank of the 2N x 2N operator through get_Batch_Wjxs (scalar CSR route: no BSR-3 form for d = 2)."""
    from esc_gnn_amd.process_pyg_data import process_pyg_data
    from esc_gnn_amd import get_Batch_Wjxs
    dev = _dev()
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "round3b_goldens.npz"))
    pos, ei, eps = g["d2_pos"], g["d2_ei"], float(g["d2_eps"])
    n = pos.shape[0]
    data = _Data(pos=torch.from_numpy(pos), edge_index=torch.from_numpy(ei), num_nodes=n)
    out = process_pyg_data(data, data_i=0, return_data_object=False, graph_construction=None,
                           local_pca_kernel_fn_kwargs={"kernel": "gaussian", "gaussian_eps": eps},
                           sing_vect_align_method=method, hdf5_tensor_dtype="float32")
    assert tuple(out["Q"]["size"]) == (2 * n, 2 * n)
    assert np.array_equal(out["P"]["indices"].cpu().numpy(), g[f"d2_{tag}_P_indices"])
    assert np.array_equal(out["P"]["values"].cpu().numpy(), g[f"d2_{tag}_P_values"])
    assert np.array_equal(out["Q"]["indices"].cpu().numpy(), g[f"d2_{tag}_Q_indices"])
    ours = out["Q"]["values"].cpu().numpy().reshape(-1, 2, 2)
    ref = g[f"d2_{tag}_Q_values"].reshape(-1, 2, 2)
    pv = g[f"d2_{tag}_P_values"]
    R = ours / pv[:, None, None]
    assert np.abs(np.einsum("eab,ecb->eac", R, R) - np.eye(2)).max() <= 1e-5
    # frame conditioning per node: relative gap of the two singular values of its weighted neighbour matrix
    v = torch.from_numpy(pos)
    gaps = np.ones(n)
    for i in range(n):
        nb = ei[1][ei[0] == i]
        V = v[nb] - v[i]
        w = bo.kernel_weights(V.norm(dim=1), "gaussian", gaussian_eps=eps)
        s = torch.linalg.svdvals((w[:, None] * V).T.double())
        gaps[i] = float((s[0] - s[1]) / s[0])
    Pi = g[f"d2_{tag}_P_indices"]
    gg = np.minimum(gaps[Pi[0]], gaps[Pi[1]])
    err = np.abs(ours - ref).max(axis=(1, 2))
    assert (err <= 4e-6 / np.maximum(gg, 1e-6) + 4e-6).all(), float((err * gg).max())
    if tag == "cd":
        obj = process_pyg_data(_Data(pos=torch.from_numpy(pos), edge_index=torch.from_numpy(ei), num_nodes=n),
                               graph_construction=None, local_pca_kernel_fn_kwargs={"kernel": "gaussian", "gaussian_eps": eps})
        assert tuple(obj.Q.shape) == (2 * n, 2 * n)
        W = get_Batch_Wjxs(torch.from_numpy(g["d2_x"]).to(dev), obj.Q, scales_type="dyadic", J=2, include_lowpass=False)
        assert W.shape == g["d2_W"].shape
        # our blocks differ from the reference's by the frame-conditioning error above, which the bank inherits
        assert np.abs(W.cpu().numpy() - g["d2_W"]).max() <= 1e-4 * np.abs(g["d2_W"]).max()
        # and on the reference's OWN operator the chain is at the 1e-5 bar
        Qr = torch.sparse_coo_tensor(torch.from_numpy(g["d2_cd_Q_indices"]), torch.from_numpy(g["d2_cd_Q_values"]), (2 * n, 2 * n)).to(dev)
        Wr = get_Batch_Wjxs(torch.from_numpy(g["d2_x"]).to(dev), Qr, scales_type="dyadic", J=2, include_lowpass=False)
        for f in range(W.shape[-1]):
            assert np.abs(Wr.cpu().numpy()[..., f] - g["d2_W"][..., f]).max() <= 1e-5 * np.abs(g["d2_W"][..., f]).max()
