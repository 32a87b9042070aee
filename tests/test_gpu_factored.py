"""GPU tests of the factored form of the vector operator (FactoredVectorOperator: Q = D Pt D^T, the chain on a CSR
operator with one value per channel, escat_sell.val_planes = 3) against the BSR-3 chain on the blocks the same
builder emits, and of its pieces (the per-channel CSR kernel, the frame rotation).  Tolerance: the chain's 1e-5 per
filter -- the two forms differ by fp32 rounding only (blocks are rounded products of the factors)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _built(B, n, k, seed, factored=True):
    from esc_gnn_amd import synthetic
    from esc_gnn_amd.process_pyg_data import build_PQ_batch
    pos = torch.from_numpy(synthetic.ellipsoid_clouds(B, n, seed=seed).reshape(-1, 3)).to(DEV)
    gp = torch.arange(B + 1, dtype=torch.int32, device=DEV) * n
    return pos, build_PQ_batch(pos, gp, k, {"kernel": "gaussian", "gaussian_eps": 0.05}, factored=factored)


def _rel(a, b):
    a, b = a.detach(), b.detach()
    return float(((a - b).abs().amax(dim=tuple(range(a.dim() - 1))) / b.abs().amax(dim=tuple(range(b.dim() - 1)))).max())


def test_rotation_kernel():
    from esc_gnn_amd.pyg_utilities import rotate_frames
    g = torch.Generator(device=DEV).manual_seed(0)
    O = torch.linalg.qr(torch.randn(1000, 3, 3, device=DEV, generator=g))[0].contiguous()
    for F in (1, 6, 17):
        X = torch.randn(1000, 3, F, device=DEV, generator=g, requires_grad=True)
        Y = rotate_frames(O, X, False)
        assert (Y - torch.einsum("nac,ncf->naf", O, X)).abs().max() < 1e-6
        Z = rotate_frames(O, X, True)
        assert (Z - torch.einsum("nca,ncf->naf", O, X)).abs().max() < 1e-6
        G = torch.randn_like(Y)
        (Y * G).sum().backward()
        assert (X.grad - torch.einsum("nca,ncf->naf", O, G)).abs().max() < 1e-6     # the adjoint is the transpose
        assert (rotate_frames(O, rotate_frames(O, X, True), False) - X).abs().max() < 1e-5   # O O^T = I


@pytest.mark.parametrize("values", ["signs", "general"])
@pytest.mark.parametrize("B,n,k", [(64, 128, 5), (40, 500, 10), (12, 300, 30), (50, 7, 3)])
def test_per_channel_csr_chain(B, n, k, values):
    """y[i][c] = sum_j v[i,j][c] x[j][c]: the per-channel pack == three single-channel chains, forward and backward
    (k = 30: rows longer than the pack's 24 slots -> tails; n = 7: many tiny graphs per workgroup).
    values = signs: the three values differ in sign only -> |value| + sign bits (val_planes = 4);
    general: unrelated values -> three planes (val_planes = 3)."""
    from esc_gnn_amd import get_Batch_Wjxs, ScatterOperator
    pos, b = _built(B, n, k, seed=n + k, factored=False)
    N = B * n
    g = torch.Generator(device=DEV).manual_seed(1)
    val3 = b["pval"][:, None] * torch.where(torch.rand(b["nnz"], 3, device=DEV, generator=g) < 0.4, -1.0, 1.0)
    if values == "general":
        val3 = val3 * (0.5 + torch.rand(b["nnz"], 3, device=DEV, generator=g))
    val3 = val3.contiguous()
    op3 = ScatterOperator.from_csr(b["rowptr"], b["col"], val3, N, validate=False)
    assert (op3.csr().signed_pack() is not None) == (values == "signs")
    assert op3.csr().sell().val_planes == (4 if values == "signs" else 3)
    x = torch.randn(N, 3, device=DEV, generator=g)
    for kw in (dict(J=4), dict(scales_type="custom", diffusion_scales=torch.tensor([0, 1, 2, 4, 6, 8, 16]))):
        xg = x.clone().requires_grad_(True)
        W = get_Batch_Wjxs(xg, op3, **kw)
        G = torch.randn(W.shape, device=DEV, generator=g)
        (W * G).sum().backward()
        for c in range(3):
            opc = ScatterOperator.from_csr(b["rowptr"], b["col"], val3[:, c].contiguous(), N, validate=False)
            xc = x[:, c:c + 1].clone().requires_grad_(True)
            Wc = get_Batch_Wjxs(xc, opc, **kw)
            (Wc * G[:, c:c + 1]).sum().backward()
            assert _rel(W[:, c], Wc[:, 0]) <= 2e-6
            assert float((xg.grad[:, c] - xc.grad[:, 0]).abs().max() / xc.grad.abs().max()) <= 2e-6
    with pytest.raises(NotImplementedError):
        get_Batch_Wjxs(x[:, :2].contiguous(), op3, J=2)            # one value per channel: C must be 3


@pytest.mark.parametrize("rotate,fmt", [("fused", "signed"), ("separate", "signed"), ("fused", "planes")])
@pytest.mark.parametrize("B,n,k,J", [(256, 128, 5, 4), (128, 500, 10, 4), (16, 500, 16, 6), (700, 37, 4, 3), (10, 400, 30, 3)])
def test_factored_chain_vs_bsr3(B, n, k, J, rotate, fmt, monkeypatch):
    """rotate: inside the chain kernel (escat_sell.frames, the default) or as two escat_rotate_frames_f32 launches.
    700 graphs: more parts than resident workgroups, so the pipelined prologue / epilogue of both directions runs."""
    from esc_gnn_amd import get_Batch_Wjxs
    monkeypatch.setenv("ESCAT_FACTORED_ROTATE", rotate)
    monkeypatch.setenv("ESCAT_FACTORED_FORMAT", fmt)            # |value| + sign bits (default) / three value planes
    pos, b = _built(B, n, k, seed=3 * n + k)
    Qf, Q = b["Q_factored"], b["Q"]
    # the factors reproduce the builder's blocks
    rp = b["rowptr"].long()
    rows = torch.repeat_interleave(torch.arange(B * n, device=DEV), rp[1:] - rp[:-1])
    O, v3 = Qf.frames, Qf.signed.csr().val
    blk = torch.einsum("eac,ec,ebc->eab", O[rows], v3, O[b["col"].long()])
    assert (blk - b["blk"]).abs().max() < 2e-7
    g = torch.Generator(device=DEV).manual_seed(2)
    x = torch.randn(3 * B * n, 1, device=DEV, generator=g)
    for kw in (dict(J=J), dict(scales_type="custom", diffusion_scales=torch.tensor([0, 1, 2, 4, 6, 8, 16])), dict(J=J, include_lowpass=False)):
        xa, xb = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
        Wf, Wq = get_Batch_Wjxs(xa, Qf, **kw), get_Batch_Wjxs(xb, Q, **kw)
        assert Wf.shape == Wq.shape and _rel(Wf, Wq) <= 1e-5
        G = torch.randn(Wq.shape, device=DEV, generator=g)
        (Wf * G).sum().backward(); (Wq * G).sum().backward()
        assert float((xa.grad - xb.grad).abs().max() / xb.grad.abs().max()) <= 1e-5


def test_scatter_with_factored_operator():
    from esc_gnn_amd import scatter
    pos, b = _built(200, 128, 5, seed=9)
    kw = {"diffusion_kwargs": dict(scales_type="dyadic", J=4, include_lowpass=True)}
    for abl in (False, True):
        Sf = scatter(track="vector", x0=pos, P_or_Q=b["Q_factored"], kwargs=kw, ablate_second_order_wavelets=abl)
        Sq = scatter(track="vector", x0=pos, P_or_Q=b["Q"], kwargs=kw, ablate_second_order_wavelets=abl)
        assert Sf.shape == Sq.shape == (200 * 128, 1, 3, 7 if abl else 22)
        assert _rel(Sf, Sq) <= 1e-5
    xg = pos.clone().requires_grad_(True)
    S = scatter(track="vector", x0=xg, P_or_Q=b["Q_factored"], kwargs=kw)
    S.square().sum().backward()
    xr = pos.clone().requires_grad_(True)
    scatter(track="vector", x0=xr, P_or_Q=b["Q"], kwargs=kw).square().sum().backward()
    assert float((xg.grad - xr.grad).abs().max() / xr.grad.abs().max()) <= 1e-5


@pytest.mark.parametrize("fmt", ["signed", "planes"])
def test_factored_large_graph_multi_launch_and_column_dot(monkeypatch, fmt):
    """graphs beyond one workgroup (1000 nodes; and a relabelled 20 000-node cloud): the multi-launch step kernels
    with one value per channel, rotations as separate launches, forward and backward vs the BSR-3 chain."""
    from esc_gnn_amd import get_Batch_Wjxs, synthetic, ScatterOperator, FactoredVectorOperator
    from esc_gnn_amd.process_pyg_data import build_PQ_batch
    monkeypatch.setenv("ESCAT_FACTORED_FORMAT", fmt)
    pos = torch.from_numpy(synthetic.ellipsoid_clouds(1, 1000, seed=1).reshape(-1, 3)).to(DEV)
    gp = torch.tensor([0, 1000], dtype=torch.int32, device=DEV)
    b = build_PQ_batch(pos, gp, 8, {"kernel": "gaussian", "gaussian_eps": 0.05}, factored=True)
    assert not b["Q_factored"].fits_register_path(4)
    g = torch.Generator(device=DEV).manual_seed(4)
    xa = torch.randn(3000, 1, device=DEV, generator=g).requires_grad_(True)
    xb = xa.detach().clone().requires_grad_(True)
    Wf, Wq = get_Batch_Wjxs(xa, b["Q_factored"], J=3), get_Batch_Wjxs(xb, b["Q"], J=3)
    assert _rel(Wf, Wq) <= 1e-5
    G = torch.randn(Wq.shape, device=DEV, generator=g)
    (Wf * G).sum().backward(); (Wq * G).sum().backward()
    assert float((xa.grad - xb.grad).abs().max() / xb.grad.abs().max()) <= 1e-5
    # one big cloud, rows relabelled for locality (ESCAT_REORDER=1 forces it at this size)
    monkeypatch.setenv("ESCAT_REORDER", "1")
    c = synthetic.single_cloud(20000, 12, workers=1)
    t = lambda a: torch.from_numpy(a).to(DEV)
    none = (torch.zeros(1, dtype=torch.int32, device=DEV), 0, 0)
    opQ = ScatterOperator.from_bsr3(t(c["rowptr"]), t(c["col"]), t(c["blk"]), c["N"], parts=none)
    fop = FactoredVectorOperator.from_frames(t(c["rowptr"]), t(c["col"]), t(c["pval"]), t(c["frames"]), parts=none)
    assert fop.signed.row_perm is not None
    xv = t(c["pos"]).reshape(-1, 1)
    assert _rel(get_Batch_Wjxs(xv, fop, J=4), get_Batch_Wjxs(xv, opQ, J=4)) <= 1e-5
    with pytest.raises(ValueError):
        build_PQ_batch(pos, gp, 8, {"kernel": "gaussian", "gaussian_eps": 0.05}, sing_vect_align_method="procrustes", factored=True)


def test_store_with_frames_and_factored_views(tmp_path):
    """dataset -> HDF5 (reference layout + the /frames extension, fp32 values) -> operator store -> shuffled
    mini-batches as views: the factored views give what the BSR-3 views give; the file still reads as the
    reference's layout (P, Q groups untouched)."""
    from esc_gnn_amd import get_Batch_Wjxs, synthetic
    from esc_gnn_amd.process_pyg_data import write_dataset_hdf5
    from esc_gnn_amd.store import load_operator_store
    from esc_gnn_amd.hdf5_min import H5Reader
    sizes = [32, 64, 48, 128, 20, 300]
    clouds = [synthetic.ellipsoid_clouds(1, n, seed=200 + n)[0] for n in sizes]
    pos = torch.from_numpy(np.concatenate(clouds)).to(DEV)
    gp = torch.from_numpy(np.cumsum([0] + sizes).astype(np.int32))
    path = str(tmp_path / "pqf.h5")
    write_dataset_hdf5(path, pos, gp, 5, {"kernel": "gaussian", "gaussian_eps": 0.1}, hdf5_tensor_dtype="float32",
                       chunk_graphs=4, store_frames=True)
    f = H5Reader(path)
    assert set(f.keys()) == {"P", "Q", "frames"} and f["frames"]["3"]["values"][...].shape == (128, 3, 3)
    f.close()
    store = load_operator_store(path, torch.device(DEV), factored=True)
    Qf, Q = store["Q_factored"], store["Q"]
    assert Qf.signed.csr().n_parts == len(sizes) and Qf.signed.csr().sell().val_planes == 4
    xv_all = pos.reshape(-1, 1)
    W_full = get_Batch_Wjxs(xv_all, Q, J=3)
    assert _rel(get_Batch_Wjxs(xv_all, Qf, J=3), W_full) <= 1e-5
    order = [5, 0, 3, 3, 1]
    vf, vq = Qf.select(order), Q.select(order)
    assert torch.equal(vf.row_index(), vq.row_index())
    x = xv_all[vq.row_index()].clone().requires_grad_(True)
    xf = x.detach().clone().requires_grad_(True)
    Wq, Wf = get_Batch_Wjxs(x, vq, J=3), get_Batch_Wjxs(xf, vf, J=3)
    assert _rel(Wf, Wq) <= 1e-5
    G = torch.randn(Wq.shape, device=DEV, generator=torch.Generator(device=DEV).manual_seed(0))
    (Wq * G).sum().backward(); (Wf * G).sum().backward()
    assert float((xf.grad - x.grad).abs().max() / x.grad.abs().max()) <= 1e-5
    # gradient-free _scatter on the views, orders from the chain's epilogue (ESCAT_ORDERS=epilogue; with frames for the factored view)
    from esc_gnn_amd import scatter
    from esc_gnn_amd.scatter import ScatteringCache
    kw = {"diffusion_kwargs": dict(scales_type="dyadic", J=3, include_lowpass=True)}
    xn = pos[vq.row_index()[::3] // 3].contiguous()                  # (nodes of the view, 3)
    cache = ScatteringCache()
    with torch.no_grad(), pytest.MonkeyPatch.context() as mp:
        mp.setenv("ESCAT_ORDERS", "epilogue")
        Sq = scatter(track="vector", x0=xn, P_or_Q=vq, kwargs=kw)
        Sf = scatter(track="vector", x0=xn, P_or_Q=vf, kwargs=kw, cache=cache)
        assert scatter(track="vector", x0=xn, P_or_Q=vf, kwargs=kw, cache=cache) is Sf and cache.hits == 1
    assert Sf.shape == Sq.shape and _rel(Sf, Sq) <= 1e-5
    with torch.no_grad():                                            # the default: bank + orders kernels
        Dq, Df = scatter(track="vector", x0=xn, P_or_Q=vq, kwargs=kw), scatter(track="vector", x0=xn, P_or_Q=vf, kwargs=kw)
    assert _rel(Dq, Sq) <= 1e-6 and _rel(Df, Sf) <= 1e-6
    with pytest.raises(KeyError):
        path2 = str(tmp_path / "plain.h5")
        write_dataset_hdf5(path2, pos, gp, 5, {"kernel": "gaussian", "gaussian_eps": 0.1}, chunk_graphs=4)
        load_operator_store(path2, torch.device(DEV), factored=True)


def test_builder_frames_hint_serves_clouds_of_any_size(monkeypatch):
    """The builder knows the frames its blocks were made from (ScatterOperator.hint_frames): Q.factored() then reads the
    signs off the blocks and checks every block instead of running the recovery -- which also covers a cloud larger
    than a workgroup's part (BASELINE cfg3's shape: multi-launch path, with and without the locality relabelling), where
    the recovery kernel does not apply.  The chain on such a Q takes the factored form by itself and stays within 1e-5
    of the C oracle on the BLOCKS; frames that do not belong to the blocks are refused."""
    from esc_gnn_amd import get_Batch_Wjxs, get_Batch_Wjxs_both, BothTracksOperator, ScatterOperator, FactoredVectorOperator, synthetic, plan_chain
    from esc_gnn_amd import operators
    from esc_gnn_amd.pyg_utilities import vector_route
    from esc_gnn_amd.process_pyg_data import build_PQ_batch
    from oracle import chain_oracle as co
    monkeypatch.setattr(operators, "LARGE_RECOVERY", False)     # (the recovery of a relabelled cloud's frames has its own test: test_gpu_recover.py)
    kern = {"kernel": "gaussian", "gaussian_eps": 0.05}
    plan = plan_chain(J=4)
    ev = np.asarray(plan.events, dtype=np.int32).reshape(-1, 4)
    g = torch.Generator(device=DEV).manual_seed(6)
    for n, reorder in ((3000, "auto"), (3000, "1")):
        monkeypatch.setenv("ESCAT_REORDER", reorder)
        pos = torch.from_numpy(synthetic.ellipsoid_clouds(1, n, seed=n).reshape(-1, 3)).to(DEV)
        b = build_PQ_batch(pos, torch.tensor([0, n], dtype=torch.int32, device=DEV), 8, kern)
        Q = b["Q"]
        assert (Q.row_perm is not None) == (reorder == "1")
        f = Q.factored()
        assert isinstance(f, FactoredVectorOperator) and Q.factor_err <= 1e-6 and not f.fits_register_path(plan.F)
        x = torch.randn(3 * n, 1, device=DEV, generator=g)
        assert vector_route(Q, x, plan.F, plan.events) is f
        xa = x.clone().requires_grad_(True)
        W = get_Batch_Wjxs(xa, Q, J=4)
        G = torch.randn(W.shape, device=DEV, generator=g)
        W.backward(G)
        rp, col, blk = b["rowptr"].cpu().numpy(), b["col"].cpu().numpy(), b["blk"].cpu().numpy()
        ref = co.c_chain_bsr3(rp, col, blk, x.cpu().numpy().reshape(-1), ev, plan.T, plan.F)
        refg = co.c_chain_bsr3_bwd(rp, col, blk, G.cpu().numpy(), ev, plan.T)
        assert np.abs(W.detach().cpu().numpy() - ref).max() <= 1e-5 * np.abs(ref).max()
        assert np.abs(xa.grad.cpu().numpy() - refg).max() <= 1e-5 * np.abs(refg).max()
        # the same blocks without the hint (and without the whole-device recovery): no factored form at this size
        Q0 = ScatterOperator.from_bsr3(b["rowptr"], b["col"], b["blk"], n, validate=False)
        assert Q0.factored() is None and vector_route(Q0, x, plan.F, plan.events) is Q0
        assert _rel(get_Batch_Wjxs(x, Q0, J=4), W) <= 1e-5
        # frames that are not the blocks': refused, the chain stays on the blocks
        O = torch.linalg.qr(torch.randn(n, 3, 3, device=DEV, generator=g))[0].contiguous()
        Q0.hint_frames(O)
        assert Q0.factored() is None and Q0.factor_err > 1e-3
        with pytest.raises(ValueError):
            Q0.hint_frames(O[:-1])
    monkeypatch.delenv("ESCAT_REORDER")
    # a batch of small graphs: the hinted form is what the recovery would find, P pairs with it, both tracks in one launch
    pos, b = _built(64, 128, 5, seed=11, factored=False)
    f = b["Q"].factored()
    assert f is not None and b["Q"].factor_err <= 1e-6 and f.fits_register_path(plan.F)
    assert torch.equal(f.signed.csr().val.abs(), b["pval"][:, None].expand(-1, 3))          # P's own magnitudes
    assert BothTracksOperator.pair(b["P"], b["Q"]) is not None
    xs, xv = torch.randn(64 * 128, 3, device=DEV, generator=g), torch.randn(3 * 64 * 128, 1, device=DEV, generator=g)
    Ws, Wv = get_Batch_Wjxs_both(xs, b["P"], xv, b["Q"], J=4)
    rp, col = b["rowptr"].cpu().numpy(), b["col"].cpu().numpy()
    refs = co.c_chain_csr(rp, col, b["pval"].cpu().numpy(), xs.cpu().numpy(), ev, plan.T, plan.F)
    refv = co.c_chain_bsr3(rp, col, b["blk"].cpu().numpy(), xv.cpu().numpy().reshape(-1), ev, plan.T, plan.F)
    assert np.abs(Ws.cpu().numpy() - refs).max() <= 1e-5 * np.abs(refs).max()
    assert np.abs(Wv.cpu().numpy().reshape(refv.shape) - refv).max() <= 1e-5 * np.abs(refv).max()
    # 'procrustes' blocks carry no hint (and have no factored form)
    bp = build_PQ_batch(pos[:128], torch.tensor([0, 128], dtype=torch.int32, device=DEV), 5, kern, sing_vect_align_method="procrustes")
    assert bp["Q"]._frames_hint is None
