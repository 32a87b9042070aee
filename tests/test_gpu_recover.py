"""
Round-3 GPU tests: the factored form of the vector operator recovered from the BSR-3 blocks alone
(include/escat.h: escat_recover_factored, csrc/escat_factor.hip) and its automatic use by the drop-in.

The reference hands get_Batch_Wjxs the COO of Q's blocks (models/escgnn_modular.py:345-351); its builder makes every block
p_ij O_i S_ij O_j^T (data_processing/process_pyg_data.py:306-313, :492) but stores no frames.  For such an operator the
ingest must find frames and signs by itself, verify them against the blocks, and run the factored kernel; for any other
operator (fp16-rounded blocks, 'procrustes', random blocks) it must say no and keep BSR-3.
"""
import numpy as np
import pytest
import torch

from oracle import chain_oracle as co

pytestmark = pytest.mark.gpu


def _dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


def _close(a, ref, rtol=1e-5):
    a = a.detach().cpu().numpy() if torch.is_tensor(a) else a
    assert a.shape == ref.shape, (a.shape, ref.shape)
    for f in range(ref.shape[-1]):
        d = np.abs(a[..., f] - ref[..., f]).max()
        assert d <= rtol * np.abs(ref[..., f]).max() + 1e-30, (f, d, np.abs(ref[..., f]).max())


def _coo(idx, val, n, dev):
    return torch.sparse_coo_tensor(torch.from_numpy(idx.astype(np.int64)).to(dev), torch.from_numpy(val.astype(np.float32)).to(dev), (n, n))


def _check_factors(op, f):
    """frames orthonormal, the three channels differ in sign only, and p G S G^T gives the blocks back"""
    b = op.bsr3()
    G = f.frames.double()
    eye = torch.eye(3, dtype=torch.float64, device=G.device)
    assert float((G.transpose(1, 2) @ G - eye).abs().max()) < 1e-6
    v3 = f.signed.csr().val
    assert torch.equal(v3.abs(), v3[:, :1].abs().expand(-1, 3))
    rp = b.rowptr.long()
    rows = torch.repeat_interleave(torch.arange(b.n, device=G.device), rp[1:] - rp[:-1])
    rec = torch.einsum("eac,ec,ebc->eab", G[rows], v3.double(), G[b.col.long()])
    err = float((rec - b.val.double()).abs().max())
    assert err <= 1.2e-6 * float(b.val.abs().max()), err


def test_recovery_on_the_reference_builders_own_blocks(sg, bg, r3g):
    """Q of the builder goldens g0-g4 (column_dot, with and without sign enforcement) and of their batch, fp32 as
    process_pyg_data returns them: a factorisation is found and reproduces every block to 1e-6.  The round-1 scatter
    goldens' batch contains the 'procrustes' graph g5: one graph without the structure refuses the whole operator."""
    from esc_gnn_amd import ScatterOperator
    dev = _dev()
    n3 = int(sg["N"]) * 3
    op = ScatterOperator.from_torch(_coo(sg["Q_indices"], sg["Q_values"], n3, dev))
    assert op.factored() is None and op.factor_err > 1e-3
    cases = [("batch", r3g["Q_indices"], r3g["Q_values_f32"], int(r3g["N3"]))]
    for g in range(5):
        cases.append((f"g{g}", bg[f"g{g}_Q_indices"], bg[f"g{g}_Q_values_f32"], int(bg[f"g{g}_Q_size"][0])))
    for name, idx, val, n in cases:
        op = ScatterOperator.from_torch(_coo(idx, val, n, dev))
        f = op.factored()
        assert f is not None, (name, op.factor_err)
        assert op.factor_err <= 1e-6, (name, op.factor_err)
        _check_factors(op, f)


def test_drop_in_takes_the_factored_kernel_by_itself_and_matches_the_reference(r3g):
    """get_Batch_Wjxs(x, Q_coo) and scatter(track='vector') on goldens generated from the reference (a batch of column_dot
    graphs, fp32 blocks): the route is the factored operator found at ingest, and the bank, x.grad and the orders 0/1/2
    stay within 1e-5 of what the reference's torch.sparse chain gave."""
    from esc_gnn_amd import get_Batch_Wjxs, as_operator, scatter
    from esc_gnn_amd.operators import FactoredVectorOperator
    from esc_gnn_amd.pyg_utilities import vector_route
    dev = _dev()
    n3 = int(r3g["N3"])
    Q = _coo(r3g["Q_indices"], r3g["Q_values_f32"], n3, dev)
    xv = torch.from_numpy(r3g["x_v"]).to(dev)
    # default ingest policy ("auto"): the first time a tensor object is seen it gets the single-use treatment (BSR-3
    # kernel on an unplanned pack: a DataLoader's batch comes once); when the same tensor comes back it is promoted and the
    # factored form is found.  Both routes must meet the goldens.
    x1 = xv.reshape(-1, 1).clone().requires_grad_(True)
    W1 = get_Batch_Wjxs(x1, Q, scales_type="dyadic", J=4)
    op1 = as_operator.__globals__["ScatterOperator"] and getattr(Q, "_escat_operator")
    assert op1.policy == "single_use" and op1._factored is False and not op1.bsr3().sell().planned
    _close(W1, r3g["f32_dyadic_J4_W"])
    W1.backward(torch.from_numpy(r3g["f32_dyadic_J4_gup"]).to(dev))
    _close(x1.grad.reshape(-1, 1), r3g["f32_dyadic_J4_gx"].reshape(-1, 1))
    assert isinstance(vector_route(as_operator(Q), xv.reshape(-1, 1), 6), FactoredVectorOperator)     # second use: promoted
    assert as_operator(Q) is op1 and op1.policy == "full"
    modes = {"dyadic_J4": dict(scales_type="dyadic", J=4),
             "custom_0_16": dict(scales_type="custom", diffusion_scales=torch.tensor([0, 1, 2, 4, 6, 8, 16])),
             "custom_1_2_4": dict(scales_type="custom", diffusion_scales=torch.tensor([1, 2, 4])),
             "dyadic_J2_nolp": dict(scales_type="dyadic", J=2, include_lowpass=False)}
    for name, kw in modes.items():
        x = xv.reshape(-1, 1).clone().requires_grad_(True)
        W = get_Batch_Wjxs(x, Q, **kw)
        _close(W, r3g[f"f32_{name}_W"])
        W.backward(torch.from_numpy(r3g[f"f32_{name}_gup"]).to(dev))
        _close(x.grad.reshape(-1, 1), r3g[f"f32_{name}_gx"].reshape(-1, 1))
    kw = {"diffusion_kwargs": dict(scales_type="dyadic", J=4, include_lowpass=True, diffusion_scales="dyadic")}
    ref = r3g["f32_scatter_vector"]
    for grad, mode in ((False, "epilogue"), (False, "kernel"), (True, "kernel")):   # orders from the chain's epilogue / the orders kernel
        x0 = xv.clone().requires_grad_(grad)
        with pytest.MonkeyPatch.context() as mp:
            mp.setenv("ESCAT_ORDERS", mode)
            S = scatter(track="vector", x0=x0, P_or_Q=Q, kwargs=kw)
        assert S.shape == ref.shape
        _close(S.reshape(-1, S.shape[-1]), ref.reshape(-1, ref.shape[-1]), rtol=2e-5)   # products of two bands: 2 x 1e-5


def test_operators_without_the_structure_stay_bsr3(bg, monkeypatch):
    """the 'procrustes' golden (g5: O_ij = (O_i O_j^T)^2), fp16-rounded blocks (even when the fp16 probe is bypassed) and
    blocks with a random rotation per edge are refused; the chain then runs the BSR-3 kernels as before."""
    from esc_gnn_amd import ScatterOperator, get_Batch_Wjxs, operators
    from esc_gnn_amd.pyg_utilities import vector_route
    dev = _dev()
    n = int(bg["g5_Q_size"][0])
    op = ScatterOperator.from_torch(_coo(bg["g5_Q_indices"], bg["g5_Q_values_f32"], n, dev))
    assert op.factored() is None and op.factor_err > 1e-3
    x = torch.randn(n, 1, device=dev)
    assert vector_route(op, x, 6) is op
    rowptr, col, blk = co.coo_to_bsr3(bg["g5_Q_indices"].astype(np.int64), bg["g5_Q_values_f32"].astype(np.float32), n)
    ev = np.asarray(operators_plan_events(4), dtype=np.int32).reshape(-1, 4)
    _close(get_Batch_Wjxs(x, op, J=4), co.c_chain_bsr3(rowptr, col, blk, x.cpu().numpy().reshape(-1), ev, 16, 6).reshape(n, 1, 6))
    # fp16-rounded blocks: the probe routes them to the half pack; forced through the recovery they fail its check
    n2 = int(bg["g2_Q_size"][0])
    A16 = _coo(bg["g2_Q_indices"], bg["g2_Q_values_f16"].astype(np.float32), n2, dev)
    op16 = ScatterOperator.from_torch(A16)
    assert op16.values_look_fp16() and op16.factored() is None and op16.factor_err is None
    monkeypatch.setattr(operators.ScatterOperator, "values_look_fp16", lambda self: False)
    op16b = ScatterOperator.from_torch(_coo(bg["g2_Q_indices"], bg["g2_Q_values_f16"].astype(np.float32), n2, dev))
    assert op16b.factored() is None and op16b.factor_err > 1e-4
    # a random rotation per edge on P's pattern (what SURVEY 8d allows for pure throughput runs): not factorable
    from esc_gnn_amd import synthetic
    b = synthetic.ellipsoid_batch(4, 100, 5, seed=5)
    R = synthetic.random_frames(b["nnz"], 99)
    blk_r = (b["pval"][:, None, None] * R).astype(np.float32)
    t = lambda a: torch.from_numpy(a).to(dev)
    opr = ScatterOperator.from_bsr3(t(b["rowptr"]), t(b["col"]), t(blk_r), b["N"])
    assert opr.factored() is None and opr.factor_err > 1e-2


def operators_plan_events(J):
    from esc_gnn_amd import plan_chain
    return plan_chain(scales_type="dyadic", J=J).events


@pytest.mark.parametrize("B,n,k", [(40, 500, 10), (64, 200, 5), (300, 7, 3), (6, 300, 12)])
def test_recovered_form_against_the_bsr3_chain_and_the_oracle(oracle_lib, monkeypatch, B, n, k):
    """builder-shaped blocks on synthetic batches (random frames, column-dot signs): 512- and 256-thread parts, parts made
    of several tiny graphs (several connected components per workgroup): factored route = BSR-3 route = C oracle to 1e-5,
    forward, backward and fused orders."""
    from esc_gnn_amd import get_Batch_Wjxs, ScatterOperator, synthetic, plan_chain, scatter, operators
    from esc_gnn_amd.operators import FactoredVectorOperator
    from esc_gnn_amd.pyg_utilities import vector_route
    dev = _dev()
    b = synthetic.ellipsoid_batch(B, n, k, seed=2000 + n)
    t = lambda a: torch.from_numpy(a).to(dev)
    mk = lambda: ScatterOperator.from_bsr3(t(b["rowptr"]), t(b["col"]), t(b["blk"]), b["N"])
    op_f = mk()
    f = op_f.factored()
    assert f is not None and op_f.factor_err <= 1e-6, op_f.factor_err
    _check_factors(op_f, f)
    monkeypatch.setattr(operators, "FACTOR_MODE", "0")
    op_b = mk()
    assert op_b.factored() is None
    monkeypatch.setattr(operators, "FACTOR_MODE", "auto")
    xv = torch.from_numpy(b["pos"]).reshape(-1, 1)
    assert isinstance(vector_route(op_f, xv.to(dev), 6), FactoredVectorOperator)
    kw = dict(scales_type="dyadic", J=4)
    plan = plan_chain(**kw)
    ev = np.asarray(plan.events, dtype=np.int32).reshape(-1, 4)
    xf, xb = xv.to(dev).requires_grad_(True), xv.to(dev).requires_grad_(True)
    Wf, Wb = get_Batch_Wjxs(xf, op_f, **kw), get_Batch_Wjxs(xb, op_b, **kw)
    ref = co.c_chain_bsr3(b["rowptr"], b["col"], b["blk"], xv.numpy().reshape(-1), ev, plan.T, plan.F).reshape(Wf.shape)
    _close(Wf, ref)
    _close(Wb, ref)
    G = torch.randn(Wf.shape, device=dev, generator=torch.Generator(device=dev).manual_seed(2))
    (Wf * G).sum().backward()
    (Wb * G).sum().backward()
    gref = co.c_chain_bsr3_bwd(b["rowptr"], b["col"], b["blk"], G.cpu().numpy().reshape(b["N"] * 3, -1), ev, plan.T)
    _close(xf.grad.reshape(-1, 1), gref.reshape(-1, 1))
    _close(xb.grad.reshape(-1, 1), gref.reshape(-1, 1))
    pos = torch.from_numpy(b["pos"]).to(dev)
    with torch.no_grad():
        Sf = scatter(track="vector", x0=pos, P_or_Q=op_f, kwargs=dict(diffusion_kwargs=kw))
        Sb = scatter(track="vector", x0=pos, P_or_Q=op_b, kwargs=dict(diffusion_kwargs=kw))
    _close(Sf.reshape(-1, Sf.shape[-1]), Sb.cpu().numpy().reshape(-1, Sb.shape[-1]), rtol=2e-5)   # products of two bands: 2 x 1e-5


def test_frames_of_one_large_cloud_recovered_from_the_blocks(monkeypatch):
    """BASELINE cfg3's shape: ONE connected cloud, far beyond a workgroup's part, blocks only (what the reference hands
    over).  esc_gnn_amd.recover_large finds the frames across the whole device (breadth-first tree by pointer doubling,
    gauge from the holonomies, level walk + one sweep); every block is then checked, and the chain takes the factored
    multi-launch path by itself -- within 1e-5 of the C oracle on the BLOCKS, forward and gradient.  Random frames (every
    second sign negative) and the builder's local-PCA frames (two columns flip together almost everywhere: the case that
    needs the joint-diagonalisation steps).  fp16-rounded and 'procrustes' blocks are refused and stay BSR-3."""
    from esc_gnn_amd import get_Batch_Wjxs, ScatterOperator, FactoredVectorOperator, synthetic, plan_chain, operators
    from esc_gnn_amd.pyg_utilities import vector_route
    from esc_gnn_amd.process_pyg_data import build_PQ_batch
    dev = _dev()
    monkeypatch.setenv("ESCAT_REORDER", "1")                    # relabel at this size too (the default does it from 16 384 rows on)
    n, k = 30000, 10
    c = synthetic.single_cloud(n, k, workers=1)
    t = lambda a: torch.from_numpy(a).to(dev)
    none = (torch.zeros(1, dtype=torch.int32, device=dev), 0, 0)
    plan = plan_chain(J=4)
    ev = np.asarray(plan.events, dtype=np.int32).reshape(-1, 4)
    kern = {"kernel": "gaussian", "gaussian_eps": 0.05}
    bb = build_PQ_batch(t(c["pos"]), torch.tensor([0, n], dtype=torch.int32), k, kern)
    cases = {"random_frames": (t(c["rowptr"]), t(c["col"]), t(c["blk"])), "local_pca_frames": (bb["rowptr"], bb["col"], bb["blk"])}
    g = torch.Generator(device=dev).manual_seed(3)
    for name, (rp, col, blk) in cases.items():
        Q = ScatterOperator.from_bsr3(rp, col, blk, n, parts=none)
        assert Q.row_perm is not None and Q._frames_hint is None
        f = Q.factored()
        assert isinstance(f, FactoredVectorOperator), (name, Q.factor_err)
        assert Q.factor_err <= 1e-6 and not f.fits_register_path(plan.F)
        x = torch.randn(3 * n, 1, device=dev, generator=g)
        assert vector_route(Q, x, plan.F, plan.events) is f
        xa = x.clone().requires_grad_(True)
        W = get_Batch_Wjxs(xa, Q, J=4)
        G = torch.randn(W.shape, device=dev, generator=g)
        W.backward(G)
        a = [z.cpu().numpy() for z in (rp, col, blk)]
        ref = co.c_chain_bsr3(a[0], a[1], a[2], x.cpu().numpy().reshape(-1), ev, plan.T, plan.F)
        refg = co.c_chain_bsr3_bwd(a[0], a[1], a[2], G.cpu().numpy(), ev, plan.T)
        assert np.abs(W.detach().cpu().numpy() - ref).max() <= 1e-5 * np.abs(ref).max(), name
        assert np.abs(xa.grad.cpu().numpy() - refg).max() <= 1e-5 * np.abs(refg).max(), name
    # two clouds in one operator (two components, relabelled one after the other: the levels are no longer row ranges)
    rp, col, blk = cases["random_frames"]
    rp2 = torch.cat([rp, rp[1:] + rp[-1]]).to(torch.int32)
    col2 = torch.cat([col, col + n]).to(torch.int32)
    blk2 = torch.cat([blk, blk.flip(0)[torch.argsort(torch.arange(blk.shape[0], device=dev).flip(0))]])     # (the same blocks twice)
    Q2 = ScatterOperator.from_bsr3(rp2, col2, blk2, 2 * n, parts=none)
    f2 = Q2.factored()
    assert Q2.row_perm is not None and isinstance(f2, FactoredVectorOperator) and Q2.factor_err <= 1e-6
    x2 = torch.randn(6 * n, 1, device=dev, generator=g)
    W2 = get_Batch_Wjxs(x2, Q2, J=4)
    W1 = get_Batch_Wjxs(x2[: 3 * n].contiguous(), ScatterOperator.from_bsr3(rp, col, blk, n, parts=none), J=4)
    assert float((W2[: 3 * n] - W1).abs().max() / W1.abs().max()) <= 1e-5
    # refused: values rounded to fp16 (probe bypassed: the check of every block has to catch it), procrustes alignment
    Qh = ScatterOperator.from_bsr3(rp, col, blk.half().float(), n, parts=none)
    assert Qh.factored() is None                                                    # (the fp16 probe)
    Qh2 = ScatterOperator.from_bsr3(rp, col, blk.half().float(), n, parts=none)
    Qh2._fp16_probe = False
    assert Qh2.factored() is None and Qh2.factor_err > 1e-5
    bp = build_PQ_batch(t(c["pos"]), torch.tensor([0, n], dtype=torch.int32), k, kern, sing_vect_align_method="procrustes")
    Qp = ScatterOperator.from_bsr3(bp["rowptr"], bp["col"], bp["blk"], n, parts=none)
    assert Qp.factored() is None
    x = torch.randn(3 * n, 1, device=dev, generator=g)
    assert vector_route(Qp, x, plan.F, plan.events) is Qp
    # switched off: the blocks stay blocks
    monkeypatch.setattr(operators, "LARGE_RECOVERY", False)
    Q0 = ScatterOperator.from_bsr3(rp, col, blk, n, parts=none)
    assert Q0.factored() is None
