"""GPU tests of the both-tracks launch (escat_chain_both_f32 / _bwd_f32, esc_gnn_amd/both.py): the scalar track on P and
the vector track on the factored Q of one batch from ONE kernel per direction.  The bar: the forward bit-identical to the
two separate launches on the same values (same arithmetic in the same order), the backward within fp32 rounding of them
(2e-6), and both <= 1e-5 from the C oracle on P and on the BSR-3 blocks."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _built(B, n, k, seed):
    from esc_gnn_amd import synthetic
    from esc_gnn_amd.process_pyg_data import build_PQ_batch
    pos = torch.from_numpy(synthetic.ellipsoid_clouds(B, n, seed=seed).reshape(-1, 3)).to(DEV)
    gp = torch.arange(B + 1, dtype=torch.int32, device=DEV) * n
    return pos, build_PQ_batch(pos, gp, k, {"kernel": "gaussian", "gaussian_eps": 0.05}, factored=True)


KWS = [dict(J=4), dict(J=2, include_lowpass=False),
       dict(scales_type="custom", diffusion_scales=torch.tensor([0, 1, 2, 4, 6, 8, 16])),
       dict(scales_type="custom", diffusion_scales=torch.tensor([1, 3, 5]))]


@pytest.mark.parametrize("B,n,k", [(64, 128, 5), (40, 500, 10), (12, 300, 30), (50, 7, 3), (9, 512, 8)])
def test_both_tracks_bit_identical_to_the_two_launches(B, n, k):
    """(k = 30: rows beyond the pack's capacity -> streamed tails; n = 7: tiny graphs; n = 512: the largest part)"""
    from esc_gnn_amd import get_Batch_Wjxs, get_Batch_Wjxs_both, BothTracksOperator
    pos, b = _built(B, n, k, seed=3 * n + k)
    P, Qf = b["P"], b["Q_factored"]
    bop = BothTracksOperator.pair(P, Qf)
    assert bop is not None and bop.mag_err == 0.0
    g = torch.Generator(device=DEV).manual_seed(5)
    N = B * n
    # bit-identical when the scalar track's own pack has the capacity of the fused one (same slots, same order); k = 30:
    # P alone packs 32 slots per row, the signed operator 24 -- the sums then run in another order: fp32 rounding
    exact = P.csr().sell().dmax == bop.fop.signed.csr().sell().dmax
    assert exact == (k != 30)

    def same(a, b):
        if exact:
            return torch.equal(a, b)
        return float((a - b).abs().max()) <= 2e-6 * float(b.abs().max())
    for kw in KWS:
        xs = torch.randn(N, 3, device=DEV, generator=g)
        xv = torch.randn(3 * N, 1, device=DEV, generator=g)
        xs1, xv1 = xs.clone().requires_grad_(True), xv.clone().requires_grad_(True)
        xs2, xv2 = xs.clone().requires_grad_(True), xv.clone().requires_grad_(True)
        Ws, Wv = get_Batch_Wjxs_both(xs1, P, xv1, Qf, **kw)
        Rs, Rv = get_Batch_Wjxs(xs2, P, **kw), get_Batch_Wjxs(xv2, Qf, **kw)
        assert Ws.shape == Rs.shape and Wv.shape == Rv.shape
        assert same(Ws.detach(), Rs.detach()), float((Ws - Rs).abs().max())
        assert same(Wv.detach(), Rv.detach()), float((Wv - Rv).abs().max())
        Gs = torch.randn(Ws.shape, device=DEV, generator=g)
        Gv = torch.randn(Wv.shape, device=DEV, generator=g)
        ((Ws * Gs).sum() + (Wv * Gv).sum()).backward()
        ((Rs * Gs).sum() + (Rv * Gv).sum()).backward()
        # backward: a step's gradient terms are summed first and then added to the vector (the separate kernels add them one
        # by one): fp32 rounding apart
        for ga, gb in ((xs1.grad, xs2.grad), (xv1.grad, xv2.grad)):
            assert float((ga - gb).abs().max()) <= 2e-6 * float(gb.abs().max()), float((ga - gb).abs().max())


def test_both_tracks_vs_c_oracle():
    """the fused launch against the CPU oracle on P (CSR) and on Q's BSR-3 blocks, forward and gradient, 1e-5 of the
    largest magnitude per output (the chain's bar)"""
    from esc_gnn_amd import get_Batch_Wjxs_both, plan_chain
    from oracle import chain_oracle as co
    B, n, k = 6, 200, 8
    pos, b = _built(B, n, k, seed=77)
    plan = plan_chain(J=4)
    ev = np.asarray(plan.events, dtype=np.int32).reshape(-1, 4)
    g = torch.Generator(device=DEV).manual_seed(6)
    xs = torch.randn(B * n, 3, device=DEV, generator=g, requires_grad=True)
    xv = torch.randn(3 * B * n, 1, device=DEV, generator=g, requires_grad=True)
    Ws, Wv = get_Batch_Wjxs_both(xs, b["P"], xv, b["Q"], J=4)          # Q as blocks: the factored form is recovered
    Gs, Gv = torch.randn(Ws.shape, device=DEV, generator=g), torch.randn(Wv.shape, device=DEV, generator=g)
    ((Ws * Gs).sum() + (Wv * Gv).sum()).backward()
    rp, ci = b["rowptr"].cpu().numpy(), b["col"].cpu().numpy()
    pv, blk = b["pval"].cpu().numpy(), b["blk"].cpu().numpy()
    rs = co.c_chain_csr(rp, ci, pv, xs.detach().cpu().numpy(), ev, plan.T, plan.F)
    rv = co.c_chain_bsr3(rp, ci, blk, xv.detach().cpu().numpy().reshape(-1), ev, plan.T, plan.F)
    gs = co.c_chain_csr_bwd(rp, ci, pv, Gs.cpu().numpy(), ev, plan.T)
    gv = co.c_chain_bsr3_bwd(rp, ci, blk, Gv.cpu().numpy(), ev, plan.T)
    for got, ref in ((Ws, rs), (Wv, rv), (xs.grad, gs), (xv.grad, gv)):
        got = got.detach().cpu().numpy().reshape(ref.shape)
        assert np.abs(got - ref).max() <= 1e-5 * np.abs(ref).max()


def test_both_tracks_falls_back_when_the_pair_does_not_qualify():
    """another channel count on the scalar track, per-channel scales, a Q that is not P's companion: the two separate
    chains, same results as get_Batch_Wjxs"""
    from esc_gnn_amd import get_Batch_Wjxs, get_Batch_Wjxs_both, BothTracksOperator, ScatterOperator
    pos, b = _built(8, 100, 6, seed=9)
    _, b2 = _built(8, 100, 7, seed=10)
    N = 800
    g = torch.Generator(device=DEV).manual_seed(7)
    xv = torch.randn(3 * N, 1, device=DEV, generator=g)
    x5 = torch.randn(N, 5, device=DEV, generator=g)
    Ws, Wv = get_Batch_Wjxs_both(x5, b["P"], xv, b["Q_factored"], J=3)
    assert torch.equal(Ws, get_Batch_Wjxs(x5, b["P"], J=3)) and torch.equal(Wv, get_Batch_Wjxs(xv, b["Q_factored"], J=3))
    assert BothTracksOperator.pair(b["P"], b2["Q_factored"]) is None          # another pattern
    x3 = torch.randn(N, 3, device=DEV, generator=g)
    Ws, Wv = get_Batch_Wjxs_both(x3, b["P"], xv, b2["Q_factored"], J=3)
    assert torch.equal(Ws, get_Batch_Wjxs(x3, b["P"], J=3)) and torch.equal(Wv, get_Batch_Wjxs(xv, b2["Q_factored"], J=3))
    # a P with other values than |Pt|: no pairing
    P2 = ScatterOperator.from_csr(b["rowptr"], b["col"], (b["pval"] * 1.01).contiguous(), N, validate=False)
    assert BothTracksOperator.pair(P2, b["Q_factored"]) is None


def test_both_tracks_c_abi_rejects_what_it_does_not_serve():
    """straight through ctypes: per-channel events and a missing pack are refused with an error code, nothing runs"""
    import ctypes
    from esc_gnn_amd import _lib, BothTracksOperator
    from esc_gnn_amd.both import _both_launch
    from esc_gnn_amd.scales import ChainPlan
    pos, b = _built(4, 64, 5, seed=11)
    bop = BothTracksOperator.pair(b["P"], b["Q_factored"])
    x = torch.randn(256, 3, device=DEV)
    bad = ChainPlan(2, 2, ((0, 0, 1, 0), (0, 0, 1, 1), (0, 0, 1, 2), (2, 1, 1, -1)), True, 3)
    with pytest.raises(_lib.EscatError, match="per-channel"):
        _both_launch(bop, bad, x, x, False)
    lib = _lib.load()
    rc = lib.escat_chain_both_f32(0, 0, 0, 0, 0, 0, 0, 1, 0, 0, 1, 0, 0, 0, 0, 0, 0, 0)
    assert rc != 0 and lib.escat_last_error()


def test_scatter_both_equals_the_two_scatter_calls_and_primes_the_cache():
    """scatter_both = ESCGNNModular.forward's two _scatter calls (orders 0/1/2 per track) on the fused first-order banks;
    with a cache the model's own two calls that follow are hits"""
    from esc_gnn_amd import scatter, scatter_both, ScatteringCache
    B, n = 12, 200
    pos, b = _built(B, n, 8, seed=21)
    g = torch.Generator(device=DEV).manual_seed(8)
    xs = torch.randn(B * n, 3, device=DEV, generator=g)
    xv = torch.randn(B * n, 3, device=DEV, generator=g)
    kw = {"diffusion_kwargs": dict(scales_type="dyadic", J=3)}
    cache = ScatteringCache()
    Ws, Wv = scatter_both(x0_scalar=xs, P=b["P"], x0_vector=xv, Q=b["Q"], kwargs=kw, cache=cache)
    Rs = scatter(track="scalar", x0=xs, P_or_Q=b["P"], kwargs=kw)
    Rv = scatter(track="vector", x0=xv, P_or_Q=b["Q"], kwargs=kw)
    assert Ws.shape == Rs.shape and Wv.shape == Rv.shape == (B * n, 1, 3, Ws.shape[-1])
    assert float((Ws - Rs).abs().max()) <= 2e-6 * float(Rs.abs().max())
    assert float((Wv - Rv).abs().max()) <= 2e-6 * float(Rv.abs().max())
    assert scatter(track="scalar", x0=xs, P_or_Q=b["P"], kwargs=kw, cache=cache) is Ws and cache.hits == 1
    assert scatter(track="vector", x0=xv, P_or_Q=b["Q"], kwargs=kw, cache=cache) is Wv and cache.hits == 2
    # with gradients: one fused launch per direction, same gradients as the two calls
    a1, v1 = xs.clone().requires_grad_(True), xv.clone().requires_grad_(True)
    a2, v2 = xs.clone().requires_grad_(True), xv.clone().requires_grad_(True)
    W1, V1 = scatter_both(x0_scalar=a1, P=b["P"], x0_vector=v1, Q=b["Q"], kwargs=kw)
    W2 = scatter(track="scalar", x0=a2, P_or_Q=b["P"], kwargs=kw)
    V2 = scatter(track="vector", x0=v2, P_or_Q=b["Q"], kwargs=kw)
    Gs, Gv = torch.randn(W1.shape, device=DEV, generator=g), torch.randn(V1.shape, device=DEV, generator=g)
    ((W1 * Gs).sum() + (V1 * Gv).sum()).backward()
    ((W2 * Gs).sum() + (V2 * Gv).sum()).backward()
    assert float((a1.grad - a2.grad).abs().max()) <= 1e-5 * float(a2.grad.abs().max())
    assert float((v1.grad - v2.grad).abs().max()) <= 1e-5 * float(v2.grad.abs().max())


def test_both_tracks_on_views_of_a_stored_operator():
    """mini-batches drawn from stored operators (select: parts in any order, repeats allowed): the stored P and Q pair once,
    the fused launch walks the listed parts in place -- same banks and gradients as the two launches on the same views"""
    from esc_gnn_amd import get_Batch_Wjxs, get_Batch_Wjxs_both
    B, n = 40, 200
    pos, b = _built(B, n, 8, seed=31)
    P, Q, Qf = b["P"], b["Q"], b["Q_factored"]
    n_parts = P.csr().n_parts
    g = torch.Generator(device=DEV).manual_seed(9)
    ids = torch.tensor([3, 0, n_parts - 1, 3, 7], device=DEV)
    for Qsel in (Q, Qf):
        vP, vQ = P.select(ids), Qsel.select(ids)
        N = vP.n
        xs = torch.randn(N, 3, device=DEV, generator=g)
        xv = torch.randn(3 * N, 1, device=DEV, generator=g)
        a1, v1 = xs.clone().requires_grad_(True), xv.clone().requires_grad_(True)
        a2, v2 = xs.clone().requires_grad_(True), xv.clone().requires_grad_(True)
        Ws, Wv = get_Batch_Wjxs_both(a1, vP, v1, vQ, J=3)
        assert type(Ws.grad_fn).__name__ == "_BothBankBackward"          # the fused launch, not the fallback
        Rs, Rv = get_Batch_Wjxs(a2, vP, J=3), get_Batch_Wjxs(v2, vQ, J=3)
        assert Ws.shape == Rs.shape and Wv.shape == Rv.shape
        tol = 0.0 if Qsel is Qf else 2e-6        # Q as blocks: the recovered magnitudes are replaced by P's values (<= 1e-6)
        assert float((Ws - Rs).detach().abs().max()) == 0.0
        assert float((Wv - Rv).detach().abs().max()) <= tol * float(Rv.detach().abs().max())
        Gs, Gv = torch.randn(Ws.shape, device=DEV, generator=g), torch.randn(Wv.shape, device=DEV, generator=g)
        torch.autograd.backward([Ws, Wv], [Gs, Gv])
        torch.autograd.backward([Rs, Rv], [Gs, Gv])
        assert float((a1.grad - a2.grad).abs().max()) <= 2e-6 * float(a2.grad.abs().max())
        assert float((v1.grad - v2.grad).abs().max()) <= 2e-6 * float(v2.grad.abs().max())
    e = torch.zeros(0, dtype=torch.long, device=DEV)
    Ws, Wv = get_Batch_Wjxs_both(torch.zeros(0, 3, device=DEV), P.select(e), torch.zeros(0, 1, device=DEV), Qf.select(e), J=3)
    assert Ws.shape[0] == 0 and Wv.shape[0] == 0
