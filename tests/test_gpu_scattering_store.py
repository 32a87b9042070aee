"""GPU tests of esc_gnn_amd.store.ScatteringStore: the scattering coefficients of a whole dataset computed once (both
tracks, orders 0/1/2), mini-batches as gathers -- against _scatter on the batch itself and against the reference's formula
on the C oracle's banks."""
import numpy as np
import pytest
import torch

from oracle import chain_oracle as co

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _dataset(sizes, k, seed, fp16_values=False):
    from esc_gnn_amd import synthetic, ScatterOperator
    from esc_gnn_amd.process_pyg_data import build_PQ_batch
    rng = np.random.RandomState(seed)
    clouds = [synthetic.ellipsoid_clouds(1, n, seed=int(rng.randint(1 << 30)))[0] for n in sizes]
    pos = torch.from_numpy(np.concatenate(clouds)).to(DEV)
    gp = torch.from_numpy(np.cumsum([0] + list(sizes)).astype(np.int32)).to(DEV)
    b = build_PQ_batch(pos, gp, k, {"kernel": "gaussian", "gaussian_eps": 0.1})
    blk = b["blk"].half().float() if fp16_values else b["blk"]
    pval = b["pval"].half().float() if fp16_values else b["pval"]
    parts = ScatterOperator.parts_from_graph_ptr(gp)
    ops = {"P": ScatterOperator.from_csr(b["rowptr"], b["col"], pval, pos.shape[0], parts=parts),
           "Q": ScatterOperator.from_bsr3(b["rowptr"], b["col"], blk, pos.shape[0], parts=parts),
           "graph_ptr": gp}
    return pos, ops, (b["rowptr"].cpu().numpy(), b["col"].cpu().numpy(), pval.cpu().numpy(), blk.cpu().numpy())


def _rel(a, b):
    return float((a - b).abs().max() / b.abs().max())


@pytest.mark.parametrize("fp16_values", [False, True])
def test_store_batches_equal_scatter_on_the_batch(fp16_values):
    """ragged samples, several chunks, shuffled ids with a repeat; fp32 values (the two banks come from the both-tracks
    launch) and fp16-stored values (no factored form: the two separate chains)"""
    from esc_gnn_amd import ScatteringStore, scatter, plan_chain
    sizes = [64, 128, 37, 200, 96, 128, 50, 500, 128, 7, 300, 64, 128]
    pos, ops, (rp, col, pval, blk) = _dataset(sizes, 5, seed=3, fp16_values=fp16_values)
    kw = {"diffusion_kwargs": dict(scales_type="dyadic", J=3, include_lowpass=True)}
    sstore = ScatteringStore(ops, pos, pos, kw, chunk_graphs=4)
    N = pos.shape[0]
    plan = plan_chain(J=3)
    Wn = 1 + plan.F + plan.F * (plan.F - 1) // 2
    assert sstore.Ws.shape == (N, 3, Wn) and sstore.Wv.shape == (N, 1, 3, Wn)
    # the whole dataset against the reference's formula on the oracle's banks
    ev = np.asarray(plan.events, dtype=np.int32).reshape(-1, 4)
    iu = np.triu_indices(plan.F, 1)
    bs = co.c_chain_csr(rp, col, pval, pos.cpu().numpy(), ev, plan.T, plan.F)
    refs = np.concatenate([pos.cpu().numpy()[..., None], bs, bs[..., iu[0]] * bs[..., iu[1]]], axis=-1)
    assert np.abs(sstore.Ws.cpu().numpy() - refs).max() <= 2e-5 * np.abs(refs).max()
    bv = co.c_chain_bsr3(rp, col, blk, pos.cpu().numpy().reshape(-1), ev, plan.T, plan.F).reshape(N, 3, plan.F)
    refv = np.concatenate([pos.cpu().numpy()[..., None], bv, bv[..., iu[0]] * bv[..., iu[1]]], axis=-1)
    assert np.abs(sstore.Wv.cpu().numpy().reshape(N, 3, Wn) - refv).max() <= 2e-5 * np.abs(refv).max()
    # mini-batches: a gather == _scatter on the views of the same samples
    ids = torch.tensor([7, 2, 11, 2, 0, 9, 12])
    rows = sstore.rows_of(ids)
    assert rows.numel() == sum(sizes[i] for i in ids.tolist())
    Ws, Wv = sstore.batch(ids)
    Ds = scatter(track="scalar", x0=pos[rows], P_or_Q=ops["P"].select(ids), kwargs=kw)
    Dv = scatter(track="vector", x0=pos[rows], P_or_Q=ops["Q"].select(ids), kwargs=kw)
    assert Ws.shape == Ds.shape and Wv.shape == Dv.shape
    assert _rel(Ws, Ds) <= 2e-6 and _rel(Wv, Dv) <= 2e-6
    assert torch.equal(Ws[..., 0], pos[rows])
    with pytest.raises(IndexError):
        sstore.batch([0, len(sizes)])
    with pytest.raises(ValueError):
        ScatteringStore(ops, pos[:-1], pos, kw)


def test_store_behind_the_models_scatter_calls():
    """escat_scatter on a model that carries the store: answered from the store while a batch is active, computed
    otherwise; ablate_second_order_wavelets follows the store's own setting"""
    from esc_gnn_amd import ScatteringStore, escat_scatter, plan_chain

    class _Model:
        ablate_second_order_wavelets = True

    sizes = [96, 128, 64, 128]
    pos, ops, _ = _dataset(sizes, 5, seed=5)
    kw = {"diffusion_kwargs": dict(scales_type="custom", diffusion_scales=torch.tensor([0, 1, 2, 4, 8]))}
    m = _Model()
    m.escat_store = ScatteringStore(ops, pos, pos, kw, ablate_second_order_wavelets=True)
    ids = [3, 1]
    rows = m.escat_store.rows_of(ids)
    Pv, Qv = ops["P"].select(ids), ops["Q"].select(ids)
    direct_s = escat_scatter(m, track="scalar", x0=pos[rows], P_or_Q=Pv, kwargs=kw)        # no active batch: computed
    m.escat_store.set_batch(ids)
    Ws = escat_scatter(m, track="scalar", x0=pos[rows], P_or_Q=Pv, kwargs=kw)
    Wv = escat_scatter(m, track="vector", x0=pos[rows], P_or_Q=Qv, kwargs=kw)
    F = plan_chain(**kw["diffusion_kwargs"]).F                                              # ablated second order: W = 1 + F
    assert Ws.shape == direct_s.shape == (rows.numel(), 3, 1 + F) and _rel(Ws, direct_s) <= 2e-6
    assert Wv.shape == (rows.numel(), 1, 3, 1 + F)
    assert Ws.data_ptr() == m.escat_store._active[0].data_ptr()                             # came out of the store
    # a gradient-carrying call is computed, not looked up
    xg = pos[rows].clone().requires_grad_(True)
    Wg = escat_scatter(m, track="scalar", x0=xg, P_or_Q=Pv, kwargs=kw)
    Wg.sum().backward()
    assert xg.grad is not None and _rel(Wg.detach(), Ws) <= 2e-6
    m.escat_store.set_batch(None)
    assert m.escat_store.lookup("scalar", pos[rows]) is None
