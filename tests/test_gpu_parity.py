"""
GPU parity tests (run on the MI355X box: pytest -m gpu).  Everything goes through
the C ABI of libescat.so via the drop-in Python signatures and is compared with
  (1) the golden vectors generated from the real reference (tests/golden/),
  (2) the CPU oracle (oracle/) on seeded inputs at sizes it finishes in seconds,
  (3) size-independent properties at the benchmark sizes.
Tolerance (BASELINE.json north_star): fp32 relative 1e-5, taken per filter as
max|delta| <= 1e-5 * max|ref| (wavelets are differences of nearly equal vectors,
SURVEY.md §7), index structure bit-exact.
"""
import ctypes
import numpy as np
import pytest
import torch

from oracle import chain_oracle as co
from tests.test_oracle import MODES

pytestmark = pytest.mark.gpu
RTOL = 1e-5


def _dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


def _close(a, ref, rtol=RTOL, atol=1e-30):
    a = a.detach().cpu().numpy() if torch.is_tensor(a) else a
    assert a.shape == ref.shape, (a.shape, ref.shape)
    for f in range(ref.shape[-1]):
        d = np.abs(a[..., f] - ref[..., f]).max()
        assert d <= rtol * np.abs(ref[..., f]).max() + atol, (f, d, np.abs(ref[..., f]).max())


def _ops(sg, dev):
    N = int(sg["N"])
    P = torch.sparse_coo_tensor(torch.from_numpy(sg["P_indices"]), torch.from_numpy(sg["P_values"]), (N, N),
                                dtype=torch.float32).to(dev)
    Q = torch.sparse_coo_tensor(torch.from_numpy(sg["Q_indices"]), torch.from_numpy(sg["Q_values"]), (3 * N, 3 * N),
                                dtype=torch.float32).to(dev)
    return N, P, Q


def test_library_loaded_and_native():
    from esc_gnn_amd import _lib
    lib = _lib.load()
    assert lib.escat_abi_version() == 5


@pytest.mark.parametrize("mode", list(MODES))
def test_get_Batch_Wjxs_vs_goldens(sg, mode):
    from esc_gnn_amd import get_Batch_Wjxs
    dev = _dev()
    N, P, Q = _ops(sg, dev)
    kw = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in MODES[mode].items()}
    for C in (1, 3, 5):
        x = torch.from_numpy(sg[f"x_C{C}"]).to(dev).requires_grad_(True)
        W = get_Batch_Wjxs(x, P, **kw)
        assert W.is_cuda and W.dtype == torch.float32 and W.is_contiguous()
        _close(W, sg[f"P_{mode}_C{C}_W"])
        (W * torch.from_numpy(sg[f"P_{mode}_C{C}_gup"]).to(dev)).sum().backward()
        _close(x.grad, sg[f"P_{mode}_C{C}_gx"], rtol=1e-5)
    xv = torch.from_numpy(sg["x_v"]).to(dev).reshape(3 * N, 1).requires_grad_(True)
    W = get_Batch_Wjxs(xv, Q, **kw)
    _close(W, sg[f"Q_{mode}_W"])
    (W * torch.from_numpy(sg[f"Q_{mode}_gup"]).to(dev)).sum().backward()
    _close(xv.grad, sg[f"Q_{mode}_gx"], rtol=1e-5)


def test_perchannel_and_wide_C(sg):
    from esc_gnn_amd import get_Batch_Wjxs
    dev = _dev()
    N, P, _ = _ops(sg, dev)
    ds2 = torch.from_numpy(sg["perchan_scales"]).to(dev)
    for lp in (True, False):
        W = get_Batch_Wjxs(torch.from_numpy(sg["x_C3"]).to(dev), P, scales_type="custom", diffusion_scales=ds2, J=3,
                           include_lowpass=lp)
        _close(W, sg[f"P_perchan_lp{int(lp)}_C3_W"])
    # C = 11 > 8 channels per launch: column chunks; compare with the torch.sparse restatement on CPU
    g = torch.Generator().manual_seed(3)
    x = torch.randn(N, 11, generator=g)
    ref = co.ref_get_Batch_Wjxs(x, P.cpu(), scales_type="dyadic", J=3)
    xg = x.to(dev).requires_grad_(True)
    W = get_Batch_Wjxs(xg, P, scales_type="dyadic", J=3)
    _close(W, ref.numpy())
    gup = torch.randn(ref.shape, generator=g)
    xr = x.clone().requires_grad_(True)
    (co.ref_get_Batch_Wjxs(xr, P.cpu(), scales_type="dyadic", J=3) * gup).sum().backward()
    (W * gup.to(dev)).sum().backward()
    _close(xg.grad, xr.grad.numpy(), rtol=1e-5)


def test_vs_fp64_dense_reference(sg):
    """both the reference (goldens) and the kernels sit within fp32 noise of the fp64 dense-power bank."""
    from esc_gnn_amd import get_Batch_Wjxs
    dev = _dev()
    N, P, Q = _ops(sg, dev)
    W = get_Batch_Wjxs(torch.from_numpy(sg["x_C3"]).to(dev), P, J=4).cpu().numpy().astype(np.float64)
    r64 = sg["f64_P_dyadic_J4_C3_W"]
    ours = np.abs(W - r64).max(axis=(0, 1)) / np.abs(r64).max(axis=(0, 1))
    theirs = np.abs(sg["P_dyadic_J4_C3_W"] - r64).max(axis=(0, 1)) / np.abs(r64).max(axis=(0, 1))
    assert ours.max() < 5e-6 and ours.max() < 10 * max(theirs.max(), 1e-7)
    Wq = get_Batch_Wjxs(torch.from_numpy(sg["x_v"]).to(dev).reshape(-1, 1), Q, J=4).cpu().numpy().astype(np.float64)
    r64 = sg["f64_Q_dyadic_J4_W"]
    assert (np.abs(Wq - r64).max(axis=(0, 1)) / np.abs(r64).max(axis=(0, 1))).max() < 5e-6


def test_scatter_vs_goldens(sg):
    from esc_gnn_amd import scatter, escat_scatter
    dev = _dev()
    N, P, Q = _ops(sg, dev)
    kw = {"diffusion_kwargs": dict(scales_type="dyadic", J=4, include_lowpass=True, diffusion_scales="dyadic")}
    xs, xv = torch.from_numpy(sg["x_C3"]).to(dev), torch.from_numpy(sg["x_v"]).to(dev)
    for abl in (False, True):
        Ws = scatter(track="scalar", x0=xs, P_or_Q=P, kwargs=kw, ablate_second_order_wavelets=abl)
        Wv = scatter(track="vector", x0=xv, P_or_Q=Q, kwargs=kw, ablate_second_order_wavelets=abl)
        rs, rv = sg[f"scatter_scalar_abl{int(abl)}"], sg[f"scatter_vector_abl{int(abl)}"]
        assert Ws.shape == rs.shape and Wv.shape == rv.shape
        assert np.array_equal(Ws[..., 0].cpu().numpy(), rs[..., 0])            # order 0 is the input, bit for bit
        assert np.abs(Ws.cpu().numpy() - rs).max() <= 2e-5 * np.abs(rs).max()
        assert np.abs(Wv.cpu().numpy() - rv).max() <= 2e-5 * np.abs(rv).max()

    class M:
        ablate_second_order_wavelets = False
    kwc = {"diffusion_kwargs": dict(scales_type="custom", J=4, include_lowpass=True,
                                    diffusion_scales=torch.tensor([0, 1, 2, 4, 6, 8, 16]))}
    Wc = escat_scatter(M(), track="scalar", x0=xs, P_or_Q=P, kwargs=kwc)
    rc = sg["scatter_scalar_custom"]
    assert Wc.shape == rc.shape and np.abs(Wc.cpu().numpy() - rc).max() <= 2e-5 * np.abs(rc).max()
    # per-batch cache: the epoch-0 double call (run_epoch_zero_methods + forward) computes once
    from esc_gnn_amd.scatter import ScatteringCache
    m = M(); m.escat_cache = ScatteringCache()
    W1 = escat_scatter(m, track="scalar", x0=xs, P_or_Q=P, kwargs=kw)
    W2 = escat_scatter(m, track="scalar", x0=xs, P_or_Q=P, kwargs=kw)
    V1 = escat_scatter(m, track="vector", x0=xv, P_or_Q=Q, kwargs=kw)
    V2 = escat_scatter(m, track="vector", x0=xv, P_or_Q=Q, kwargs=kw)
    assert W2 is W1 and V2 is V1 and m.escat_cache.hits == 2 and m.escat_cache.misses == 2
    assert torch.equal(W1, scatter(track="scalar", x0=xs, P_or_Q=P, kwargs=kw))
    xg = xs.clone().requires_grad_(True)
    assert escat_scatter(m, track="scalar", x0=xg, P_or_Q=P, kwargs=kw).requires_grad and m.escat_cache.hits == 2
    # gradient of the orders-0/1/2 epilogue against torch autograd of the same formula
    xs2 = xs.clone().requires_grad_(True)
    W = scatter(track="scalar", x0=xs2, P_or_Q=P, kwargs=kw)
    gup = torch.randn(W.shape, device=dev, generator=torch.Generator(device=dev).manual_seed(5))
    (W * gup).sum().backward()
    xr = torch.from_numpy(sg["x_C3"]).requires_grad_(True)
    Pc = P.cpu()
    (co.ref_scatter("scalar", xr, Pc, kw["diffusion_kwargs"]) * gup.cpu()).sum().backward()
    assert np.abs(xs2.grad.cpu().numpy() - xr.grad.numpy()).max() <= 5e-5 * np.abs(xr.grad.numpy()).max()


def test_single_products_and_errors(sg):
    """K1 / K2 entry points called straight through ctypes; error codes instead of faults."""
    from esc_gnn_amd import _lib, ScatterOperator
    dev = _dev()
    lib = _lib.load()
    N, P, Q = _ops(sg, dev)
    opP, opQ = ScatterOperator.from_torch(P), ScatterOperator.from_torch(Q)
    c, b = opP.csr(), opQ.bsr3()
    st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    for C in (1, 3, 5, 8):
        x = torch.randn(N, C, device=dev)
        y = torch.empty(N, C, device=dev)
        rc = lib.escat_csr_spmm_f32(c.rowptr.data_ptr(), c.col.data_ptr(), c.val.data_ptr(), N, c.nnz,
                                    x.data_ptr(), C, y.data_ptr(), C, C, st)
        assert rc == 0
        ref = torch.sparse.mm(P.cpu(), x.cpu())
        assert (y.cpu() - ref).abs().max() <= 1e-5 * ref.abs().max()
    xv = torch.randn(3 * N, device=dev)
    yv = torch.empty(3 * N, device=dev)
    assert lib.escat_bsr3_spmv_f32(b.rowptr.data_ptr(), b.col.data_ptr(), b.val.data_ptr(), N, b.nnz,
                                   xv.data_ptr(), yv.data_ptr(), st) == 0
    ref = torch.sparse.mm(Q.cpu(), xv.cpu()[:, None])[:, 0]
    assert (yv.cpu() - ref).abs().max() <= 1e-5 * ref.abs().max()
    # bad arguments come back as codes with a message
    assert lib.escat_csr_spmm_f32(c.rowptr.data_ptr(), c.col.data_ptr(), c.val.data_ptr(), N, c.nnz,
                                  x.data_ptr(), 9, y.data_ptr(), 9, 9, st) == -2
    assert b"outside" in lib.escat_last_error()
    assert lib.escat_csr_spmm_f32(None, None, None, N, c.nnz, x.data_ptr(), 8, y.data_ptr(), 8, 8, st) == -1


def _random_batch(dev, B, n, k, seed):
    from esc_gnn_amd import synthetic, ScatterOperator
    b = synthetic.ellipsoid_batch(B, n, k, seed=seed)
    t = lambda a: torch.from_numpy(a).to(dev)
    opP = ScatterOperator.from_csr(t(b["rowptr"]), t(b["col"]), t(b["pval"]), b["N"])
    opQ = ScatterOperator.from_bsr3(t(b["rowptr"]), t(b["col"]), t(b["blk"]), b["N"])
    return b, opP, opQ


@pytest.mark.parametrize("B,n,k,J", [(6, 200, 5, 4), (3, 500, 10, 4), (1, 3000, 16, 6), (40, 7, 3, 2)])
def test_vs_c_oracle_seeded(oracle_lib, B, n, k, J):
    """seeded ellipsoid batches vs the plain-C oracle: fused LDS path (small graphs), multi-launch path
    (n = 3000 does not fit LDS), tiny ragged graphs (n = 7 merged many per workgroup)."""
    from esc_gnn_amd import get_Batch_Wjxs, plan_chain
    dev = _dev()
    b, opP, opQ = _random_batch(dev, B, n, k, seed=100 + n)
    plan = plan_chain(scales_type="dyadic", J=J)
    ev = np.asarray(plan.events, dtype=np.int32).reshape(-1, 4)
    g = torch.Generator().manual_seed(n)
    xs = torch.randn(b["N"], 3, generator=g)
    W = get_Batch_Wjxs(xs.to(dev).requires_grad_(True), opP, J=J)
    ref = co.c_chain_csr(b["rowptr"], b["col"], b["pval"], xs.numpy(), ev, plan.T, plan.F)
    _close(W, ref)
    gup = torch.randn(W.shape, generator=g)
    xg = xs.to(dev).requires_grad_(True)
    (get_Batch_Wjxs(xg, opP, J=J) * gup.to(dev)).sum().backward()
    _close(xg.grad, co.c_chain_csr_bwd(b["rowptr"], b["col"], b["pval"], gup.numpy(), ev, plan.T), rtol=1e-5)
    xv = torch.from_numpy(b["pos"]).reshape(-1, 1)
    xvg = xv.to(dev).requires_grad_(True)
    Wq = get_Batch_Wjxs(xvg, opQ, J=J)
    refq = co.c_chain_bsr3(b["rowptr"], b["col"], b["blk"], xv.numpy().reshape(-1), ev, plan.T, plan.F)
    _close(Wq, refq)
    gq = torch.randn(Wq.shape, generator=g)
    (Wq * gq.to(dev)).sum().backward()
    _close(xvg.grad, co.c_chain_bsr3_bwd(b["rowptr"], b["col"], b["blk"], gq.numpy(), ev, plan.T), rtol=1e-5)


def test_all_three_chain_paths_agree(monkeypatch):
    """register-resident (one thread per row), LDS-fused streaming (lane groups) and multi-launch step kernels
    on the same operator, forward and backward, against each other and the C oracle."""
    from esc_gnn_amd import get_Batch_Wjxs, ScatterOperator, synthetic, plan_chain
    dev = _dev()
    b = synthetic.ellipsoid_batch(8, 300, 10, seed=9)
    t = lambda a: torch.from_numpy(a).to(dev)
    opP = ScatterOperator.from_csr(t(b["rowptr"]), t(b["col"]), t(b["pval"]), b["N"])
    opQ = ScatterOperator.from_bsr3(t(b["rowptr"]), t(b["col"]), t(b["blk"]), b["N"])
    assert opP.csr().max_part_rows == 300 and opQ.bsr3().max_row_nnz >= 11
    monkeypatch.setenv("ESCAT_BSR3_STEP", "lanes")     # the lane-group step kernels (the staged ones: test_gpu_round2.py)
    monkeypatch.setenv("ESCAT_CSR_STEP", "lanes")
    g = torch.Generator().manual_seed(1)
    x = torch.randn(b["N"], 3, generator=g)
    xv = torch.from_numpy(b["pos"]).reshape(-1, 1)
    for kw in (dict(J=4), dict(scales_type="custom", diffusion_scales=torch.tensor([0, 1, 2, 4, 6, 8, 16]))):
        plan = plan_chain(**kw)
        ev = np.asarray(plan.events, dtype=np.int32).reshape(-1, 4)
        gP = torch.randn(b["N"], 3, plan.F, generator=g)
        gQ = torch.randn(3 * b["N"], 1, plan.F, generator=g)
        refP = co.c_chain_csr(b["rowptr"], b["col"], b["pval"], x.numpy(), ev, plan.T, plan.F)
        refQ = co.c_chain_bsr3(b["rowptr"], b["col"], b["blk"], xv.numpy().reshape(-1), ev, plan.T, plan.F)
        refgP = co.c_chain_csr_bwd(b["rowptr"], b["col"], b["pval"], gP.numpy(), ev, plan.T)
        refgQ = co.c_chain_bsr3_bwd(b["rowptr"], b["col"], b["blk"], gQ.numpy(), ev, plan.T)
        outs = {}
        for path in ("reg", "stream", "multi", "multi_packed"):
            monkeypatch.setenv("ESCAT_CHAIN_PATH", path)
            xg = x.to(dev).requires_grad_(True)
            W = get_Batch_Wjxs(xg, opP, **kw)
            W.backward(gP.to(dev))
            xq = xv.to(dev).requires_grad_(True)
            Wq = get_Batch_Wjxs(xq, opQ, **kw)
            Wq.backward(gQ.to(dev))
            _close(W, refP); _close(Wq, refQ)
            _close(xg.grad, refgP, rtol=1e-5); _close(xq.grad, refgQ, rtol=1e-5)
            outs[path] = (W.detach(), Wq.detach())
        # lane-group paths share lane mapping and summation order
        assert (outs["stream"][0] - outs["multi"][0]).abs().max() <= 1e-6 * outs["multi"][0].abs().max()
        assert (outs["stream"][1] - outs["multi"][1]).abs().max() <= 1e-6 * outs["multi"][1].abs().max()
    monkeypatch.delenv("ESCAT_CHAIN_PATH")


_B128_GROUP0 = [0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27]


def _gather_cycles(lds_rows):
    """average bank-slot multiplicity of a (columns, 64) table of LDS rows, per 16-lane ds_read_b128 group
    (distinct rows on one slot serialise; equal rows broadcast)."""
    g0 = np.asarray(_B128_GROUP0)
    g1 = np.setdiff1d(np.arange(32), g0)
    tot, cnt = 0, 0
    for grp in (g0, g1, g0 + 32, g1 + 32):
        sub = lds_rows[:, grp]
        for row in sub:
            u = np.unique(row)
            tot += np.bincount(u & 15, minlength=16).max(); cnt += 1
    return tot / max(cnt, 1)


def test_sell_slot_plan_is_a_valid_permutation_and_cuts_conflicts():
    """the pack's slot plan (escat_sell_pack_fill): every kept entry gets a distinct slot of its row, its LDS row is the
    POSITION of its column's owner in copy A or copy B (layout 1), and the planned gathers hit fewer equal bank slots
    than the natural order."""
    from esc_gnn_amd import ScatterOperator, synthetic
    from esc_gnn_amd.operators import sell_pad_rows, sell_swz_b
    dev = _dev()
    b = synthetic.ellipsoid_batch(3, 500, 10, seed=21)
    t = lambda a: torch.from_numpy(a).to(dev)
    form = ScatterOperator.from_bsr3(t(b["rowptr"]), t(b["col"]), t(b["blk"]), b["N"]).bsr3()
    sell = form.sell(keep_plan=True)
    assert sell.layout == 1
    slot = sell.slot_of_e.cpu().numpy()[:form.nnz]; lrow = sell.lc_of_e.cpu().numpy()[:form.nnz]
    pos = sell.pos_of_row.cpu().numpy().astype(np.int64)
    rp, col, pp = b["rowptr"].astype(np.int64), b["col"].astype(np.int64), form.part_ptr.cpu().numpy().astype(np.int64)
    part_of_row = np.repeat(np.arange(form.n_parts), np.diff(pp))
    for r in range(form.n):
        e0, e1 = rp[r], rp[r + 1]
        kept = min(e1 - e0, sell.dmax)
        sl = slot[e0:e1]
        assert sorted(sl[sl >= 0].tolist()) == list(range(kept))
        gs = pp[part_of_row[r]]; R = sell_pad_rows(int(pp[part_of_row[r] + 1] - gs))
        for e in range(e0, e1):
            if slot[e] >= 0:
                a = int(pos[col[e]])
                assert lrow[e] in (a, R + sell_swz_b(a)), (r, e, a, lrow[e])
    # positions: a permutation of 0..rows-1 inside every part, rows sorted by length (descending, stable)
    for g in range(form.n_parts):
        pr = pos[pp[g]:pp[g + 1]]
        assert sorted(pr.tolist()) == list(range(len(pr)))
        deg = np.minimum(np.diff(rp)[pp[g]:pp[g + 1]], sell.dmax)
        order = np.argsort(pr)
        assert (np.diff(deg[order]) <= 0).all()
    # conflict statistic of the first slice of the first part: planned vs natural order
    perm = sell.perm.cpu().numpy()[:64]
    rows0 = int(pp[1] - pp[0]); w = int(sell.slice_doff[1] - sell.slice_doff[0])
    plan_tab = np.full((w, 64), rows0, dtype=np.int64); nat_tab = plan_tab.copy()
    for lane, r in enumerate(perm):
        if r < 0:
            continue
        for d, e in enumerate(range(rp[r], min(rp[r + 1], rp[r] + sell.dmax))):
            nat_tab[d, lane] = pos[col[e]]
            plan_tab[slot[e], lane] = lrow[e]
    c_nat, c_plan = _gather_cycles(nat_tab), _gather_cycles(plan_tab)
    assert c_plan < 0.8 * c_nat, (c_plan, c_nat)
    # stores: the 8 consecutive lanes of a ds_write_b128 lane group hit 8 distinct 16-byte slots (mod 8) in both copies
    p = np.arange(512)
    for grp in p.reshape(-1, 8):
        assert len(set((grp % 8).tolist())) == 8 and len(set((sell_swz_b(grp) % 8).tolist())) == 8
    assert ((sell_swz_b(p) & 15) != (p & 15)).all() and sorted(sell_swz_b(p).tolist()) == p.tolist()


def test_gpu_pack_layout0_is_bit_identical_to_the_round1_torch_pack():
    """escat_sell_pack_plan / _fill (HIP) against round 1's torch-op pack (tests/torch_sell_reference.py) in the layout
    both know (0): val, lc, lane_meta, perm and the slice offsets bit for bit -- scalar CSR, BSR-3, per-channel CSR
    (three planes) and the sign-bit form; ragged parts, rows longer than dmax."""
    from esc_gnn_amd import ScatterOperator, synthetic
    from esc_gnn_amd.operators import Sell
    from tests.torch_sell_reference import TorchSell
    dev = _dev()
    t = lambda a: torch.from_numpy(a).to(dev)
    b = _concat_batches([synthetic.ellipsoid_batch(3, 180, 9, seed=1), synthetic.ellipsoid_batch(2, 90, 30, seed=2),
                         synthetic.ellipsoid_batch(5, 33, 4, seed=3), synthetic.ellipsoid_batch(2, 400, 10, seed=4)])
    rowptr, col = t(b["rowptr"]), t(b["col"])
    g = torch.Generator().manual_seed(0)
    v3 = torch.randn(b["nnz"], 3, generator=g).to(dev)
    sg = t(b["pval"])[:, None] * torch.where(torch.rand(b["nnz"], 3, generator=g).to(dev) < 0.5, -1.0, 1.0)
    forms = [ScatterOperator.from_csr(rowptr, col, t(b["pval"]), b["N"]).csr(),
             ScatterOperator.from_bsr3(rowptr, col, t(b["blk"]), b["N"]).bsr3(),
             ScatterOperator.from_csr(rowptr, col, v3, b["N"], validate=False).csr(),
             ScatterOperator.from_csr(rowptr, col, sg.contiguous(), b["N"], validate=False).csr()]
    for k, form in enumerate(forms):
        ref = TorchSell(form)
        got = Sell(form, keep_plan=True, layout=0)
        assert got.dmax == ref.dmax and got.n_slices == ref.n_slices and got.slices_per_part == ref.slices_per_part
        assert got.val_planes == ref.val_planes
        assert torch.equal(got.perm, ref.perm), k
        assert torch.equal(got.slice_doff.long(), ref.slice_doff.long()) and torch.equal(got.slice_poff.long(), ref.slice_poff.long())
        assert torch.equal(got.slot_of_e[:form.nnz], ref.slot_of_e[:form.nnz]) and torch.equal(got.lc_of_e[:form.nnz], ref.lc_of_e[:form.nnz])
        assert torch.equal(got.val, ref.val), k
        assert torch.equal(got.lc, ref.lc), k
        assert torch.equal(got.lane_meta, ref.lane_meta), k


def test_transpose_on_symmetric_pattern_matches_the_sorted_transpose():
    """escat_transpose_sym (values moved on A's own pattern) = the sort-based transpose, bit for bit, for CSR and BSR-3;
    a non-symmetric pattern is detected and takes the general path."""
    from esc_gnn_amd import ScatterOperator, synthetic
    dev = _dev()
    b = synthetic.ellipsoid_batch(7, 120, 6, seed=12)
    t = lambda a: torch.from_numpy(a).to(dev)
    for op, form_of, tr in ((ScatterOperator.from_csr(t(b["rowptr"]), t(b["col"]), t(b["pval"]), b["N"]), "csr", "csr_t"),
                            (ScatterOperator.from_bsr3(t(b["rowptr"]), t(b["col"]), t(b["blk"]), b["N"]), "bsr3", "bsr3_t")):
        f, ft = getattr(op, form_of)(), getattr(op, tr)()
        assert ft.rowptr.data_ptr() == f.rowptr.data_ptr()                  # shares the pattern
        n = f.n
        rp = f.rowptr.long()
        rows = torch.repeat_interleave(torch.arange(n, device=dev), rp[1:] - rp[:-1])
        order = torch.argsort(f.col.long() * n + rows)
        want = f.val[order] if f.val.dim() == 1 else f.val[order].transpose(1, 2)
        assert torch.equal(ft.val, want.contiguous())
        assert torch.equal(ft.col.long(), rows[order])
    # upper-triangular operator: not symmetric -> general path, still correct
    rowptr = torch.tensor([0, 2, 3, 4], dtype=torch.int32, device=dev)
    col = torch.tensor([0, 2, 1, 2], dtype=torch.int32, device=dev)
    val = torch.tensor([1., 2., 3., 4.], device=dev)
    opU = ScatterOperator.from_csr(rowptr, col, val, 3)
    ct = opU.csr_t()
    assert ct.rowptr.tolist() == [0, 1, 2, 4] and ct.col.tolist() == [0, 1, 0, 2] and ct.val.tolist() == [1., 3., 2., 4.]


def _concat_batches(bs):
    """block-diagonal concatenation of synthetic batches (ragged graph sizes)."""
    rowptr, col, pval, blk, pos = [np.zeros(1, np.int32)], [], [], [], []
    N = nnz = 0
    for b in bs:
        rowptr.append(b["rowptr"][1:] + nnz); col.append(b["col"] + N)
        pval.append(b["pval"]); blk.append(b["blk"]); pos.append(b["pos"])
        N += b["N"]; nnz += b["nnz"]
    return dict(rowptr=np.concatenate(rowptr).astype(np.int32), col=np.concatenate(col).astype(np.int32),
                pval=np.concatenate(pval), blk=np.concatenate(blk), pos=np.concatenate(pos), N=N, nnz=nnz)


def test_persistent_pipeline_many_ragged_parts():
    """more parts than resident workgroups: every workgroup walks several parts of different sizes (metadata prefetch,
    staged-input DMA, operator refill under the previous epilogue); P backward takes the single-slab branch (two slabs
    of 500 x 3 x 6 floats do not fit twice per CU), Q backward the double-slab one."""
    from esc_gnn_amd import get_Batch_Wjxs, ScatterOperator, synthetic, plan_chain
    dev = _dev()
    b = _concat_batches([synthetic.ellipsoid_batch(300, 500, 8, seed=1), synthetic.ellipsoid_batch(300, 333, 8, seed=2),
                         synthetic.ellipsoid_batch(320, 420, 8, seed=3)])
    t = lambda a: torch.from_numpy(a).to(dev)
    opP = ScatterOperator.from_csr(t(b["rowptr"]), t(b["col"]), t(b["pval"]), b["N"])
    opQ = ScatterOperator.from_bsr3(t(b["rowptr"]), t(b["col"]), t(b["blk"]), b["N"])
    assert opP.csr().n_parts == 920 and opP.csr().max_part_rows == 500
    g = torch.Generator().manual_seed(4)
    x = torch.randn(b["N"], 3, generator=g)
    xv = torch.from_numpy(b["pos"]).reshape(-1, 1)
    for kw in (dict(J=3), dict(scales_type="custom", diffusion_scales=torch.tensor([0]))):     # T = 8 and T = 0
        plan = plan_chain(**kw)
        ev = np.asarray(plan.events, dtype=np.int32).reshape(-1, 4)
        gP = torch.randn(b["N"], 3, plan.F, generator=g)
        gQ = torch.randn(3 * b["N"], 1, plan.F, generator=g)
        xg = x.to(dev).requires_grad_(True)
        W = get_Batch_Wjxs(xg, opP, **kw); W.backward(gP.to(dev))
        xq = xv.to(dev).requires_grad_(True)
        Wq = get_Batch_Wjxs(xq, opQ, **kw); Wq.backward(gQ.to(dev))
        _close(W, co.c_chain_csr(b["rowptr"], b["col"], b["pval"], x.numpy(), ev, plan.T, plan.F))
        _close(Wq, co.c_chain_bsr3(b["rowptr"], b["col"], b["blk"], xv.numpy().reshape(-1), ev, plan.T, plan.F))
        _close(xg.grad, co.c_chain_csr_bwd(b["rowptr"], b["col"], b["pval"], gP.numpy(), ev, plan.T), rtol=1e-5)
        _close(xq.grad, co.c_chain_bsr3_bwd(b["rowptr"], b["col"], b["blk"], gQ.numpy(), ev, plan.T), rtol=1e-5)


def test_operator_batch_views_of_a_stored_operator():
    """mini-batches drawn by part ids from a stored (canonical + packed) operator, with no copy: forward and backward of
    a shuffled subset equal the corresponding rows of the full run bit for bit (graphs are independent)."""
    from esc_gnn_amd import get_Batch_Wjxs, ScatterOperator, synthetic
    dev = _dev()
    bs = [synthetic.ellipsoid_batch(200, 500, 8, seed=11), synthetic.ellipsoid_batch(250, 210, 8, seed=12),
          synthetic.ellipsoid_batch(200, 380, 8, seed=13)]
    b = _concat_batches(bs)
    gp = np.concatenate([[0], np.cumsum([n for bb, n in zip(bs, (500, 210, 380)) for _ in range(bb["N"] // n)])])
    t = lambda a: torch.from_numpy(a).to(dev)
    parts = ScatterOperator.parts_from_graph_ptr(t(gp))
    opP = ScatterOperator.from_csr(t(b["rowptr"]), t(b["col"]), t(b["pval"]), b["N"], parts=parts)
    opQ = ScatterOperator.from_bsr3(t(b["rowptr"]), t(b["col"]), t(b["blk"]), b["N"], parts=parts)
    assert opP.csr().n_parts == 650 and opQ.bsr3().n_parts == 650
    g = torch.Generator().manual_seed(5)
    x = torch.randn(b["N"], 3, generator=g).to(dev)
    xv = torch.from_numpy(b["pos"]).reshape(-1, 1).to(dev)
    ids = torch.randperm(650, generator=g)[:601]
    for J in (3, 0):
        Wf = get_Batch_Wjxs(x, opP, J=J); Wqf = get_Batch_Wjxs(xv, opQ, J=J)
        vP, vQ = opP.select(ids), opQ.select(ids)
        rP, rQ = vP.row_index(), vQ.row_index()
        assert vP.n == rP.numel() and vQ.n == rQ.numel() == 3 * rP.numel()
        xs = x[rP].clone().requires_grad_(True); xq = xv[rQ].clone().requires_grad_(True)
        W = get_Batch_Wjxs(xs, vP, J=J); Wq = get_Batch_Wjxs(xq, vQ, J=J)
        assert torch.equal(W, Wf[rP]) and torch.equal(Wq, Wqf[rQ])
        # backward: upstream gradient of the batch scattered into the store's rows
        G = torch.randn(W.shape, generator=g).to(dev); Gq = torch.randn(Wq.shape, generator=g).to(dev)
        W.backward(G); Wq.backward(Gq)
        xf = x.clone().requires_grad_(True); xqf = xv.clone().requires_grad_(True)
        Gf = torch.zeros_like(Wf); Gf[rP] = G
        Gqf = torch.zeros_like(Wqf); Gqf[rQ] = Gq
        get_Batch_Wjxs(xf, opP, J=J).backward(Gf); get_Batch_Wjxs(xqf, opQ, J=J).backward(Gqf)
        assert torch.equal(xs.grad, xf.grad[rP]) and torch.equal(xq.grad, xqf.grad[rQ])
    # repeats are allowed; an empty batch is a no-op
    rep = opP.select(torch.tensor([5, 5, 7]))
    Wr = get_Batch_Wjxs(x[rep.row_index()], rep, J=2)
    n5 = int(gp[6] - gp[5])
    assert torch.equal(Wr[:n5], Wr[n5:2 * n5])
    assert get_Batch_Wjxs(x[:0], opP.select(torch.zeros(0, dtype=torch.long)), J=2).shape[0] == 0


def test_orders_kernels_all_bank_sizes():
    """escat_scatter_orders[_bwd]_f32: tiled kernels (F <= 9) and the element-wise fallback (F > 9, unaligned output)
    against the reference formula (models/escgnn_modular.py:333-341) and its torch autograd, ragged row counts."""
    from esc_gnn_amd.scatter import _Orders
    dev = _dev()
    g = torch.Generator().manual_seed(3)
    for F in (1, 2, 5, 6, 9, 11):
        for rows, C in ((1, 1), (63, 3), (64, 1), (1000, 3), (257, 5)):
            for second in (True, False):
                x0 = torch.randn(rows, C, generator=g).to(dev).requires_grad_(True)
                W1 = torch.randn(rows, C, F, generator=g).to(dev).requires_grad_(True)
                W = _Orders.apply(x0, W1, second)
                parts = [x0.unsqueeze(-1), W1]
                if second:
                    i, j = torch.triu_indices(F, F, 1)
                    if i.numel():
                        parts.append(W1[..., i.to(dev)] * W1[..., j.to(dev)])
                ref = torch.cat(parts, dim=-1)
                assert W.shape == ref.shape and torch.equal(W, ref), (F, rows, C, second)
                G = torch.randn(ref.shape, generator=g).to(dev)
                gx, gw = torch.autograd.grad(W, (x0, W1), G)
                rx, rw = torch.autograd.grad(ref, (x0, W1), G)
                assert torch.equal(gx, rx)
                assert (gw - rw).abs().max() <= 1e-5 * max(float(rw.abs().max()), 1.0), (F, rows, C, second)


def test_locality_reordering_is_transparent(monkeypatch):
    """a connected block too large for LDS is relabelled breadth-first at ingest (operators._reorder_if_large);
    get_Batch_Wjxs permutes x in and the bank out, so results and gradients stay those of the original order."""
    from esc_gnn_amd import get_Batch_Wjxs, ScatterOperator, synthetic, plan_chain
    dev = _dev()
    monkeypatch.setenv("ESCAT_REORDER", "1")                      # force it at a size the oracle finishes quickly
    b = synthetic.single_cloud(3000, 8, workers=1)
    t = lambda a: torch.from_numpy(a).to(dev)
    opQ = ScatterOperator.from_bsr3(t(b["rowptr"]), t(b["col"]), t(b["blk"]), b["N"])
    opP = ScatterOperator.from_csr(t(b["rowptr"]), t(b["col"]), t(b["pval"]), b["N"])
    Pcoo = torch.sparse_coo_tensor(torch.stack([torch.repeat_interleave(torch.arange(b["N"]), torch.from_numpy(np.diff(b["rowptr"]).astype(np.int64))),
                                                torch.from_numpy(b["col"]).long()]), torch.from_numpy(b["pval"]), (b["N"], b["N"])).to(dev)
    opT = ScatterOperator.from_torch(Pcoo)
    for op in (opQ, opP, opT):
        assert op.row_perm is not None and sorted(op.row_perm.tolist()) == list(range(op.n))
    # locality: mean |row - col| of the relabelled structure is far below the sampler's random order
    f = opQ.bsr3(); rp = f.rowptr.long()
    rows = torch.repeat_interleave(torch.arange(f.n, device=dev), rp[1:] - rp[:-1])
    assert float((rows - f.col.long()).abs().float().mean()) < 0.2 * float(np.abs(np.repeat(np.arange(b["N"]), np.diff(b["rowptr"])) - b["col"]).mean())
    plan = plan_chain(J=3)
    ev = np.asarray(plan.events, dtype=np.int32).reshape(-1, 4)
    g = torch.Generator().manual_seed(8)
    x = torch.randn(b["N"], 3, generator=g)
    xv = torch.from_numpy(b["pos"]).reshape(-1, 1)
    refP = co.c_chain_csr(b["rowptr"], b["col"], b["pval"], x.numpy(), ev, plan.T, plan.F)
    refQ = co.c_chain_bsr3(b["rowptr"], b["col"], b["blk"], xv.numpy().reshape(-1), ev, plan.T, plan.F)
    for op in (opP, opT):
        xg = x.to(dev).requires_grad_(True)
        W = get_Batch_Wjxs(xg, op, J=3)
        _close(W, refP)
        gP = torch.randn(W.shape, generator=g)
        W.backward(gP.to(dev))
        _close(xg.grad, co.c_chain_csr_bwd(b["rowptr"], b["col"], b["pval"], gP.numpy(), ev, plan.T), rtol=1e-5)
    xq = xv.to(dev).requires_grad_(True)
    Wq = get_Batch_Wjxs(xq, opQ, J=3)
    _close(Wq, refQ)
    gQ = torch.randn(Wq.shape, generator=g)
    Wq.backward(gQ.to(dev))
    _close(xq.grad, co.c_chain_bsr3_bwd(b["rowptr"], b["col"], b["blk"], gQ.numpy(), ev, plan.T), rtol=1e-5)


def test_scatter_fused_and_two_kernel_forms_agree(monkeypatch):
    """_scatter through the fused launch (orders epilogue inside the chain kernel) and through bank + orders kernels
    (forced by ESCAT_CHAIN_PATH=stream, and taken by itself for a graph that does not fit LDS): same values, same grads."""
    from esc_gnn_amd import scatter, ScatterOperator, synthetic
    dev = _dev()
    b = synthetic.ellipsoid_batch(20, 300, 8, seed=31)
    t = lambda a: torch.from_numpy(a).to(dev)
    opP = ScatterOperator.from_csr(t(b["rowptr"]), t(b["col"]), t(b["pval"]), b["N"])
    opQ = ScatterOperator.from_bsr3(t(b["rowptr"]), t(b["col"]), t(b["blk"]), b["N"])
    g = torch.Generator().manual_seed(2)
    xs = torch.randn(b["N"], 3, generator=g).to(dev); xv = t(b["pos"])
    res = {}
    for path in ("reg", "stream"):
        monkeypatch.setenv("ESCAT_CHAIN_PATH", path)
        for abl in (False, True):
            for kw in ({"diffusion_kwargs": dict(scales_type="dyadic", J=3)},
                       {"diffusion_kwargs": dict(scales_type="custom", diffusion_scales=torch.tensor([0, 1, 2, 4, 6]))}):
                a = xs.clone().requires_grad_(True); v = xv.clone().requires_grad_(True)
                Ws = scatter(track="scalar", x0=a, P_or_Q=opP, kwargs=kw, ablate_second_order_wavelets=abl)
                Wv = scatter(track="vector", x0=v, P_or_Q=opQ, kwargs=kw, ablate_second_order_wavelets=abl)
                Gs = torch.randn(Ws.shape, generator=torch.Generator().manual_seed(7)).to(dev)
                Gv = torch.randn(Wv.shape, generator=torch.Generator().manual_seed(8)).to(dev)
                torch.autograd.backward([Ws, Wv], [Gs, Gv])
                key = (abl, str(kw["diffusion_kwargs"].get("scales_type")))
                if path == "reg":
                    res[key] = (Ws.detach(), Wv.detach(), a.grad, v.grad)
                    # without a gradient and with ESCAT_ORDERS=epilogue the same call is ONE launch (orders written by the
                    # chain's own epilogue)
                    monkeypatch.setenv("ESCAT_ORDERS", "epilogue")
                    with torch.no_grad():
                        Fs = scatter(track="scalar", x0=xs, P_or_Q=opP, kwargs=kw, ablate_second_order_wavelets=abl)
                        Fv = scatter(track="vector", x0=xv, P_or_Q=opQ, kwargs=kw, ablate_second_order_wavelets=abl)
                    assert torch.equal(Fs[..., 0], xs) and Fs.shape == Ws.shape and Fv.shape == Wv.shape
                    assert (Fs - Ws).abs().max() <= 1e-6 * Ws.abs().max() and (Fv - Wv).abs().max() <= 1e-6 * Wv.abs().max()
                    monkeypatch.delenv("ESCAT_ORDERS")
                    with torch.no_grad():                       # the default: bank + orders kernels, bit-identical to the call above
                        Ds = scatter(track="scalar", x0=xs, P_or_Q=opP, kwargs=kw, ablate_second_order_wavelets=abl)
                        Dv = scatter(track="vector", x0=xv, P_or_Q=opQ, kwargs=kw, ablate_second_order_wavelets=abl)
                    assert torch.equal(Ds, Ws.detach()) and torch.equal(Dv, Wv.detach())
                else:
                    for got, want in zip((Ws.detach(), Wv.detach(), a.grad, v.grad), res[key]):
                        assert got.shape == want.shape
                        assert (got - want).abs().max() <= 2e-5 * want.abs().max()
    monkeypatch.delenv("ESCAT_CHAIN_PATH")
    big = synthetic.single_cloud(3000, 6, workers=1)
    opB = ScatterOperator.from_csr(t(big["rowptr"]), t(big["col"]), t(big["pval"]), big["N"])
    xb = t(big["pos"]).clone().requires_grad_(True)
    Wb = scatter(track="scalar", x0=xb, P_or_Q=opB, kwargs={"diffusion_kwargs": dict(J=2)})
    assert Wb.shape == (3000, 3, 1 + 4 + 6) and torch.equal(Wb[..., 0], xb.detach())
    assert torch.allclose(Wb[..., 5], Wb[..., 1] * Wb[..., 2])
    Wb.sum().backward(); assert torch.isfinite(xb.grad).all()


def test_register_path_long_rows_and_small_parts():
    """rows longer than the register capacity (DMAX) take the streamed tail; parts of many tiny graphs."""
    from esc_gnn_amd import get_Batch_Wjxs, ScatterOperator, synthetic, plan_chain
    dev = _dev()
    b = synthetic.ellipsoid_batch(2, 120, 30, seed=5)            # k=30: rows of 31..60 entries > DMAX=22/32
    t = lambda a: torch.from_numpy(a).to(dev)
    opP = ScatterOperator.from_csr(t(b["rowptr"]), t(b["col"]), t(b["pval"]), b["N"])
    opQ = ScatterOperator.from_bsr3(t(b["rowptr"]), t(b["col"]), t(b["blk"]), b["N"])
    assert opQ.bsr3().max_row_nnz > 32
    plan = plan_chain(J=3)
    ev = np.asarray(plan.events, dtype=np.int32).reshape(-1, 4)
    x = torch.randn(b["N"], 5, generator=torch.Generator().manual_seed(2))
    _close(get_Batch_Wjxs(x.to(dev), opP, J=3), co.c_chain_csr(b["rowptr"], b["col"], b["pval"], x.numpy(), ev, plan.T, plan.F))
    xv = torch.from_numpy(b["pos"]).reshape(-1, 1)
    _close(get_Batch_Wjxs(xv.to(dev), opQ, J=3),
           co.c_chain_bsr3(b["rowptr"], b["col"], b["blk"], xv.numpy().reshape(-1), ev, plan.T, plan.F))


def test_properties_at_benchmark_size():
    """BASELINE cfg2 shape at reduced B (256 x 500, k=10): size-independent properties.
      * P is row-stochastic: constant input -> every band 0, low-pass = the constant;
      * linearity: W(a x + b y) = a W(x) + b W(y);
      * telescoping: sum of bands + low-pass = x;
      * adjointness: <W(x), G> = <x, W^T(G)> ties backward to forward;
      * frameless Q = P (x) I_3 reproduces the scalar track on coordinate columns."""
    from esc_gnn_amd import get_Batch_Wjxs, ScatterOperator, synthetic
    dev = _dev()
    b = synthetic.ellipsoid_batch(256, 500, 10, seed=457893, with_q=True)
    t = lambda a: torch.from_numpy(a).to(dev)
    opP = ScatterOperator.from_csr(t(b["rowptr"]), t(b["col"]), t(b["pval"]), b["N"])
    opQ = ScatterOperator.from_bsr3(t(b["rowptr"]), t(b["col"]), t(b["blk"]), b["N"])
    N = b["N"]
    W1 = get_Batch_Wjxs(torch.full((N, 2), 3.0, device=dev), opP, J=4)
    assert W1[..., :-1].abs().max() < 1e-5 and (W1[..., -1] - 3.0).abs().max() < 1e-5
    x, y = torch.randn(N, 3, device=dev), torch.randn(N, 3, device=dev)
    Wx, Wy = get_Batch_Wjxs(x, opP, J=4), get_Batch_Wjxs(y, opP, J=4)
    Wl = get_Batch_Wjxs(2.0 * x - 0.5 * y, opP, J=4)
    assert (Wl - (2.0 * Wx - 0.5 * Wy)).abs().max() <= 1e-5 * Wl.abs().max()
    assert (Wx.sum(-1) - x).abs().max() < 2e-5
    for op, xin in ((opP, x), (opQ, t(b["pos"]).reshape(-1, 1))):
        xin = xin.clone().requires_grad_(True)
        W = get_Batch_Wjxs(xin, op, J=4)
        G = torch.randn_like(W)
        (W * G).sum().backward()
        lhs = (W.detach().double() * G.double()).sum()
        rhs = (xin.detach().double() * xin.grad.double()).sum()
        assert abs(lhs - rhs) <= 1e-4 * max(abs(lhs), 1.0)
    Wq = get_Batch_Wjxs(t(b["pos"]).reshape(-1, 1), opQ, J=4)
    assert (Wq.sum(-1).reshape(-1) - t(b["pos"]).reshape(-1)).abs().max() < 2e-5
    eye = (t(b["pval"])[:, None, None] * torch.eye(3, device=dev)).contiguous()
    opI = ScatterOperator.from_bsr3(t(b["rowptr"]), t(b["col"]), eye, N)
    Wi = get_Batch_Wjxs(x.reshape(-1, 1), opI, J=4).reshape(N, 3, -1)
    assert (Wi - Wx).abs().max() <= 1e-5 * Wx.abs().max()


def test_empty_and_degenerate_inputs():
    from esc_gnn_amd import get_Batch_Wjxs, ScatterOperator
    dev = _dev()
    # identity operator with isolated rows (rows of A = 0 -> P = I/2 ... here just diag), T = 0 plan, 1-node graphs
    n = 5
    rowptr = torch.arange(n + 1, dtype=torch.int32, device=dev)
    col = torch.arange(n, dtype=torch.int32, device=dev)
    op = ScatterOperator.from_csr(rowptr, col, torch.full((n,), 0.5, device=dev), n)
    x = torch.arange(n * 2, dtype=torch.float32, device=dev).reshape(n, 2)
    W = get_Batch_Wjxs(x, op, J=2)                       # powers 0,1,2,4 of I/2
    exp = torch.stack([x - x / 2, x / 2 - x / 4, x / 4 - x / 16, x / 16], -1)
    assert torch.equal(W, exp)
    W0 = get_Batch_Wjxs(x, op, scales_type="custom", diffusion_scales=torch.tensor([0]))
    assert torch.equal(W0, x.unsqueeze(-1))
    # operator with an empty row
    rowptr = torch.tensor([0, 1, 1, 2], dtype=torch.int32, device=dev)
    col = torch.tensor([0, 2], dtype=torch.int32, device=dev)
    op = ScatterOperator.from_csr(rowptr, col, torch.ones(2, device=dev), 3)
    W = get_Batch_Wjxs(torch.ones(3, 1, device=dev), op, scales_type=None, diffusion_scales=None)
    assert W.reshape(-1).tolist() == [1.0, 0.0, 1.0]
    with pytest.raises(RuntimeError):
        get_Batch_Wjxs(torch.ones(4, 1, device=dev), op, J=1)      # size mismatch
    with pytest.raises(TypeError):
        get_Batch_Wjxs(torch.ones(3, 1, device=dev, dtype=torch.float64), op, J=1)


def _within_fp32_noise_of_f64(W, ref32, ref64, factor=3.0):
    """per band: |ours - f64 oracle| <= factor * |f32 oracle - f64 oracle| + 1e-5 of the band -- the f32 oracle sums a
    row in the reference's ascending-column order, so its distance to the f64 result IS the reference's own rounding
    noise at this size; ours (a different summation order) must sit inside a small multiple of it."""
    a = W.detach().cpu().numpy().astype(np.float64)
    worst = 0.0
    for f in range(ref64.shape[-1]):
        ours = np.abs(a[..., f] - ref64[..., f]).max()
        theirs = np.abs(ref32[..., f].astype(np.float64) - ref64[..., f]).max()
        scale = np.abs(ref64[..., f]).max()
        assert ours <= factor * theirs + 1e-5 * scale, (f, ours, theirs, scale)
        worst = max(worst, ours / max(theirs, 1e-300))
    return worst


def test_cfg1_full_size_vs_c_oracle(oracle_lib):
    """BASELINE cfg1 at its full size (500 graphs x 200 nodes, k=5, seed 457892): pos (C=3), the two Dirac columns
    (C=2, farthest / nearest node from the centroid, ellipsoid_data_classes.py:255-262) and the vector track, with the
    dyadic J=4 bank and the shipped custom scales [0,1,2,4,6,8,16], forward and backward, against the plain-C oracle.
    A late band is the difference of two powers of the input's magnitude and 30x smaller than them here (smooth
    coordinate signal), so a bound of 1e-5 OF THE BAND sits below fp32 rounding of the powers themselves; instead of a
    loosened tolerance, forward results are held against the fp64 oracle: |ours - f64| <= 3 |f32 oracle - f64| + 1e-5
    of the band, where the f32 oracle (the reference's summation order) measures the reference's own rounding noise."""
    from esc_gnn_amd import get_Batch_Wjxs, plan_chain, ScatterOperator, synthetic
    dev = _dev()
    B, n = 500, 200
    b = synthetic.ellipsoid_batch(B, n, 5, seed=457892)
    t = lambda a: torch.from_numpy(a).to(dev)
    opP = ScatterOperator.from_csr(t(b["rowptr"]), t(b["col"]), t(b["pval"]), b["N"])
    opQ = ScatterOperator.from_bsr3(t(b["rowptr"]), t(b["col"]), t(b["blk"]), b["N"])
    pos = torch.from_numpy(b["pos"])
    r = (pos.view(B, n, 3) - pos.view(B, n, 3).mean(1, keepdim=True)).norm(dim=2)
    dirac = torch.zeros(B, n, 2)
    dirac[torch.arange(B), r.argmax(1), 0] = 1.0
    dirac[torch.arange(B), r.argmin(1), 1] = 1.0
    dirac = dirac.view(B * n, 2)
    g = torch.Generator().manual_seed(1)
    worst = 0.0
    for kw in (dict(scales_type="dyadic", J=4), dict(scales_type="custom", diffusion_scales=torch.tensor([0, 1, 2, 4, 6, 8, 16]))):
        plan = plan_chain(**kw)
        ev = np.asarray(plan.events, dtype=np.int32).reshape(-1, 4)
        for xs in (pos, dirac):
            xg = xs.to(dev).requires_grad_(True)
            W = get_Batch_Wjxs(xg, opP, **kw)
            r32 = co.c_chain_csr(b["rowptr"], b["col"], b["pval"], xs.numpy(), ev, plan.T, plan.F)
            r64 = co.c_chain_csr(b["rowptr"], b["col"], b["pval"].astype(np.float64), xs.numpy().astype(np.float64), ev,
                                 plan.T, plan.F, f64=True)
            worst = max(worst, _within_fp32_noise_of_f64(W, r32, r64))
            gup = torch.randn(W.shape, generator=g)
            (W * gup.to(dev)).sum().backward()
            _close(xg.grad, co.c_chain_csr_bwd(b["rowptr"], b["col"], b["pval"], gup.numpy(), ev, plan.T), rtol=1e-5)
        xv = pos.reshape(-1, 1)
        xvg = xv.to(dev).requires_grad_(True)
        Wq = get_Batch_Wjxs(xvg, opQ, **kw)
        q32 = co.c_chain_bsr3(b["rowptr"], b["col"], b["blk"], xv.numpy().reshape(-1), ev, plan.T, plan.F)
        q64 = co.c_chain_bsr3(b["rowptr"], b["col"], b["blk"].astype(np.float64), xv.numpy().reshape(-1).astype(np.float64),
                              ev, plan.T, plan.F, f64=True)
        worst = max(worst, _within_fp32_noise_of_f64(Wq, q32, q64))
        gq = torch.randn(Wq.shape, generator=g)
        (Wq * gq.to(dev)).sum().backward()
        _close(xvg.grad, co.c_chain_bsr3_bwd(b["rowptr"], b["col"], b["blk"], gq.numpy(), ev, plan.T), rtol=1e-5)
    print("worst |ours - f64| / |f32 oracle - f64| over bands:", worst)


def test_properties_at_full_cfg2_size():
    """BASELINE cfg2 at its full size (4096 graphs x 500 nodes, k=10; operators from the GPU builder): the
    size-independent properties of test_properties_at_benchmark_size on both tracks, plus orders 0/1/2."""
    from esc_gnn_amd import get_Batch_Wjxs, scatter, synthetic
    from esc_gnn_amd.process_pyg_data import build_PQ_batch
    dev = _dev()
    B, n = 4096, 500
    pos = torch.from_numpy(synthetic.ellipsoid_clouds(B, n, seed=457893).reshape(-1, 3)).to(dev)
    gp = torch.arange(B + 1, dtype=torch.int32, device=dev) * n
    b = build_PQ_batch(pos, gp, 10, {"kernel": "gaussian", "gaussian_eps": 0.05})
    opP, opQ, N = b["P"], b["Q"], B * n
    W1 = get_Batch_Wjxs(torch.full((N, 2), 3.0, device=dev), opP, J=4)
    assert W1[..., :-1].abs().max() < 1e-5 and (W1[..., -1] - 3.0).abs().max() < 1e-5
    gen = torch.Generator(device=dev).manual_seed(2)
    x, y = torch.randn(N, 3, device=dev, generator=gen), torch.randn(N, 3, device=dev, generator=gen)
    Wx, Wy = get_Batch_Wjxs(x, opP, J=4), get_Batch_Wjxs(y, opP, J=4)
    Wl = get_Batch_Wjxs(2.0 * x - 0.5 * y, opP, J=4)
    assert (Wl - (2.0 * Wx - 0.5 * Wy)).abs().max() <= 1e-5 * Wl.abs().max()
    assert (Wx.sum(-1) - x).abs().max() < 2e-5
    for op, xin in ((opP, x), (opQ, pos.reshape(-1, 1))):
        xin = xin.clone().requires_grad_(True)
        W = get_Batch_Wjxs(xin, op, J=4)
        G = torch.randn(W.shape, device=dev, generator=gen)
        (W * G).sum().backward()
        lhs = (W.detach().double() * G.double()).sum()
        rhs = (xin.detach().double() * xin.grad.double()).sum()
        assert abs(lhs - rhs) <= 1e-4 * max(abs(lhs), 1.0)
    Wq = get_Batch_Wjxs(pos.reshape(-1, 1), opQ, J=4)
    assert (Wq.sum(-1).reshape(-1) - pos.reshape(-1)).abs().max() < 2e-5
    # orders 0/1/2: slot 0 is the input, slots 1..F the bank, the rest the triu products of the bank
    kw = {"diffusion_kwargs": dict(scales_type="dyadic", J=4, include_lowpass=True)}
    S = scatter(track="scalar", x0=x, P_or_Q=opP, kwargs=kw)
    F = Wx.shape[-1]
    iu = torch.triu_indices(F, F, 1, device=dev)
    assert S.shape == (N, 3, 1 + F + F * (F - 1) // 2)
    assert torch.equal(S[..., 0], x) and torch.equal(S[..., 1:1 + F], Wx)
    assert (S[..., 1 + F:] - Wx[..., iu[0]] * Wx[..., iu[1]]).abs().max() <= 1e-6 * S.abs().max()


def test_full_cfg2_batch_sampled_graphs_vs_c_oracle(oracle_lib, monkeypatch):
    """the FULL cfg2 batch (4096 graphs x 500 nodes, k = 10, dyadic J = 4) through the drop-in
    call, forward and gradient, and 32 sampled graphs' rows against the C oracle run on just those diagonal blocks
    (graphs are independent, so a block's rows depend on nothing outside it) -- both tracks, and the vector track on
    both of its routes (the factored form recovered at ingest, and the BSR-3 register kernel)."""
    from esc_gnn_amd import get_Batch_Wjxs, plan_chain, operators, ScatterOperator
    from esc_gnn_amd.process_pyg_data import build_PQ_batch
    from esc_gnn_amd import synthetic
    dev = _dev()
    B, n = 4096, 500
    pos = torch.from_numpy(synthetic.ellipsoid_clouds(B, n, seed=457893).reshape(-1, 3)).to(dev)
    gp = torch.arange(B + 1, dtype=torch.int32, device=dev) * n
    b = build_PQ_batch(pos, gp, 10, {"kernel": "gaussian", "gaussian_eps": 0.05})
    opP, opQ = b["P"], b["Q"]
    plan = plan_chain(scales_type="dyadic", J=4)
    ev = np.asarray(plan.events, dtype=np.int32).reshape(-1, 4)
    sample = np.unique(np.concatenate([[0, B - 1], np.random.default_rng(11).choice(B, 30, replace=False)]))
    cp, cq = opP.csr(), opQ.bsr3()
    rp_all = cp.rowptr.cpu().numpy()
    assert np.array_equal(rp_all, cq.rowptr.cpu().numpy())
    gen = torch.Generator(device=dev).manual_seed(5)

    def blocks_of(g, col_t, val_t):
        e0, e1 = int(rp_all[g * n]), int(rp_all[(g + 1) * n])
        return (rp_all[g * n:(g + 1) * n + 1] - e0, col_t[e0:e1].cpu().numpy() - g * n, val_t[e0:e1].cpu().numpy())

    def check(op, x, oracle_f, oracle_b, col_t, val_t, rows_per):
        xg = x.clone().requires_grad_(True)
        W = get_Batch_Wjxs(xg, op, scales_type="dyadic", J=4)
        G = torch.randn(W.shape, device=dev, generator=gen)
        (W * G).sum().backward()
        for g in sample:
            rp, cl, vl = blocks_of(int(g), col_t, val_t)
            sl = slice(int(g) * n * rows_per, (int(g) + 1) * n * rows_per)
            xin = x[sl].cpu().numpy()
            _close(W[sl], oracle_f(rp, cl, vl, xin.reshape(-1) if rows_per == 3 else xin, ev, plan.T, plan.F))
            _close(xg.grad[sl], oracle_b(rp, cl, vl, G[sl].cpu().numpy(), ev, plan.T), rtol=1e-5)

    check(opP, torch.randn(B * n, 3, device=dev, generator=gen), co.c_chain_csr, co.c_chain_csr_bwd, cp.col, cp.val, 1)
    xv = pos.reshape(-1, 1)
    check(opQ, xv, co.c_chain_bsr3, co.c_chain_bsr3_bwd, cq.col, cq.val, 3)
    # both tracks in one launch per direction (the bench's headline step): the same 32 graphs, both banks and both gradients
    from esc_gnn_amd import get_Batch_Wjxs_both, BothTracksOperator
    assert BothTracksOperator.pair(opP, opQ) is not None
    xs_b = torch.randn(B * n, 3, device=dev, generator=gen)
    xs_g, xv_g = xs_b.clone().requires_grad_(True), xv.clone().requires_grad_(True)
    Ws, Wv = get_Batch_Wjxs_both(xs_g, opP, xv_g, opQ, scales_type="dyadic", J=4)
    Gs, Gv = torch.randn(Ws.shape, device=dev, generator=gen), torch.randn(Wv.shape, device=dev, generator=gen)
    torch.autograd.backward([Ws, Wv], [Gs, Gv])
    for g in sample:
        g = int(g)
        rp, cl, vl = blocks_of(g, cp.col, cp.val)
        sl = slice(g * n, (g + 1) * n)
        _close(Ws[sl], co.c_chain_csr(rp, cl, vl, xs_b[sl].cpu().numpy(), ev, plan.T, plan.F))
        _close(xs_g.grad[sl], co.c_chain_csr_bwd(rp, cl, vl, Gs[sl].cpu().numpy(), ev, plan.T), rtol=1e-5)
        rp, cl, vl = blocks_of(g, cq.col, cq.val)
        sl = slice(g * n * 3, (g + 1) * n * 3)
        _close(Wv[sl], co.c_chain_bsr3(rp, cl, vl, xv[sl].cpu().numpy().reshape(-1), ev, plan.T, plan.F))
        _close(xv_g.grad[sl], co.c_chain_bsr3_bwd(rp, cl, vl, Gv[sl].cpu().numpy(), ev, plan.T), rtol=1e-5)
    from esc_gnn_amd.pyg_utilities import vector_route
    route_default = type(vector_route(opQ, xv, plan.F, plan.events)).__name__
    assert route_default == "FactoredVectorOperator"        # the builder's blocks: the factored form is recovered at ingest
    monkeypatch.setattr(operators, "FACTOR_MODE", "0")
    opQ2 = ScatterOperator.from_bsr3(cq.rowptr, cq.col, cq.val, B * n)
    assert type(vector_route(opQ2, xv, plan.F, plan.events)).__name__ == "ScatterOperator"
    check(opQ2, xv, co.c_chain_bsr3, co.c_chain_bsr3_bwd, cq.col, cq.val, 3)
    print("vector track routes checked at full cfg2 size:", route_default, "and BSR-3 (ESCAT_Q_FACTOR=0)")


def test_large_cloud_properties():
    """BASELINE cfg3 in kind (one connected cloud, k = 16, J = 6, vector track; 200 000 points so that the host kNN stays
    within seconds): the multi-launch path with the breadth-first relabelling, BSR-3 and factored form.  Properties:
    telescoping sum = x, adjointness of backward, agreement of the two operator forms."""
    from esc_gnn_amd import get_Batch_Wjxs, ScatterOperator, FactoredVectorOperator, synthetic
    dev = _dev()
    c = synthetic.single_cloud(200_000, 16, workers=1)
    t = lambda a: torch.from_numpy(a).to(dev)
    none = (torch.zeros(1, dtype=torch.int32, device=dev), 0, 0)
    opQ = ScatterOperator.from_bsr3(t(c["rowptr"]), t(c["col"]), t(c["blk"]), c["N"], parts=none)
    fop = FactoredVectorOperator.from_frames(t(c["rowptr"]), t(c["col"]), t(c["pval"]), t(c["frames"]), parts=none)
    assert opQ.row_perm is not None and fop.signed.row_perm is not None          # relabelled for locality
    xv = t(c["pos"]).reshape(-1, 1)
    Ws = {}
    for name, op in (("bsr3", opQ), ("factored", fop)):
        x = xv.clone().requires_grad_(True)
        W = get_Batch_Wjxs(x, op, J=6)
        assert W.shape == (3 * c["N"], 1, 8)
        assert (W.detach().sum(-1).reshape(-1) - xv.reshape(-1)).abs().max() < 5e-6 * float(xv.abs().max())
        G = torch.randn(W.shape, device=dev, generator=torch.Generator(device=dev).manual_seed(3))
        (W * G).sum().backward()
        lhs = (W.detach().double() * G.double()).sum()
        rhs = (x.detach().double() * x.grad.double()).sum()
        assert abs(lhs - rhs) <= 1e-4 * max(abs(lhs), 1.0)
        Ws[name] = W.detach()
    d = (Ws["factored"] - Ws["bsr3"]).abs().amax(dim=(0, 1)) / Ws["bsr3"].abs().amax(dim=(0, 1))
    assert float(d.max()) <= 1e-5
