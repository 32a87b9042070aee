"""
Round-2 GPU tests: the |.| epilogue flag (ESCAT_EVENT_ABS), the fused-orders guard for rows larger than the LDS tile,
through the drop-in Python signatures and the C ABI, against the oracle.
"""
import ctypes
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
        assert d <= rtol * np.abs(ref[..., f]).max() + 1e-30, (f, d)


@pytest.mark.parametrize("shape", ["register", "multi_launch"])
def test_apply_abs_flag_vs_abs_of_oracle_bank(oracle_lib, shape, monkeypatch):
    """apply_abs=True (off by default, not a reference argument; the caller shape is models/comparisons/legs.py:299-302)
    = torch.abs of the oracle's signed bank, on the register-resident path and on the multi-launch path, P and Q,
    dyadic and custom scales; with a gradient to x the modulus goes through autograd (sign(W) * upstream)."""
    from esc_gnn_amd import get_Batch_Wjxs, ScatterOperator, synthetic, plan_chain
    dev = _dev()
    if shape == "multi_launch":
        monkeypatch.setenv("ESCAT_CHAIN_PATH", "multi")
    b = synthetic.ellipsoid_batch(6, 150, 6, seed=77)
    t = lambda a: torch.from_numpy(a).to(dev)
    opP = ScatterOperator.from_csr(t(b["rowptr"]), t(b["col"]), t(b["pval"]), b["N"])
    opQ = ScatterOperator.from_bsr3(t(b["rowptr"]), t(b["col"]), t(b["blk"]), b["N"])
    x = torch.randn(b["N"], 3, generator=torch.Generator().manual_seed(5))
    xv = torch.from_numpy(b["pos"]).reshape(-1, 1)
    for kw in (dict(scales_type="dyadic", J=3), dict(scales_type="custom", diffusion_scales=torch.tensor([0, 1, 2, 4, 6]))):
        plan = plan_chain(**kw)
        ev = np.asarray(plan.events, dtype=np.int32).reshape(-1, 4)
        refP = np.abs(co.c_chain_csr(b["rowptr"], b["col"], b["pval"], x.numpy(), ev, plan.T, plan.F))
        refQ = np.abs(co.c_chain_bsr3(b["rowptr"], b["col"], b["blk"], xv.numpy().reshape(-1), ev, plan.T, plan.F))
        W = get_Batch_Wjxs(x.to(dev), opP, apply_abs=True, **kw)
        assert float(W.min()) >= 0.0
        _close(W, refP)
        _close(get_Batch_Wjxs(xv.to(dev), opQ, apply_abs=True, **kw), refQ)
        # the default stays signed (the reference's _scatter keeps signs)
        assert float(get_Batch_Wjxs(x.to(dev), opP, **kw).min()) < 0.0
    xg = x.to(dev).requires_grad_(True)
    Wg = get_Batch_Wjxs(xg, opP, J=3, apply_abs=True)
    Ws = get_Batch_Wjxs(x.to(dev), opP, J=3)
    G = torch.randn(Wg.shape, device=dev, generator=torch.Generator(device=dev).manual_seed(1))
    (Wg * G).sum().backward()
    xr = x.to(dev).requires_grad_(True)
    (get_Batch_Wjxs(xr, opP, J=3) * (G * torch.sign(Ws))).sum().backward()
    assert torch.equal(xg.grad, xr.grad)


def test_abs_event_rejected_by_backward_and_when_not_last():
    from esc_gnn_amd import _lib
    lib = _lib.load()
    dev = _dev()
    n = 4
    rowptr = torch.arange(n + 1, dtype=torch.int32, device=dev)
    col = torch.arange(n, dtype=torch.int32, device=dev)
    val = torch.full((n,), 0.5, device=dev)
    x = torch.ones(n, 1, device=dev)
    out = torch.empty(n, 1, 1, device=dev)
    vp = lambda t: ctypes.c_void_p(t.data_ptr())

    def call(fn, events):
        arr = (_lib.EscatEvent * len(events))()
        for k, e in enumerate(events):
            arr[k].step, arr[k].slot, arr[k].sign, arr[k].chan = e
        ws = torch.empty(64, device=dev)
        return fn(vp(rowptr), vp(col), vp(val), n, n, vp(x), 1, 1, ctypes.cast(arr, ctypes.c_void_p), len(events), 1, vp(out),
                  ctypes.c_void_p(0), 0, 0, ctypes.c_void_p(0), vp(ws), 256, ctypes.c_void_p(0))
    assert call(lib.escat_chain_csr_f32, [(0, 0, 1, -1), (1, 0, -1 + 256, -1)]) == 0
    torch.cuda.synchronize()
    assert out.reshape(-1).tolist() == [0.5] * n                       # |x - x/2|
    assert call(lib.escat_chain_csr_f32, [(0, 0, 1 + 256, -1), (1, 0, -1, -1)]) == -1     # |.| must close the slot
    assert call(lib.escat_chain_csr_bwd_f32, [(0, 0, 1, -1), (1, 0, -1 + 256, -1)]) == -1  # forward only


def test_fused_orders_with_rows_larger_than_the_lds_tile():
    """C = 1, parts of <= 15 rows and F >= 11 make one output row (1 + F + F(F-1)/2 floats) larger than the
    LDS tile of the fused orders epilogue; the library rejects it (ESCAT_ERANGE) and scatter() takes the two-kernel
    form.  Compared with the reference formula on the oracle's bank."""
    from esc_gnn_amd import scatter, ScatterOperator, synthetic, plan_chain, _lib
    from esc_gnn_amd.pyg_utilities import _chain_launch, NoFusedOrders
    dev = _dev()
    b = synthetic.ellipsoid_batch(40, 8, 3, seed=9)                 # 8-node graphs
    t = lambda a: torch.from_numpy(a).to(dev)
    gp = torch.arange(41, dtype=torch.int32, device=dev) * 8
    opP = ScatterOperator.from_csr(t(b["rowptr"]), t(b["col"]), t(b["pval"]), b["N"],
                                   parts=ScatterOperator.parts_from_graph_ptr(gp))       # one part per graph: 8 rows
    scales = torch.tensor([0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 12])                         # F = 12
    kw = {"diffusion_kwargs": dict(scales_type="custom", diffusion_scales=scales)}
    plan = plan_chain(**kw["diffusion_kwargs"])
    assert plan.F >= 11
    x = torch.randn(b["N"], 1, generator=torch.Generator().manual_seed(4))
    with pytest.raises(NoFusedOrders):
        _chain_launch(opP, plan.events, plan.T, plan.F, x.to(dev), 1, False, orders=2)
    W = scatter(track="scalar", x0=x.to(dev), P_or_Q=opP, kwargs=kw)                      # falls back to bank + orders
    ev = np.asarray(plan.events, dtype=np.int32).reshape(-1, 4)
    bank = co.c_chain_csr(b["rowptr"], b["col"], b["pval"], x.numpy(), ev, plan.T, plan.F)
    iu = np.triu_indices(plan.F, 1)
    ref = np.concatenate([x.numpy()[..., None], bank, bank[..., iu[0]] * bank[..., iu[1]]], axis=-1)
    assert W.shape == ref.shape
    assert np.abs(W.cpu().numpy() - ref).max() <= 2e-5 * np.abs(ref).max()


def test_closed_row_ranges_gpu_kernel_matches_the_torch_scan():
    """diagonal blocks of a block-diagonal operator: escat_row_cover + prefix sum + escat_merge_row_ranges (GPU path)
    against the cummax / cummin formulation (the CPU path), ragged graphs, several merge targets, and a non-symmetric
    pattern whose blocks are only visible from the columns of later rows."""
    from esc_gnn_amd import synthetic
    from esc_gnn_amd.operators import _closed_row_ranges, _PARTS_CACHE
    from tests.test_gpu_parity import _concat_batches
    dev = _dev()
    b = _concat_batches([synthetic.ellipsoid_batch(5, 37, 4, seed=1), synthetic.ellipsoid_batch(3, 300, 8, seed=2),
                         synthetic.ellipsoid_batch(9, 12, 3, seed=3)])
    rp, col = torch.from_numpy(b["rowptr"]), torch.from_numpy(b["col"])
    for target in (1, 64, 256, 100000):
        _PARTS_CACHE.clear()
        want = _closed_row_ranges(rp, col, b["N"], target)
        got = _closed_row_ranges(rp.to(dev), col.to(dev), b["N"], target)
        assert got[1:] == want[1:] and torch.equal(got[0].cpu(), want[0])
    # rows 0..2 only reach each other through row 2's first column and row 0's last one; row 3 alone; rows 4..5 coupled
    rowptr = torch.tensor([0, 2, 3, 5, 6, 8, 9], dtype=torch.int32)
    colx = torch.tensor([0, 2, 1, 0, 2, 3, 4, 5, 5], dtype=torch.int32)
    _PARTS_CACHE.clear()
    want = _closed_row_ranges(rowptr, colx, 6, 1)
    got = _closed_row_ranges(rowptr.to(dev), colx.to(dev), 6, 1)
    assert want[0].tolist() == [0, 3, 4, 6] and got[0].cpu().tolist() == [0, 3, 4, 6] and got[1:] == want[1:]


def test_shared_plan_refill_equals_a_fresh_pack():
    """an operator and its transpose on a symmetric pattern (and P / Q of one batch) share one pack plan: the second
    pack is a values-only refill (escat_sell_pack_refill) aliasing lc / lane_meta -- it must equal a pack built from
    scratch, and chains on it must match the oracle (covered by the parity suite, which runs on shared plans)."""
    from esc_gnn_amd import ScatterOperator, synthetic
    from esc_gnn_amd.operators import Sell, _PLAN_CACHE
    dev = _dev()
    b = synthetic.ellipsoid_batch(9, 260, 9, seed=5)
    t = lambda a: torch.from_numpy(a).to(dev)
    rowptr, col = t(b["rowptr"]), t(b["col"])
    _PLAN_CACHE.clear()
    opP = ScatterOperator.from_csr(rowptr, col, t(b["pval"]), b["N"])
    opQ = ScatterOperator.from_bsr3(rowptr, col, t(b["blk"]), b["N"])
    sP = opP.csr().sell()                                     # plans
    sPt = opP.csr_t().sell()                                  # shares P's plan (same rowptr / col tensors)
    assert sPt.lc.data_ptr() == sP.lc.data_ptr() and sPt.lane_meta.data_ptr() == sP.lane_meta.data_ptr()
    sQ = opQ.bsr3().sell()                                    # same pattern and capacity -> shares too
    sQt = opQ.bsr3_t().sell()
    if sQ.dmax == sP.dmax:
        assert sQ.lc.data_ptr() == sP.lc.data_ptr() and sQt.lc.data_ptr() == sP.lc.data_ptr()
    for form, shared in ((opP.csr_t(), sPt), (opQ.bsr3(), sQ), (opQ.bsr3_t(), sQt)):
        fresh = Sell(form, keep_plan=True)                    # keep_plan: never shared
        assert fresh.lc.data_ptr() != shared.lc.data_ptr()
        assert fresh.dmax == shared.dmax and torch.equal(fresh.val, shared.val)
        assert torch.equal(fresh.lc, shared.lc) and torch.equal(fresh.lane_meta, shared.lane_meta)
        assert torch.equal(fresh.pos_of_row, shared.pos_of_row)


def test_gpu_coo_ingest_matches_the_coalesced_structure():
    """escat_ingest_coo (one HIP pass over the reference's COO layouts: sorted P, block-emission Q, both flagged
    un-coalesced, int64) gives the index structure of torch's own coalesce -- bit-exact -- and anything else (shuffled
    entries, duplicates) falls back to the general route with the same result."""
    from esc_gnn_amd import ScatterOperator, synthetic
    dev = _dev()
    b = synthetic.ellipsoid_batch(6, 90, 6, seed=3)
    N, nnz = b["N"], b["nnz"]
    rows = np.repeat(np.arange(N), np.diff(b["rowptr"])).astype(np.int64)
    cols = b["col"].astype(np.int64)
    P = torch.sparse_coo_tensor(torch.from_numpy(np.stack([rows, cols])), torch.from_numpy(b["pval"]), (N, N))
    a = np.arange(3)
    qr = ((3 * rows)[:, None, None] + a[None, :, None] + 0 * a[None, None, :]).reshape(-1)
    qc = ((3 * cols)[:, None, None] + 0 * a[None, :, None] + a[None, None, :]).reshape(-1)
    Q = torch.sparse_coo_tensor(torch.from_numpy(np.stack([qr, qc])), torch.from_numpy(b["blk"].reshape(-1)), (3 * N, 3 * N))
    assert not P.is_coalesced() and not Q.is_coalesced()
    opP = ScatterOperator.from_torch(P.to(dev))
    c = opP.csr()
    assert np.array_equal(c.rowptr.cpu().numpy(), b["rowptr"]) and np.array_equal(c.col.cpu().numpy(), b["col"])
    assert np.array_equal(c.val.cpu().numpy(), b["pval"])
    assert c.n_parts == ScatterOperator.from_torch(P.coalesce().to(dev)).csr().n_parts      # same parts as the general route
    opQ = ScatterOperator.from_torch(Q.to(dev))
    assert opQ.has_bsr3
    q = opQ.bsr3()
    assert np.array_equal(q.rowptr.cpu().numpy(), b["rowptr"]) and np.array_equal(q.col.cpu().numpy(), b["col"])
    assert np.array_equal(q.val.cpu().numpy(), b["blk"])
    # the same structure as torch's coalesce of the scalar COO
    Qc = Q.coalesce()
    cs = opQ.csr()
    rp = cs.rowptr.long().cpu()
    r2 = torch.repeat_interleave(torch.arange(3 * N), rp[1:] - rp[:-1])
    assert torch.equal(r2, Qc.indices()[0]) and torch.equal(cs.col.long().cpu(), Qc.indices()[1])
    assert torch.equal(cs.val.cpu(), Qc.values())
    # shuffled entries with a duplicate pair: not the fast layout -> coalesce route, same canonical operator
    perm = torch.randperm(nnz, generator=torch.Generator().manual_seed(0))
    idx = torch.from_numpy(np.stack([rows, cols]))[:, perm]
    val = torch.from_numpy(b["pval"])[perm]
    idx = torch.cat([idx, idx[:, :1]], 1); val = torch.cat([val * 1.0, val[:1] * 0.0])
    Ps = torch.sparse_coo_tensor(idx, val, (N, N))
    cs2 = ScatterOperator.from_torch(Ps.to(dev)).csr()
    assert np.array_equal(cs2.rowptr.cpu().numpy(), b["rowptr"]) and np.array_equal(cs2.col.cpu().numpy(), b["col"])
    assert np.array_equal(cs2.val.cpu().numpy(), b["pval"])


@pytest.mark.parametrize("blocks3", [0, 1])
def test_gpu_coo_ingest_flags_out_of_range_rows_without_writing_outside_rowptr(blocks3):
    """a negative or >= n row index sets *bad and nothing is written before or after rowptr (guard words on
    both sides of the n + 1 entries), for the entry itself and for the entry after it (whose 'previous row' it is)."""
    from esc_gnn_amd import _lib
    lib = _lib.load()
    dev = _dev()
    n = 40
    s = 3 if blocks3 else 1
    rows = np.repeat(np.arange(n), 2).astype(np.int64)
    cols = np.stack([np.arange(n), (np.arange(n) + 1) % n], 1)
    cols.sort(axis=1)
    cols = cols.reshape(-1).astype(np.int64)
    vp = lambda t: ctypes.c_void_p(t.data_ptr())
    GUARD = 0x5A5A5A5A

    def run(r, c):
        if blocks3:
            a = np.arange(3)
            r = ((3 * r)[:, None, None] + a[None, :, None] + 0 * a[None, None, :]).reshape(-1)
            c = ((3 * c)[:, None, None] + 0 * a[None, :, None] + a[None, None, :]).reshape(-1)
        rt, ct = torch.from_numpy(r).to(dev), torch.from_numpy(c).to(dev)
        buf = torch.full((64 + n + 1 + 64,), GUARD, dtype=torch.int32, device=dev)
        col32 = torch.full((64 + len(rows) + 64,), GUARD, dtype=torch.int32, device=dev)
        bad = torch.zeros(1, dtype=torch.int32, device=dev)
        rc = lib.escat_ingest_coo(vp(rt), vp(ct), len(r), s * n, blocks3, vp(buf[64:]), vp(col32[64:]), vp(bad),
                                  ctypes.c_void_p(0))
        assert rc == 0
        torch.cuda.synchronize()
        assert bool((buf[:64] == GUARD).all()) and bool((buf[64 + n + 1:] == GUARD).all())
        assert bool((col32[:64] == GUARD).all()) and bool((col32[64 + len(rows):] == GUARD).all())
        return int(bad.item()), buf[64:64 + n + 1].cpu().numpy()

    ok, rp = run(rows, cols)
    assert ok == 0 and np.array_equal(rp, 2 * np.arange(n + 1))
    for at, value in ((10, -7), (10, -1), (11, n + 5), (0, -3), (len(rows) - 1, n), (len(rows) - 1, -2)):
        r = rows.copy()
        r[at] = value
        assert run(r, cols)[0] == 1, (at, value)
    r = rows.copy()
    r[10], r[31] = -5, n + 2                                   # one negative and one >= n in the same tensor
    assert run(r, cols)[0] == 1


def _coo_of(b):
    N = b["N"]
    rows = np.repeat(np.arange(N), np.diff(b["rowptr"])).astype(np.int64)
    cols = b["col"].astype(np.int64)
    a = np.arange(3)
    qr = ((3 * rows)[:, None, None] + a[None, :, None] + 0 * a[None, None, :]).reshape(-1)
    qc = ((3 * cols)[:, None, None] + 0 * a[None, :, None] + a[None, None, :]).reshape(-1)
    return rows, cols, qr, qc


def test_single_pass_ingest_cover_and_row_blocks_match_the_scan():
    """round 3: escat_ingest_coo_cover (CSR + row-cover marks in one pass) and escat_row_blocks (marks -> parts in one
    call) give the rowptr / col of escat_ingest_coo and the parts of the cummax / cummin formulation (CPU path of
    _closed_row_ranges), on ragged graphs and several merge targets."""
    from esc_gnn_amd import _lib, synthetic
    from esc_gnn_amd.operators import _closed_row_ranges, _row_blocks_from_marks, _PARTS_CACHE
    from tests.test_gpu_parity import _concat_batches
    lib = _lib.load()
    dev = _dev()
    b = _concat_batches([synthetic.ellipsoid_batch(5, 37, 4, seed=1), synthetic.ellipsoid_batch(3, 300, 8, seed=2),
                         synthetic.ellipsoid_batch(40, 12, 3, seed=3), synthetic.ellipsoid_batch(2, 2100, 6, seed=4)])
    rows, cols, _, _ = _coo_of(b)
    N, nnz = b["N"], b["nnz"]
    vp = lambda t: ctypes.c_void_p(t.data_ptr())
    rt, ct = torch.from_numpy(rows).to(dev), torch.from_numpy(cols).to(dev)
    for target in (1, 64, 256, 100000):
        rowptr = torch.empty(N + 1, dtype=torch.int32, device=dev)
        col32 = torch.empty(nnz, dtype=torch.int32, device=dev)
        bad = torch.zeros(1, dtype=torch.int32, device=dev)
        delta = torch.full((N + 1,), 12345, dtype=torch.int32, device=dev)          # the call zeroes it
        assert lib.escat_ingest_coo_cover(vp(rt), vp(ct), nnz, N, vp(rowptr), vp(col32), vp(bad), vp(delta), ctypes.c_void_p(0)) == 0
        assert int(bad.item()) == 0
        assert np.array_equal(rowptr.cpu().numpy(), b["rowptr"]) and np.array_equal(col32.cpu().numpy(), b["col"])
        _PARTS_CACHE.clear()
        got = _row_blocks_from_marks(delta, rowptr, col32, N, target)
        want = _closed_row_ranges(torch.from_numpy(b["rowptr"]), torch.from_numpy(b["col"]), N, target)
        assert got[1:] == want[1:] and torch.equal(got[0].cpu(), want[0]), target
        assert _PARTS_CACHE["last"][3] is got                                       # where Q of the batch finds it


def test_like_ingest_recognises_the_pattern_of_the_batch():
    """round 3: escat_ingest_coo_like -- Q in block-emission order on P's pattern gives flags 0; blocks out of order or
    with a broken 3x3 layout set bit 0; a well-formed Q on another pattern sets bit 1 (and only that).  Through
    from_torch, Q then SHARES P's rowptr / col / parts tensors, and a Q that arrives alone takes the general route to
    the same canonical form."""
    from esc_gnn_amd import _lib, synthetic, ScatterOperator
    from esc_gnn_amd.operators import _PARTS_CACHE, _PLAN_CACHE
    lib = _lib.load()
    dev = _dev()
    b = synthetic.ellipsoid_batch(7, 110, 6, seed=8)
    rows, cols, qr, qc = _coo_of(b)
    N, nnz = b["N"], b["nnz"]
    vp = lambda t: ctypes.c_void_p(t.data_ptr())
    rp, cl = torch.from_numpy(b["rowptr"]).to(dev), torch.from_numpy(b["col"]).to(dev)

    def flags_of(r, c, ref_rp=rp, ref_cl=cl):
        rt, ct = torch.from_numpy(r).to(dev), torch.from_numpy(c).to(dev)
        fl = torch.zeros(1, dtype=torch.int32, device=dev)
        assert lib.escat_ingest_coo_like(vp(rt), vp(ct), len(r), 3 * N, vp(ref_rp), vp(ref_cl), ref_cl.numel(), vp(fl), ctypes.c_void_p(0)) == 0
        return int(fl.item())

    assert flags_of(qr, qc) == 0
    sw = np.arange(nnz); sw[[5, 6]] = [6, 5]                                      # two blocks swapped: well-formed, other pattern
    swap = (sw[:, None] * 9 + np.arange(9)[None, :]).reshape(-1)
    assert flags_of(qr[swap], qc[swap]) == 2
    r2 = qr.copy(); r2[9 * 40 + 4] += 1                                           # a block that is not 3x3 row-major
    assert flags_of(r2, qc) & 1
    c2 = qc.copy(); c2[9 * 17:9 * 18] += 3                                        # a block moved one block column to the right
    f = flags_of(qr, c2)
    assert f == 2 or (b["col"][17] + 1 >= N and f & 1)
    cl2 = cl.clone(); cl2[100] = (cl2[100] + 1) % N                               # the reference differs in one column
    assert flags_of(qr, qc, rp, cl2) == 2
    r3 = qr.copy(); r3[-9:] = -3                                                  # out of range
    assert flags_of(r3, qc) & 1
    # NULL / size errors come back as codes
    fl = torch.zeros(1, dtype=torch.int32, device=dev)
    rt = torch.from_numpy(qr).to(dev)
    assert lib.escat_ingest_coo_like(vp(rt), vp(rt), len(qr), 3 * N, vp(rp), vp(cl), nnz - 1, vp(fl), ctypes.c_void_p(0)) == -1

    P = torch.sparse_coo_tensor(torch.from_numpy(np.stack([rows, cols])), torch.from_numpy(b["pval"]), (N, N)).to(dev)
    Q = torch.sparse_coo_tensor(torch.from_numpy(np.stack([qr, qc])), torch.from_numpy(b["blk"].reshape(-1)), (3 * N, 3 * N)).to(dev)
    _PARTS_CACHE.clear(); _PLAN_CACHE.clear()
    opQ_alone = ScatterOperator.from_torch(Q)
    _PARTS_CACHE.clear(); _PLAN_CACHE.clear()
    opP = ScatterOperator.from_torch(P)
    opQ = ScatterOperator.from_torch(Q)
    cp, cq, ca = opP.csr(), opQ.bsr3(), opQ_alone.bsr3()
    assert cq.rowptr is cp.rowptr and cq.col is cp.col and cq.part_ptr is cp.part_ptr
    assert ca.rowptr is not cp.rowptr and torch.equal(ca.rowptr, cq.rowptr) and torch.equal(ca.col, cq.col)
    assert torch.equal(ca.val, cq.val) and torch.equal(ca.part_ptr, cq.part_ptr) and ca.n_parts == cq.n_parts
    assert cq.sell().lc.data_ptr() == cp.sell().lc.data_ptr() or cq.sell().dmax != cp.sell().dmax      # one plan for both


def test_direct_chain_without_a_value_pack_equals_the_packed_chain(oracle_lib, monkeypatch):
    """round 3 (escat_sell.ent_of_slot): a Q that is seen once and arrives after its batch's P runs the BSR-3 register
    kernel straight from the canonical blocks, on P's plan -- bit for bit what the same plan gives with a value pack
    (ESCAT_Q_DIRECT=0), forward, backward and with the orders epilogue fused, and within 1e-5 of the C oracle.  Ragged
    graphs, 512- and 256-thread parts, rows longer than the capacity (tails)."""
    from esc_gnn_amd import get_Batch_Wjxs, scatter, synthetic, plan_chain, operators, as_operator
    from tests.test_gpu_parity import _concat_batches
    dev = _dev()
    cases = [_concat_batches([synthetic.ellipsoid_batch(5, 300, 8, seed=1), synthetic.ellipsoid_batch(3, 500, 10, seed=2),
                              synthetic.ellipsoid_batch(20, 30, 4, seed=3)]),
             synthetic.ellipsoid_batch(12, 200, 6, seed=4),
             synthetic.ellipsoid_batch(2, 400, 26, seed=5)]
    kw = dict(scales_type="dyadic", J=3)
    plan = plan_chain(**kw)
    ev = np.asarray(plan.events, dtype=np.int32).reshape(-1, 4)
    n_shared = 0
    for b in cases:
        rows, cols, qr, qc = _coo_of(b)
        N = b["N"]
        Pi, Qi = torch.from_numpy(np.stack([rows, cols])).to(dev), torch.from_numpy(np.stack([qr, qc])).to(dev)
        Pv, Qv = torch.from_numpy(b["pval"]).to(dev), torch.from_numpy(b["blk"].reshape(-1)).to(dev)
        xv = torch.from_numpy(b["pos"]).reshape(-1, 1).to(dev)
        xs = torch.from_numpy(b["pos"]).to(dev)
        outs = {}
        for mode in ("auto", "0"):
            monkeypatch.setattr(operators, "DIRECT_MODE", mode)
            operators._PLAN_CACHE.clear(); operators._PARTS_CACHE.clear()
            P = torch.sparse_coo_tensor(Pi, Pv, (N, N), check_invariants=False)
            Q = torch.sparse_coo_tensor(Qi, Qv, (3 * N, 3 * N), check_invariants=False)
            get_Batch_Wjxs(xs, P, **kw)
            x = xv.clone().requires_grad_(True)
            W = get_Batch_Wjxs(x, Q, **kw)
            opq = Q._escat_operator
            # (P's and Q's packs share a plan when the capacity rule picks the same dmax for both: not for every histogram)
            shared = opq.bsr3().sell().lc.data_ptr() == P._escat_operator.csr().sell().lc.data_ptr()
            n_shared += int(shared)
            assert opq.policy == "single_use" and opq.bsr3().sell().direct == (mode == "auto" and shared)
            assert (opq.bsr3().sell().val is None) == (mode == "auto" and shared)
            G = torch.randn(W.shape, device=dev, generator=torch.Generator(device=dev).manual_seed(3))
            (W * G).sum().backward()
            assert opq.bsr3_t().sell().direct == (mode == "auto")      # the transpose always finds a plan: Q's own, or P's
            with torch.no_grad():
                Q2 = torch.sparse_coo_tensor(Qi, Qv, (3 * N, 3 * N), check_invariants=False)
                monkeypatch.setenv("ESCAT_ORDERS", "epilogue")
                S = scatter(track="vector", x0=xs, P_or_Q=Q2, kwargs=dict(diffusion_kwargs=kw))
                monkeypatch.delenv("ESCAT_ORDERS")
            outs[mode] = (W.detach(), x.grad.clone(), S, G)
        assert torch.equal(outs["auto"][0], outs["0"][0]) and torch.equal(outs["auto"][1], outs["0"][1])
        assert torch.equal(outs["auto"][2], outs["0"][2])
        W, gx, S, G = outs["auto"]
        _close(W, co.c_chain_bsr3(b["rowptr"], b["col"], b["blk"], xv.cpu().numpy().reshape(-1), ev, plan.T, plan.F))
        _close(gx, co.c_chain_bsr3_bwd(b["rowptr"], b["col"], b["blk"], G.cpu().numpy(), ev, plan.T), rtol=1e-5)
        assert torch.equal(S[..., 0].reshape(-1, 1), xv) and torch.equal(S[..., 1:1 + plan.F].reshape(W.shape), W)
    assert n_shared >= 4, n_shared                       # the direct kernel really ran (both modes of >= 2 cases)


@pytest.mark.parametrize("graphs,n,k", [(3, 1500, 6), (2, 1100, 40), (1, 37, 5)])
def test_bsr3_step_staged_vs_oracle(monkeypatch, graphs, n, k):
    """Multi-launch vector chain through the LDS-staged step kernel (k_bsr3_step_staged, the default) and through the
    round-1 lane-group kernel (ESCAT_BSR3_STEP=lanes): forward and x.grad against the C oracle.  k = 40 gives trips of
    32 rows with more than 768 blocks: the kernel's straight-from-global branch."""
    from esc_gnn_amd import get_Batch_Wjxs, ScatterOperator, synthetic, plan_chain
    dev = _dev()
    b = synthetic.ellipsoid_batch(graphs, n, k, seed=21)
    t = lambda a: torch.from_numpy(a).to(dev)
    if k == 40:
        assert np.diff(b["rowptr"][::32]).max() > 768
    op = ScatterOperator.from_bsr3(t(b["rowptr"]), t(b["col"]), t(b["blk"]), b["N"])
    kw = dict(J=3)
    plan = plan_chain(**kw)
    ev = np.asarray(plan.events, dtype=np.int32).reshape(-1, 4)
    xv = torch.from_numpy(b["pos"]).reshape(-1, 1)
    gq = torch.randn(3 * b["N"], 1, plan.F, generator=torch.Generator().manual_seed(2))
    ref = co.c_chain_bsr3(b["rowptr"], b["col"], b["blk"], xv.numpy().reshape(-1), ev, plan.T, plan.F)
    refg = co.c_chain_bsr3_bwd(b["rowptr"], b["col"], b["blk"], gq.numpy(), ev, plan.T)
    monkeypatch.setenv("ESCAT_CHAIN_PATH", "multi")
    outs = []
    for step in (None, "lanes"):
        if step:
            monkeypatch.setenv("ESCAT_BSR3_STEP", step)
        xq = xv.to(dev).requires_grad_(True)
        W = get_Batch_Wjxs(xq, op, **kw)
        W.backward(gq.to(dev))
        for got, want in ((W, ref), (xq.grad, refg)):
            got = got.detach().cpu().numpy().reshape(want.shape)
            assert np.abs(got - want).max() <= 1e-5 * np.abs(want).max()       # north_star's bar
        outs.append(W.detach())
    assert (outs[0] - outs[1]).abs().max() <= 2e-6 * outs[1].abs().max()


def test_bsr3_single_product_staged():
    """escat_bsr3_spmv_f32 (one torch.sparse.mm of the vector track) runs the staged step kernel too."""
    from esc_gnn_amd import synthetic
    from esc_gnn_amd import _lib
    dev = _dev()
    b = synthetic.ellipsoid_batch(5, 333, 9, seed=4)
    t = lambda a: torch.from_numpy(a).to(dev)
    rp, ci, blk = t(b["rowptr"]), t(b["col"]), t(b["blk"])
    x = torch.randn(3 * b["N"], generator=torch.Generator().manual_seed(3))
    y = torch.empty(3 * b["N"], device=dev)
    rc = _lib.load().escat_bsr3_spmv_f32(rp.data_ptr(), ci.data_ptr(), blk.data_ptr(), b["N"], b["nnz"], x.to(dev).data_ptr(),
                                   y.data_ptr(), torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    ev = np.array([[1, 0, 1, -1]], dtype=np.int32)
    ref = co.c_chain_bsr3(b["rowptr"], b["col"], b["blk"], x.numpy(), ev, 1, 1).reshape(-1)
    assert np.abs(y.cpu().numpy() - ref).max() <= 1e-5 * np.abs(ref).max()


@pytest.mark.parametrize("graphs,n,k,C", [(3, 1500, 6, 3), (2, 1100, 40, 3), (2, 1300, 9, 1), (2, 1300, 9, 5), (1, 2000, 30, 8), (1, 37, 5, 2)])
def test_csr_step_staged_vs_oracle(monkeypatch, graphs, n, k, C):
    """Multi-launch scalar chain through the LDS-staged step kernel (k_csr_step_staged, the default) and through the
    round-1 lane-group kernel (ESCAT_CSR_STEP=lanes): forward and x.grad against the C oracle, every channel count's
    trip shape (64 rows for C <= 4, 32 beyond), and trips beyond the LDS capacity (k = 30, 40: straight from global)."""
    from esc_gnn_amd import get_Batch_Wjxs, ScatterOperator, synthetic, plan_chain
    dev = _dev()
    b = synthetic.ellipsoid_batch(graphs, n, k, seed=17)
    t = lambda a: torch.from_numpy(a).to(dev)
    rows_per_trip = 64 if C <= 4 else 32
    if k >= 30:
        assert np.diff(b["rowptr"][::rows_per_trip]).max() > rows_per_trip * 24
    op = ScatterOperator.from_csr(t(b["rowptr"]), t(b["col"]), t(b["pval"]), b["N"])
    kw = dict(J=3)
    plan = plan_chain(**kw)
    ev = np.asarray(plan.events, dtype=np.int32).reshape(-1, 4)
    gen = torch.Generator().manual_seed(8)
    x = torch.randn(b["N"], C, generator=gen)
    g = torch.randn(b["N"], C, plan.F, generator=gen)
    ref = co.c_chain_csr(b["rowptr"], b["col"], b["pval"], x.numpy(), ev, plan.T, plan.F)
    refg = co.c_chain_csr_bwd(b["rowptr"], b["col"], b["pval"], g.numpy(), ev, plan.T)
    monkeypatch.setenv("ESCAT_CHAIN_PATH", "multi")
    outs = []
    for step in (None, "lanes"):
        if step:
            monkeypatch.setenv("ESCAT_CSR_STEP", step)
        xg = x.to(dev).requires_grad_(True)
        W = get_Batch_Wjxs(xg, op, **kw)
        W.backward(g.to(dev))
        for got, want in ((W, ref), (xg.grad, refg)):
            got = got.detach().cpu().numpy().reshape(want.shape)
            assert np.abs(got - want).max() <= 1e-5 * np.abs(want).max()       # north_star's bar
        outs.append(W.detach())
    assert (outs[0] - outs[1]).abs().max() <= 2e-6 * outs[1].abs().max()


def test_csr_single_product_staged_strided():
    """escat_csr_spmm_f32 (one torch.sparse.mm(P, X)) through the staged kernel with leading dimensions larger than C."""
    from esc_gnn_amd import synthetic, _lib
    dev = _dev()
    b = synthetic.ellipsoid_batch(7, 200, 7, seed=6)
    t = lambda a: torch.from_numpy(a).to(dev)
    rp, ci, pv = t(b["rowptr"]), t(b["col"]), t(b["pval"])
    N = b["N"]
    X = torch.randn(N, 6, device=dev)
    Y = torch.full((N, 5), 7.0, device=dev)
    rc = _lib.load().escat_csr_spmm_f32(rp.data_ptr(), ci.data_ptr(), pv.data_ptr(), N, b["nnz"], X.data_ptr(), 6, Y.data_ptr(), 5, 4,
                                        torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    rows = np.repeat(np.arange(N), np.diff(b["rowptr"]))
    P = torch.sparse_coo_tensor(torch.from_numpy(np.stack([rows, b["col"]])).long(), torch.from_numpy(b["pval"]), (N, N))
    ref = torch.sparse.mm(P, X.cpu()[:, :4])
    assert (Y.cpu()[:, :4] - ref).abs().max() <= 1e-5 * ref.abs().max() and bool((Y[:, 4] == 7.0).all())


def test_lds_budget_edge_falls_back_to_the_multi_launch_path():
    """512-node graphs with 20 filters: the forward slab + vectors + staged input + row->position table need 164 864 B of
    LDS, 1 KiB more than a CU has -- the Python side must see that (its size formula mirrors escat_sell_lds_bytes) and
    take the multi-launch path instead of handing the library a launch it refuses; 19 filters still run register-resident."""
    from esc_gnn_amd import get_Batch_Wjxs, ScatterOperator, synthetic, plan_chain
    from esc_gnn_amd.pyg_utilities import chain_lds_bytes, LDS_BUDGET
    dev = _dev()
    b = synthetic.ellipsoid_batch(2, 512, 8, seed=12)
    t = lambda a: torch.from_numpy(a).to(dev)
    op = ScatterOperator.from_csr(t(b["rowptr"]), t(b["col"]), t(b["pval"]), b["N"])
    assert op.csr().max_part_rows == 512
    assert chain_lds_bytes(512, 3, 20, False) > LDS_BUDGET >= chain_lds_bytes(512, 3, 19, False)
    x = torch.randn(b["N"], 3, generator=torch.Generator().manual_seed(4))
    for nsc in (20, 19):
        kw = dict(scales_type="custom", diffusion_scales=torch.arange(nsc))
        plan = plan_chain(**kw)
        assert plan.F == nsc
        ev = np.asarray(plan.events, dtype=np.int32).reshape(-1, 4)
        W = get_Batch_Wjxs(x.to(dev), op, **kw)
        ref = co.c_chain_csr(b["rowptr"], b["col"], b["pval"], x.numpy(), ev, plan.T, plan.F)
        _close(W, ref)
