"""
CPU tests of the product's host logic (no GPU, no kernel launches):
  * the chain planner (esc_gnn_amd/scales.py) drives the C ORACLE to the golden
    numbers of every mode -> the event tables the kernels receive are right;
  * operator ingest (esc_gnn_amd/operators.py) gives the bit-exact CSR / BSR-3
    index structure of the oracle's canonicalisation, correct transposes and
    closed row ranges;
  * libescat.so loads and exports every symbol include/escat.h declares.
"""
import os
import re
import numpy as np
import pytest
import torch

from oracle import chain_oracle as co
from esc_gnn_amd.scales import plan_chain
from esc_gnn_amd.operators import ScatterOperator, _closed_row_ranges
from esc_gnn_amd import synthetic

from tests.test_oracle import MODES

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _ev(plan):
    return np.asarray(plan.events, dtype=np.int32).reshape(-1, 4)


@pytest.mark.parametrize("mode", list(MODES))
def test_plan_drives_oracle_to_goldens(sg, oracle_lib, mode):
    N = int(sg["N"])
    rp, ci, v = co.coo_to_csr(sg["P_indices"], sg["P_values"], N)
    plan = plan_chain(**MODES[mode])
    for C in (1, 3, 5):
        ref = sg[f"P_{mode}_C{C}_W"]
        W = co.c_chain_csr(rp, ci, v, sg[f"x_C{C}"], _ev(plan), plan.T, plan.F)
        assert W.shape == ref.shape
        assert np.abs(W - ref).max() <= 1e-5 * np.abs(ref).max()
        gx = co.c_chain_csr_bwd(rp, ci, v, sg[f"P_{mode}_C{C}_gup"], _ev(plan), plan.T)
        assert np.abs(gx - sg[f"P_{mode}_C{C}_gx"]).max() <= 1e-5 * np.abs(sg[f"P_{mode}_C{C}_gx"]).max()


def test_plan_perchannel(sg, oracle_lib):
    N = int(sg["N"])
    rp, ci, v = co.coo_to_csr(sg["P_indices"], sg["P_values"], N)
    for lp in (True, False):
        plan = plan_chain(scales_type="custom", diffusion_scales=torch.from_numpy(sg["perchan_scales"]), J=3,
                          include_lowpass=lp)
        assert plan.per_channel and plan.n_channels == 3
        W = co.c_chain_csr(rp, ci, v, sg["x_C3"], _ev(plan), plan.T, plan.F)
        ref = sg[f"P_perchan_lp{int(lp)}_C3_W"]
        assert W.shape == ref.shape and np.abs(W - ref).max() <= 1e-5 * np.abs(ref).max()


def test_plan_matches_oracle_tables():
    for kw in MODES.values():
        plan = plan_chain(**kw)
        ev, T, F = co.events_for_mode(3, **kw)
        assert (plan.F, sorted(plan.events)) == (F, sorted(map(tuple, ev.tolist())))
        assert plan.T <= T


def test_plan_errors():
    with pytest.raises(NotImplementedError):
        plan_chain(scales_type="bogus")
    with pytest.raises(NotImplementedError):
        plan_chain(scales_type="custom", diffusion_scales="dyadic")
    with pytest.raises(IndexError):                 # too few kept powers for the band count (reference: list index)
        plan_chain(scales_type="custom", diffusion_scales=torch.tensor([0, 0, 1]))
    p = plan_chain(scales_type="custom", diffusion_scales=torch.tensor([0]))   # lowpass = x itself
    assert (p.T, p.F, p.events) == (0, 1, ((0, 0, 1, -1),))
    p = plan_chain(scales_type="custom", diffusion_scales=torch.tensor([1, 2, 4]))
    assert p.F == 3 and p.T == 4                     # x-P, P-P^2, lowpass P^4: the P^2-P^4 band is dropped
    assert (2, 1, -1, -1) in p.events and (4, 2, 1, -1) in p.events and (4, 1, -1, -1) not in p.events


def test_ingest_index_structure_bit_exact(sg):
    N = int(sg["N"])
    P = torch.sparse_coo_tensor(torch.from_numpy(sg["P_indices"]), torch.from_numpy(sg["P_values"]), (N, N))
    Q = torch.sparse_coo_tensor(torch.from_numpy(sg["Q_indices"]), torch.from_numpy(sg["Q_values"]), (3 * N, 3 * N))
    assert not P.is_coalesced()
    opP = ScatterOperator.from_torch(P, allow_cpu=True)
    rp, ci, v = co.coo_to_csr(sg["P_indices"], sg["P_values"], N)
    c = opP.csr()
    assert c.rowptr.dtype == torch.int32 and c.col.dtype == torch.int32
    assert np.array_equal(c.rowptr.numpy(), rp) and np.array_equal(c.col.numpy(), ci) and np.array_equal(c.val.numpy(), v)
    assert not opP.has_bsr3                                     # P is not made of dense 3x3 blocks
    opQ = ScatterOperator.from_torch(Q, allow_cpu=True)
    brp, bci, blk = co.coo_to_bsr3(sg["Q_indices"], sg["Q_values"], 3 * N)
    assert opQ.has_bsr3 and opQ.bsr_fill == 1.0
    b = opQ.bsr3()
    assert np.array_equal(b.rowptr.numpy(), brp) and np.array_equal(b.col.numpy(), bci)
    assert np.array_equal(b.val.numpy(), blk)
    assert np.array_equal(b.rowptr.numpy(), rp) and np.array_equal(b.col.numpy(), ci)    # Q sits on P's pattern
    # scalar CSR of Q agrees with the oracle's too
    qrp, qci, qv = co.coo_to_csr(sg["Q_indices"], sg["Q_values"], 3 * N)
    cq = opQ.csr()
    assert np.array_equal(cq.rowptr.numpy(), qrp) and np.array_equal(cq.col.numpy(), qci) and np.array_equal(cq.val.numpy(), qv)
    # closed row ranges = the graphs of the batch (32 + 64 + 48 nodes; merged up to 256 rows -> one part)
    assert c.n_parts >= 1 and int(c.part_ptr[-1]) == N
    pp, npart, mx = _closed_row_ranges(c.rowptr, c.col, N, target=1)
    assert np.array_equal(pp.numpy(), sg["graph_ptr"].astype(np.int32)) and mx == 64
    # CSR input (what the probe in SURVEY §8c used) canonicalises to the same thing
    opc = ScatterOperator.from_torch(P.coalesce().to_sparse_csr(), allow_cpu=True)
    assert np.array_equal(opc.csr().col.numpy(), ci) and np.array_equal(opc.csr().val.numpy(), v)


def test_ingest_transposes(sg):
    N = int(sg["N"])
    P = torch.sparse_coo_tensor(torch.from_numpy(sg["P_indices"]), torch.from_numpy(sg["P_values"]), (N, N))
    Q = torch.sparse_coo_tensor(torch.from_numpy(sg["Q_indices"]), torch.from_numpy(sg["Q_values"]), (3 * N, 3 * N))
    opP = ScatterOperator.from_torch(P, allow_cpu=True)
    t = opP.csr_t()
    dense_t = torch.sparse_csr_tensor(t.rowptr.long(), t.col.long(), t.val, (N, N)).to_dense()
    assert torch.equal(dense_t, P.to_dense().T)
    opQ = ScatterOperator.from_torch(Q, allow_cpu=True)
    bt = opQ.bsr3_t()
    Qt = torch.zeros(3 * N, 3 * N)
    rp = bt.rowptr.numpy()
    for i in range(N):
        for e in range(rp[i], rp[i + 1]):
            j = int(bt.col[e])
            Qt[3 * i:3 * i + 3, 3 * j:3 * j + 3] = bt.val[e]
    assert torch.equal(Qt, Q.to_dense().T)
    # BSR-only operator can be expanded to scalar CSR
    b = opQ.bsr3()
    op2 = ScatterOperator.from_bsr3(b.rowptr, b.col, b.val, N)
    c2, cq = op2.csr(), opQ.csr()
    assert torch.equal(c2.rowptr, cq.rowptr) and torch.equal(c2.col, cq.col) and torch.equal(c2.val, cq.val)


def test_row_ranges_edge_cases():
    # empty rows, a dense coupling block, and a long chain that can never be cut
    rowptr = torch.tensor([0, 1, 1, 3, 5, 6], dtype=torch.int32)      # row 1 empty
    col = torch.tensor([0, 2, 3, 2, 3, 4], dtype=torch.int32)
    pp, n, mx = _closed_row_ranges(rowptr, col, 5, target=1)
    assert pp.tolist() == [0, 1, 2, 4, 5] and mx == 2
    pp, n, mx = _closed_row_ranges(rowptr, col, 5, target=3)
    assert pp.tolist()[0] == 0 and pp.tolist()[-1] == 5 and mx <= 3
    n = 50
    rows = torch.arange(n).repeat_interleave(2)[1:-1]
    cols = torch.stack([torch.arange(n - 1), torch.arange(1, n)], 1).reshape(-1)
    A = torch.sparse_coo_tensor(torch.stack([rows, cols]), torch.ones(rows.numel()), (n, n))
    op = ScatterOperator.from_torch(A, allow_cpu=True)
    assert op.csr().n_parts == 1 and op.csr().max_part_rows == n
    with pytest.raises(ValueError):
        ScatterOperator.from_csr(torch.tensor([0, 2, 1]), torch.tensor([0, 1]), torch.ones(2), 2)
    with pytest.raises(ValueError):
        ScatterOperator.from_csr(torch.tensor([0, 1, 2]), torch.tensor([0, 5]), torch.ones(2), 2)
    with pytest.raises(RuntimeError):
        ScatterOperator.from_torch(A)          # CPU operator refused on the product path


def test_synthetic_batch_matches_reference_P(bg):
    """synthetic.lazy_walk_values == get_P of the reference on the golden graphs (bit-exact)."""
    for ci, c in enumerate(bg["cases"]):
        n, k = int(str(c).split(",")[0]), int(str(c).split(",")[1])
        rp, col = synthetic.knn_sym_csr(bg[f"g{ci}_pos"], k)
        Pi = bg[f"g{ci}_P_indices"]
        assert np.array_equal(col, Pi[1]) and rp[-1] == Pi.shape[1]
        assert np.array_equal(synthetic.lazy_walk_values(rp, col), bg[f"g{ci}_P_values_f32"])
    b = synthetic.ellipsoid_batch(3, 40, 5)
    R = b["blk"] / b["pval"][:, None, None]
    assert np.abs(np.einsum("eab,ecb->eac", R, R) - np.eye(3)).max() < 1e-5


def test_library_exports_every_declared_symbol():
    from esc_gnn_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "escat.h")).read()
    declared = set(re.findall(r"\b(escat_[A-Za-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    lib = _lib.load()
    for name in declared:
        assert hasattr(lib, name)
    assert lib.escat_abi_version() == 5
    assert lib.escat_chain_workspace_bytes(1000, 3) >= 2 * 1000 * 3 * 4


def test_product_path_refuses_cpu(sg):
    from esc_gnn_amd import get_Batch_Wjxs
    N = int(sg["N"])
    P = torch.sparse_coo_tensor(torch.from_numpy(sg["P_indices"]), torch.from_numpy(sg["P_values"]), (N, N))
    with pytest.raises(RuntimeError):
        get_Batch_Wjxs(torch.from_numpy(sg["x_C3"]), P, J=2)
    with pytest.raises(NotImplementedError):
        get_Batch_Wjxs(torch.from_numpy(sg["x_C3"]), P, scales_type="nope")
    with pytest.raises(NotImplementedError):
        get_Batch_Wjxs(torch.from_numpy(sg["x_C3"]), P, J=2, rescale_filtrations=True)


def test_operator_batch_view_index_logic():
    """OperatorBatch (host logic, CPU tensors): part lists, first rows and the row index of a drawn batch."""
    import torch
    from esc_gnn_amd import ScatterOperator
    # three graphs of 2, 3, 1 nodes: block-diagonal identity-like CSR
    n = 6
    rowptr = torch.arange(n + 1, dtype=torch.int32); col = torch.arange(n, dtype=torch.int32); val = torch.ones(n)
    gp = torch.tensor([0, 2, 5, 6])
    op = ScatterOperator.from_csr(rowptr, col, val, n, parts=ScatterOperator.parts_from_graph_ptr(gp))
    v = op.select([2, 0, 0, 1])
    assert v.n == 1 + 2 + 2 + 3
    assert v.row_index().tolist() == [5, 0, 1, 0, 1, 2, 3, 4]
    pl, r0 = v.lists_for(op.csr())
    assert pl.tolist() == [2, 0, 0, 1] and r0.tolist() == [0, 1, 3, 5] and pl.dtype == torch.int32
    # the transposed form keeps the forward form's parts
    assert op.csr_t().n_parts == 3 and v.lists_for(op.csr_t())[1].tolist() == [0, 1, 3, 5]
    import pytest
    with pytest.raises(IndexError):
        op.select([3])
    blk = torch.eye(3).repeat(n, 1, 1)
    q = ScatterOperator.from_bsr3(rowptr, col, blk, n, parts=ScatterOperator.parts_from_graph_ptr(gp)).select([1])
    assert q.n == 9 and q.row_index().tolist() == list(range(6, 15))


def test_sort_free_ingest_of_reference_layouts():
    """from_torch on the layouts the reference produces (row-major P, block-emission-order Q, both flagged
    un-coalesced) skips the sort and still gives the canonical structure of the coalesce path; anything else falls back."""
    import numpy as np
    import torch
    from esc_gnn_amd import synthetic, ScatterOperator
    b = synthetic.ellipsoid_batch(6, 50, 5, seed=3)
    N = b["N"]; rows = np.repeat(np.arange(N), np.diff(b["rowptr"])); cols = b["col"].astype(np.int64)
    a = np.arange(3)
    qr = ((3 * rows)[:, None, None] + a[None, :, None] + 0 * a[None, None, :]).reshape(-1)
    qc = ((3 * cols)[:, None, None] + 0 * a[None, :, None] + a[None, None, :]).reshape(-1)
    Q = torch.sparse_coo_tensor(torch.from_numpy(np.stack([qr, qc])), torch.from_numpy(b["blk"].reshape(-1)), (3 * N, 3 * N))
    P = torch.sparse_coo_tensor(torch.from_numpy(np.stack([rows, cols])), torch.from_numpy(b["pval"]), (N, N))
    assert not Q.is_coalesced() and not P.is_coalesced()
    assert ScatterOperator._from_emission_order(Q, 3 * N) is not None and ScatterOperator._from_emission_order(P, N) is not None
    fq, fp = ScatterOperator.from_torch(Q, allow_cpu=True), ScatterOperator.from_torch(P, allow_cpu=True)
    sq, sp = ScatterOperator.from_torch(Q.coalesce(), allow_cpu=True), ScatterOperator.from_torch(P.coalesce(), allow_cpu=True)
    for x, y in ((fq.bsr3(), sq.bsr3()), (fp.csr(), sp.csr()), (fq.csr(), sq.csr())):
        assert torch.equal(x.rowptr, y.rowptr) and torch.equal(x.col, y.col) and torch.equal(x.val, y.val)
        assert torch.equal(x.part_ptr, y.part_ptr)
    perm = torch.randperm(Q._nnz(), generator=torch.Generator().manual_seed(0))
    Qs = torch.sparse_coo_tensor(Q._indices()[:, perm], Q._values()[perm], Q.shape)
    assert ScatterOperator._from_emission_order(Qs, 3 * N) is None
    assert torch.equal(ScatterOperator.from_torch(Qs, allow_cpu=True).bsr3().val, sq.bsr3().val)
    # duplicates must not take the fast path (coalesce sums them)
    Pd = torch.sparse_coo_tensor(torch.tensor([[0, 0, 1], [0, 0, 1]]), torch.tensor([1.0, 2.0, 3.0]), (2, 2))
    assert ScatterOperator._from_emission_order(Pd, 2) is None
    assert ScatterOperator.from_torch(Pd, allow_cpu=True).csr().val.tolist() == [3.0, 3.0]


def test_scattering_cache_keys():
    """per-batch memo of scatter results: identity + version + settings are the key."""
    from esc_gnn_amd.scatter import ScatteringCache
    c = ScatteringCache()
    x, op = torch.randn(6, 2), object()
    dkw = dict(scales_type="dyadic", J=4, include_lowpass=True)
    assert c.lookup("scalar", x, op, dkw, False) is None
    W = torch.zeros(6, 2, 22)
    c.store("scalar", x, op, dkw, False, W)
    assert c.lookup("scalar", x, op, dkw, False) is W
    assert c.lookup("vector", x, op, dkw, False) is None                  # one entry per track
    assert c.lookup("scalar", x, op, dict(dkw, J=3), False) is None       # other settings
    assert c.lookup("scalar", x, op, dkw, True) is None
    assert c.lookup("scalar", x.clone(), op, dkw, False) is None          # another tensor
    assert c.lookup("scalar", x, object(), dkw, False) is None            # another operator
    x.add_(1.0)                                                           # in-place edit bumps the version counter
    assert c.lookup("scalar", x, op, dkw, False) is None
    ds = torch.tensor([0, 1, 2, 4])
    c.store("scalar", x, op, dict(scales_type="custom", diffusion_scales=ds), False, W)
    assert c.lookup("scalar", x, op, dict(scales_type="custom", diffusion_scales=ds.clone()), False) is W
    assert c.lookup("scalar", x, op, dict(scales_type="custom", diffusion_scales=torch.tensor([0, 1, 2, 5])), False) is None
    assert c.hits == 2 and c.misses == 8
    c.clear()
    assert c.lookup("scalar", x, op, dict(scales_type="custom", diffusion_scales=ds), False) is None


def test_signed_pack_of_per_channel_values(monkeypatch):
    """|value| + three sign bits in the column word (escat_sell.val_planes = 4): host-side packing logic."""
    from esc_gnn_amd.operators import _Csr
    rowptr = torch.tensor([0, 2, 3, 5], dtype=torch.int32)
    col = torch.tensor([0, 2, 1, 0, 2], dtype=torch.int32)
    p = torch.tensor([0.5, 0.25, 1.0, 0.125, 0.75])
    sgn = torch.tensor([[1, 1, 1], [-1, 1, -1], [1, -1, 1], [-1, -1, -1], [1, 1, -1]], dtype=torch.float32)
    parts = (torch.tensor([0, 3], dtype=torch.int32), 1, 3)
    form = _Csr(rowptr, col, (p[:, None] * sgn).contiguous(), 3, parts)
    colp, mag, bits = form.signed_pack()
    assert torch.equal(colp & 0x1fffffff, col) and torch.equal(mag, p)
    want_bits = ((sgn[:, 0] < 0).int() | ((sgn[:, 1] < 0).int() << 1) | ((sgn[:, 2] < 0).int() << 2))
    assert torch.equal(bits, want_bits) and torch.equal((colp >> 29) & 7, want_bits)
    # decoding the way the kernels do: value_c = |v| with the sign bit of channel c
    dec = torch.stack([torch.where(((colp >> (29 + c)) & 1) == 1, -mag, mag) for c in range(3)], dim=1)
    assert torch.equal(dec, form.val)
    # unrelated per-channel values have no such form; nor has anything with ESCAT_FACTORED_FORMAT=planes
    assert _Csr(rowptr, col, torch.rand(5, 3) + 0.1, 3, parts).signed_pack() is None
    monkeypatch.setenv("ESCAT_FACTORED_FORMAT", "planes")
    assert _Csr(rowptr, col, (p[:, None] * sgn).contiguous(), 3, parts).signed_pack() is None
    # a zero value carries no sign
    z = _Csr(rowptr, col, torch.zeros(5, 3) * -1.0, 3, parts)
    monkeypatch.delenv("ESCAT_FACTORED_FORMAT")
    assert int(z.signed_pack()[2].sum()) == 0


def test_ellipsoid_sampler_bit_identical_to_reference_golden():
    """synthetic.ellipsoid_clouds restates EllipsoidPointCloudGenerator.generate_ellipsoid_points_with_params
    (ellipsoid_data_classes.py:82-134) -- same draws, same arithmetic order; golden generated from the reference class
    (tests/golden/make_round2_goldens.py), seed 457892, 4 x 16 points."""
    import os
    import numpy as np
    from esc_gnn_amd import synthetic
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "round2_goldens.npz"))
    assert np.array_equal(synthetic.ellipsoid_clouds(4, 16, seed=457892, dtype=np.float64), g["ell_clouds"])
    assert np.array_equal(synthetic.ellipsoid_clouds(3, 16, seed=457892, dtype=np.float64), g["ell_clouds_own_rng"])
    assert np.array_equal(synthetic.ellipsoid_clouds(4, 16, seed=457892), g["ell_clouds"].astype(np.float32))


def test_escgnn_data_surface():
    """the object process_pyg_data returns (models/escgnn_data_classes.py:13-39): operator keys collate block-diagonally"""
    import torch
    from esc_gnn_amd.data_classes import ESCGNNData
    d = ESCGNNData(pos=torch.zeros(4, 3), edge_index=torch.zeros(2, 5, dtype=torch.long), P=torch.eye(4).to_sparse())
    assert d.__cat_dim__("P", d.P) == (0, 1) and d.__cat_dim__("Q", None) == (0, 1)
    assert d.__cat_dim__("edge_index", d.edge_index) == 1 and d.__cat_dim__("pos", d.pos) == 0
    assert set(["pos", "edge_index", "P"]) <= set(d.keys()) and d["pos"].shape == (4, 3)
    d2 = ESCGNNData(operator_keys=("P", "Q", "L"), x=torch.ones(2))
    assert d2.__cat_dim__("L", None) == (0, 1)
    assert d.to("cpu") is not None


def test_operator_cache_follows_in_place_edits():
    """as_operator caches the canonical operator on the tensor; an in-place edit of its values rebuilds it."""
    import torch
    from esc_gnn_amd import operators
    A = torch.sparse_coo_tensor(torch.tensor([[0, 0, 1, 1], [0, 1, 0, 1]]), torch.tensor([.5, .5, .5, .5]), (2, 2)).coalesce()
    calls = []
    orig = operators.ScatterOperator.from_torch
    operators.ScatterOperator.from_torch = staticmethod(lambda t, *a, **k: calls.append(1) or orig(t, allow_cpu=True))
    try:
        op1 = operators.as_operator(A)
        assert operators.as_operator(A) is op1 and len(calls) == 1
        A._values().mul_(2.0)
        op2 = operators.as_operator(A)
        assert op2 is not op1 and len(calls) == 2
        assert float(op2.csr().val.sum()) == 4.0
    finally:
        operators.ScatterOperator.from_torch = orig


def test_frames_of_a_relabelled_cloud_recovered_from_its_blocks_on_the_host():
    """esc_gnn_amd/recover_large.py is batched torch: the same code runs here on CPU tensors.  One cloud, blocks built from
    random orthonormal frames, rows relabelled breadth-first (operators.locality_order): the recovered frames reproduce
    every block as p_ij G_i diag(+-1) G_j^T to 1e-6 of max|Q| (they differ from the true frames by a column permutation
    and signs, which is the operator's own gauge freedom)."""
    import torch
    from esc_gnn_amd import synthetic
    from esc_gnn_amd.operators import locality_order
    from esc_gnn_amd.recover_large import recover_frames_large
    c = synthetic.single_cloud(1500, 8, workers=1)
    n = c["N"]
    rowptr, col, blk = torch.from_numpy(c["rowptr"]), torch.from_numpy(c["col"]), torch.from_numpy(c["blk"])
    perm = locality_order(rowptr, col, n)
    inv = torch.empty_like(perm); inv[perm] = torch.arange(n)
    rp = rowptr.long()
    rows_old = torch.repeat_interleave(torch.arange(n), rp[1:] - rp[:-1])
    r2, c2 = inv[rows_old], inv[col.long()]
    order = torch.argsort(r2 * n + c2)
    rows_new, col_new, blk_new = r2[order], c2[order], blk[order]
    rowptr_new = torch.zeros(n + 1, dtype=torch.int64); rowptr_new[1:] = torch.cumsum(torch.bincount(rows_new, minlength=n), 0)
    G = recover_frames_large(rowptr_new.to(torch.int32), col_new.to(torch.int32), blk_new)
    assert G is not None and G.shape == (n, 3, 3)
    assert (torch.einsum("nac,nbc->nab", G, G) - torch.eye(3)).abs().max() < 1e-6                     # orthonormal
    M = G.double()[rows_new].transpose(1, 2) @ blk_new.double() @ G.double()[col_new]
    d = torch.diagonal(M, dim1=1, dim2=2)
    p = blk_new.double()[:, 0, :].norm(dim=1)
    off = (M - torch.diag_embed(torch.where(d >= 0, p[:, None], -p[:, None]))).abs().max()
    assert float(off / blk_new.abs().max()) <= 1e-6
    # not such an operator (every block rotated at random): frames come back, but they do not reproduce the blocks
    g = torch.Generator().manual_seed(0)
    Rn = torch.linalg.qr(torch.randn(blk_new.shape[0], 3, 3, generator=g))[0]
    Gb = recover_frames_large(rowptr_new.to(torch.int32), col_new.to(torch.int32), (blk_new @ Rn).contiguous())
    if Gb is not None:
        Mb = Gb.double()[rows_new].transpose(1, 2) @ (blk_new @ Rn).double() @ Gb.double()[col_new]
        db = torch.diagonal(Mb, dim1=1, dim2=2)
        assert float((Mb - torch.diag_embed(db)).abs().max() / blk_new.abs().max()) > 1e-3
