"""
The allocation-free entry point escat_chain_csr_static_f32 (include/escat.h; no counterpart in the reference: for
callers that cannot allocate -- HIP graph capture, serving loops).  Called straight through the C ABI on graphs that do
not fit LDS (the multi-launch path, whose ping-pong vectors are the scratch), twice in a row, the second call on a
smaller input, and compared bit for bit with escat_chain_csr_f32 given a caller workspace (same kernels, same order of
operations: integer-exact agreement is the bar), plus the oracle as an absolute check.
"""
import ctypes
import numpy as np
import pytest
import torch

from oracle import chain_oracle as co

pytestmark = pytest.mark.gpu


def _call(lib, fn_static, rowptr, col, val, x, ev_arr, n_ev, T, F, ws=None):
    n, C = x.shape
    out = torch.full((n, C, F), float("nan"), device=x.device)
    vp = lambda t: ctypes.c_void_p(t.data_ptr())
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    common = (vp(rowptr), vp(col), vp(val), n, int(col.numel()), vp(x), C, T, ctypes.cast(ev_arr, ctypes.c_void_p), n_ev, F,
              vp(out), ctypes.c_void_p(0), 0, 0, ctypes.c_void_p(0))
    if fn_static:
        rc = lib.escat_chain_csr_static_f32(*common, st)
    else:
        rc = lib.escat_chain_csr_f32(*common, vp(ws), ws.numel() * 4, st)
    assert rc == 0, lib.escat_last_error()
    torch.cuda.synchronize()
    return out


def test_static_scratch_entry_point_matches_workspace_entry_point(oracle_lib):
    from esc_gnn_amd import _lib, synthetic, plan_chain
    from esc_gnn_amd.pyg_utilities import _events_array
    lib = _lib.load()
    dev = torch.device("cuda:0")
    plan = plan_chain(scales_type="dyadic", J=3)
    ev_arr = _events_array(plan.events)
    ev_np = np.asarray(plan.events, dtype=np.int32).reshape(-1, 4)
    t = lambda a: torch.from_numpy(a).to(dev)
    for n_nodes, C, seed in ((6000, 3, 1), (2500, 5, 2)):        # two calls in a row, the second one smaller
        c = synthetic.single_cloud(n_nodes, 6, seed=seed, workers=1)
        rowptr, col, val = t(c["rowptr"]), t(c["col"]), t(c["pval"])
        x = torch.randn(n_nodes, C, generator=torch.Generator().manual_seed(seed)).to(dev)
        ws = torch.empty(lib.escat_chain_workspace_bytes(n_nodes, C) // 4, dtype=torch.float32, device=dev)
        ref = _call(lib, False, rowptr, col, val, x, ev_arr, len(plan.events), plan.T, plan.F, ws)
        got = _call(lib, True, rowptr, col, val, x, ev_arr, len(plan.events), plan.T, plan.F)
        assert torch.equal(got, ref)                               # same kernels, scratch in place of the workspace
        orc = co.c_chain_csr(c["rowptr"], c["col"], c["pval"], x.cpu().numpy(), ev_np, plan.T, plan.F)
        for f in range(plan.F):
            assert np.abs(got.cpu().numpy()[..., f] - orc[..., f]).max() <= 1e-5 * np.abs(orc[..., f]).max()


def test_static_entry_on_the_register_path_is_not_refused_for_its_size_and_allocates_nothing(oracle_lib):
    """with part_ptr and a pack the chain stays in LDS and registers and uses no scratch, so a batch whose
    multi-launch workspace would exceed the 8 MiB array (N = 600 k rows x C = 3) is served, bit-equal to the workspace
    entry point; torch's allocator sees no allocation across the call (the library has no other allocator: grep
    hipMalloc csrc/)."""
    from esc_gnn_amd import _lib, synthetic, plan_chain, ScatterOperator
    from esc_gnn_amd.pyg_utilities import _events_array
    lib = _lib.load()
    dev = torch.device("cuda:0")
    plan = plan_chain(scales_type="dyadic", J=3)
    ev_arr = _events_array(plan.events)
    ev_np = np.asarray(plan.events, dtype=np.int32).reshape(-1, 4)
    b = synthetic.ellipsoid_batch(1200, 500, 8, seed=3)
    n, C = b["N"], 3
    assert lib.escat_chain_workspace_bytes(n, C) > 8 << 20
    t = lambda a: torch.from_numpy(a).to(dev)
    op = ScatterOperator.from_csr(t(b["rowptr"]), t(b["col"]), t(b["pval"]), n)
    form = op.csr()
    sell = form.sell()
    assert sell is not None and form.n_parts > 0
    half = getattr(sell, "half_val", None) is not None
    ss = _lib.EscatSell(sell.val.data_ptr(), sell.lc.data_ptr(), sell.lane_meta.data_ptr(), sell.dmax, sell.n_slices,
                        sell.slices_per_part, sell.val_planes, 0, 0, 0, 0, 0, 0, 0, 0, sell.layout, sell.pos_of_row.data_ptr(),
                        sell.half_val.data_ptr() if half else 0, sell.half_aux.data_ptr() if half else 0)
    sref = ctypes.cast(ctypes.pointer(ss), ctypes.c_void_p)
    x = torch.randn(n, C, generator=torch.Generator().manual_seed(1)).to(dev)
    vp = lambda a: ctypes.c_void_p(a.data_ptr())
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    outs = []
    for static in (False, True):
        out = torch.full((n, C, plan.F), float("nan"), device=dev)
        common = (vp(form.rowptr), vp(form.col), vp(form.val), n, form.nnz, vp(x), C, plan.T,
                  ctypes.cast(ev_arr, ctypes.c_void_p), len(plan.events), plan.F, vp(out),
                  vp(form.part_ptr), form.n_parts, form.max_part_rows, sref)
        torch.cuda.synchronize()
        before = (torch.cuda.memory_allocated(dev), torch.cuda.memory_reserved(dev))
        if static:
            rc = lib.escat_chain_csr_static_f32(*common, st)
        else:
            rc = lib.escat_chain_csr_f32(*common, ctypes.c_void_p(0), 0, st)
        assert rc == 0, lib.escat_last_error()
        torch.cuda.synchronize()
        assert (torch.cuda.memory_allocated(dev), torch.cuda.memory_reserved(dev)) == before
        outs.append(out)
    assert torch.equal(outs[0], outs[1])
    sub = slice(0, 20 * 500)                                       # the oracle on the first 20 graphs (block-diagonal)
    rp = b["rowptr"][:20 * 500 + 1]
    orc = co.c_chain_csr(rp, b["col"][:rp[-1]], b["pval"][:rp[-1]], x[sub].cpu().numpy(), ev_np, plan.T, plan.F)
    got = outs[1][sub].cpu().numpy()
    for f in range(plan.F):
        assert np.abs(got[..., f] - orc[..., f]).max() <= 1e-5 * np.abs(orc[..., f]).max()


def test_static_scratch_rejects_oversize():
    from esc_gnn_amd import _lib
    from esc_gnn_amd.pyg_utilities import _events_array
    from esc_gnn_amd import plan_chain
    lib = _lib.load()
    dev = torch.device("cuda:0")
    n, C = 400_000, 8                                             # 2 * n * C * 4 * 4/3 B > 8 MiB
    plan = plan_chain(scales_type="dyadic", J=1)
    ev_arr = _events_array(plan.events)
    rowptr = torch.arange(n + 1, dtype=torch.int32, device=dev)
    col = torch.arange(n, dtype=torch.int32, device=dev)
    val = torch.ones(n, device=dev)
    x = torch.ones(n, C, device=dev)
    out = torch.empty(n, C, plan.F, device=dev)
    vp = lambda t: ctypes.c_void_p(t.data_ptr())
    rc = lib.escat_chain_csr_static_f32(vp(rowptr), vp(col), vp(val), n, n, vp(x), C, plan.T,
                                        ctypes.cast(ev_arr, ctypes.c_void_p), len(plan.events), plan.F, vp(out),
                                        ctypes.c_void_p(0), 0, 0, ctypes.c_void_p(0), ctypes.c_void_p(0))
    assert rc == -2 and b"static" in lib.escat_last_error()
