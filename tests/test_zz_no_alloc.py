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
