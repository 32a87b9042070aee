"""
oracle/chain_oracle.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

CPU restatement of the reference's wavelet filter bank.  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg import this module;
esc_gnn_amd never does.

Two restatements of /root/reference/pyg_utilities.py:283-560 (get_Batch_Wjxs):

  ref_get_Batch_Wjxs   the same library call the reference makes
                       (torch.sparse.mm on the COO operand, fp32), the same
                       save / subtract / stack sequence; bit-identical to the
                       goldens on CPU.  This is what bench.py times as the
                       cpu_baseline ("port").
  c_chain_*            plain-C CSR/BSR walk (oracle/escat_oracle.c) driven by
                       an event table; used for sizes where the COO product is
                       too slow, for the fp64 twin and for the backward pass.

plus ref_scatter (models/escgnn_modular.py:309-362) and the canonical
COO -> CSR / BSR-3 conversion the parity tests compare index structure against.

Pinned by tests/test_oracle.py against tests/golden/scatter_goldens.npz.
"""
import ctypes
import os
import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libescat_oracle.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} missing: run `make -C oracle` (or __graft_entry__.build())")
        _LIB = ctypes.CDLL(path)
    return _LIB


# ------------------------------------------------------------------ the modes
def powers_for_mode(scales_type="dyadic", diffusion_scales=None, J=5, include_lowpass=True, n_channels=None):
    """
    Independent restatement of which powers P^t x the reference keeps and how it
    differences them.  Returns (T, bands, lowpass_power, perchan) where bands is
    a list of (plus_power, minus_power, chan) and lowpass_power is None or t.

    pyg_utilities.py:448-473  shared powers (custom 1-D / dyadic):
       Ptxs = [x] + [P^j x for j = 1..powers[-1] if j in powers (once per
       occurrence when diffusion_scales is given)];
       bands j = 1..range_upper_lim-1 : Ptxs[j-1] - Ptxs[j]; lowpass Ptxs[-1].
    pyg_utilities.py:392-444  2-D per-channel: all t = 0..2^J kept; channel c
       band i = P^{t_is[i-1]} - P^{t_is[i]}; lowpass P^{2^J}.
    pyg_utilities.py:476-479  scales_type None & no scales: P^1 only.
    """
    ds = diffusion_scales
    if isinstance(ds, str) and ds == "dyadic":
        ds = None
        is_str = True
    else:
        is_str = False
    if ds is not None:
        ds = np.asarray(ds.cpu() if torch.is_tensor(ds) else ds, dtype=np.int64)
    if ds is not None and ds.ndim == 2:
        T = int(2 ** J)
        bands = []
        for c, t_is in enumerate(ds):
            for i in range(1, len(t_is)):
                bands.append((int(t_is[i - 1]), int(t_is[i]), c))
        return T, bands, (T if include_lowpass else None), True
    if ds is not None and ds.ndim == 1:
        powers, upper = ds, int(ds.shape[0])
    elif scales_type == "dyadic":
        powers, upper = 2 ** np.arange(J + 1, dtype=np.int64), J + 2
    elif diffusion_scales is None and scales_type is None and not is_str:
        return 1, [], 1, False   # P^1 x, unsqueezed: one "lowpass-like" slot
    else:
        raise NotImplementedError(f"No method implemented for scales_type={scales_type}")
    T = int(powers[-1])
    kept = [0]
    for j in range(1, T + 1):
        kept += [j] * int((powers == j).sum())
    if upper - 1 > len(kept) - 1:
        raise IndexError("list index out of range")
    bands = [(kept[j - 1], kept[j], -1) for j in range(1, upper)]
    return T, bands, (kept[-1] if include_lowpass else None), False


def events_for_mode(C, **kw):
    """event table (n_ev, 4) int32 {step, slot, sign, chan}, T, F for the C oracle."""
    T, bands, lp, perchan = powers_for_mode(**kw)
    ev = []
    if perchan:
        per_c = {}
        for (p, m, c) in bands:
            f = per_c.get(c, 0); per_c[c] = f + 1
            ev += [(p, f, +1, c), (m, f, -1, c)]
        F = max(per_c.values()) if per_c else 0
    else:
        for f, (p, m, _) in enumerate(bands):
            ev += [(p, f, +1, -1), (m, f, -1, -1)]
        F = len(bands)
    if lp is not None:
        ev.append((lp, F, +1, -1)); F += 1
    return np.asarray(ev, dtype=np.int32).reshape(-1, 4), T, F


# ------------------------------------------------------------------ torch.sparse restatement
def ref_get_Batch_Wjxs(x, P_sparse, scales_type="dyadic", diffusion_scales=None, J=5,
                       include_lowpass=True, filter_stack_dim=-1, rescale_filtrations=False):
    """Same calls, same order as pyg_utilities.py:350-385 / :392-444 / :476-479."""
    ds = diffusion_scales
    if ds is not None and not isinstance(ds, str) and ds.dim() == 2:
        Ptxs, Ptx = [x], x
        for _ in range(1, int(2 ** J) + 1):
            Ptx = torch.sparse.mm(P_sparse, Ptx)
            Ptxs.append(Ptx)
        W = torch.stack([torch.stack([Ptxs[t[i - 1]][:, c] - Ptxs[t[i]][:, c] for i in range(1, len(t))], dim=-1)
                         for c, t in enumerate(ds)], dim=1)
        if include_lowpass:
            W = torch.concatenate((W, Ptxs[-1].unsqueeze(-1)), dim=-1)
    elif (ds is not None and not isinstance(ds, str) and ds.dim() == 1) or scales_type == "dyadic":
        if ds is not None and not isinstance(ds, str):
            powers, upper = ds, ds.shape[0]
        else:
            powers, upper = 2 ** torch.arange(J + 1), J + 2
        Ptxs, Ptx = [x], x
        for j in range(1, int(powers[-1]) + 1):
            Ptx = torch.sparse.mm(P_sparse, Ptx)
            if j in powers:
                reps = int((powers == j).sum()) if ds is not None else 1
                Ptxs += [Ptx] * reps
        Wl = [Ptxs[j - 1] - Ptxs[j] for j in range(1, upper)]
        if include_lowpass:
            Wl.append(Ptxs[-1])
        W = torch.stack(Wl, dim=filter_stack_dim)
    elif ds is None and scales_type is None:
        W = torch.sparse.mm(P_sparse, x).unsqueeze(-1)
    else:
        raise NotImplementedError(f"No method implemented for scales_type={scales_type}")
    if rescale_filtrations:
        raise NotImplementedError("Rescaling filtrations within 'get_Batch_Wjxs' has been deprecated.")
    return W


def ref_scatter(track, x0, P_or_Q, diffusion_kwargs, ablate_second_order_wavelets=False):
    """models/escgnn_modular.py:309-362 (orders 0/1/2, triu pair order of :507-516)."""
    if track == "scalar":
        flat = x0
    else:
        N, d = x0.shape
        flat = x0.reshape(N * d, 1)
    W1 = ref_get_Batch_Wjxs(flat, P_or_Q, **diffusion_kwargs)
    parts = [flat.unsqueeze(-1), W1]
    if not ablate_second_order_wavelets:
        i, j = torch.triu_indices(W1.shape[-1], W1.shape[-1], offset=1)
        parts.append(W1[..., i] * W1[..., j])
    W = torch.cat(parts, dim=-1)
    if track == "scalar":
        return W
    return W.view(N, d, 1, -1).permute(0, 2, 1, 3)


# ------------------------------------------------------------------ canonical index structure
def coo_to_csr(indices, values, n):
    """coalesce (sum duplicates) + row-major sort -> (rowptr int32, col int32, val f32)."""
    idx = np.asarray(indices, dtype=np.int64)
    key = idx[0] * n + idx[1]
    uk, inv = np.unique(key, return_inverse=True)
    val = np.zeros(uk.shape[0], dtype=np.float64)
    np.add.at(val, inv, np.asarray(values, dtype=np.float64))
    rows = uk // n
    rowptr = np.zeros(n + 1, dtype=np.int64)
    np.add.at(rowptr, rows + 1, 1)
    return np.cumsum(rowptr).astype(np.int32), (uk % n).astype(np.int32), val.astype(np.float32)


def coo_to_bsr3(indices, values, n3):
    """scalar COO of Q (3n x 3n) -> BSR-3: (rowptr[nb+1], col[nnzb], blk[nnzb,3,3]).
    Block (i,j) exists iff any scalar entry of it exists; missing entries are 0."""
    nb = n3 // 3
    idx = np.asarray(indices, dtype=np.int64)
    bkey = (idx[0] // 3) * nb + (idx[1] // 3)
    ub, inv = np.unique(bkey, return_inverse=True)
    blk = np.zeros((ub.shape[0], 3, 3), dtype=np.float64)
    np.add.at(blk, (inv, idx[0] % 3, idx[1] % 3), np.asarray(values, dtype=np.float64))
    rows = ub // nb
    rowptr = np.zeros(nb + 1, dtype=np.int64)
    np.add.at(rowptr, rows + 1, 1)
    return np.cumsum(rowptr).astype(np.int32), (ub % nb).astype(np.int32), blk.astype(np.float32)


# ------------------------------------------------------------------ C oracle drivers
def _p(a, t):
    return a.ctypes.data_as(ctypes.POINTER(t))


def _prep(*arrs):
    return [np.ascontiguousarray(a) for a in arrs]


def c_chain_csr(rowptr, col, val, X, ev, T, F, f64=False):
    rowptr, col, val, X, ev = _prep(rowptr.astype(np.int32), col.astype(np.int32), val.astype(np.float32),
                                    X.astype(np.float32), ev.astype(np.int32))
    n, C = X.shape
    out = np.empty((n, C, F), dtype=np.float64 if f64 else np.float32)
    fn = lib().orc_chain_csr_f64 if f64 else lib().orc_chain_csr_f32
    rc = fn(_p(rowptr, ctypes.c_int32), _p(col, ctypes.c_int32), _p(val, ctypes.c_float),
            ctypes.c_int64(n), ctypes.c_int(C), _p(X, ctypes.c_float), ctypes.c_int(T),
            ev.ctypes.data_as(ctypes.c_void_p), ctypes.c_int(ev.shape[0]), ctypes.c_int(F),
            out.ctypes.data_as(ctypes.c_void_p))
    assert rc == 0
    return out


def c_chain_bsr3(rowptr, col, blk, x, ev, T, F, f64=False):
    rowptr, col, blk, x, ev = _prep(rowptr.astype(np.int32), col.astype(np.int32), blk.astype(np.float32),
                                    x.astype(np.float32).reshape(-1), ev.astype(np.int32))
    nb = rowptr.shape[0] - 1
    out = np.empty((3 * nb, 1, F), dtype=np.float64 if f64 else np.float32)
    fn = lib().orc_chain_bsr3_f64 if f64 else lib().orc_chain_bsr3_f32
    rc = fn(_p(rowptr, ctypes.c_int32), _p(col, ctypes.c_int32), _p(blk, ctypes.c_float),
            ctypes.c_int64(nb), _p(x, ctypes.c_float), ctypes.c_int(T),
            ev.ctypes.data_as(ctypes.c_void_p), ctypes.c_int(ev.shape[0]), ctypes.c_int(F),
            out.ctypes.data_as(ctypes.c_void_p))
    assert rc == 0
    return out


def c_chain_csr_bwd(rowptr, col, val, G, ev, T):
    rowptr, col, val, G, ev = _prep(rowptr.astype(np.int32), col.astype(np.int32), val.astype(np.float32),
                                    G.astype(np.float32), ev.astype(np.int32))
    n, C, F = G.shape
    dX = np.empty((n, C), dtype=np.float32)
    rc = lib().orc_chain_csr_bwd_f32(
        _p(rowptr, ctypes.c_int32), _p(col, ctypes.c_int32), _p(val, ctypes.c_float),
        ctypes.c_int64(n), ctypes.c_int(C), _p(G, ctypes.c_float), ctypes.c_int(T),
        ev.ctypes.data_as(ctypes.c_void_p), ctypes.c_int(ev.shape[0]), ctypes.c_int(F), _p(dX, ctypes.c_float))
    assert rc == 0
    return dX


def c_chain_bsr3_bwd(rowptr, col, blk, G, ev, T):
    rowptr, col, blk, G, ev = _prep(rowptr.astype(np.int32), col.astype(np.int32), blk.astype(np.float32),
                                    G.astype(np.float32), ev.astype(np.int32))
    nb = rowptr.shape[0] - 1
    F = G.shape[-1]
    dx = np.empty((3 * nb, 1), dtype=np.float32)
    rc = lib().orc_chain_bsr3_bwd_f32(
        _p(rowptr, ctypes.c_int32), _p(col, ctypes.c_int32), _p(blk, ctypes.c_float),
        ctypes.c_int64(nb), _p(G, ctypes.c_float), ctypes.c_int(T),
        ev.ctypes.data_as(ctypes.c_void_p), ctypes.c_int(ev.shape[0]), ctypes.c_int(F), _p(dx, ctypes.c_float))
    assert rc == 0
    return dx
