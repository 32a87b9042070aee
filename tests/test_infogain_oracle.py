"""CPU: the infogain restatement (oracle/infogain_oracle.py) against the goldens generated from the reference's
models/infogain.py (tests/golden/make_infogain_goldens.py), and the host logic of esc_gnn_amd.infogain that needs no GPU."""
import contextlib
import io
import numpy as np
import pytest
import torch

from oracle import infogain_oracle as io_


def _ops(sg):
    N = int(sg["N"])
    P = torch.sparse_coo_tensor(torch.from_numpy(sg["P_indices"]), torch.from_numpy(sg["P_values"]), (N, N))
    Q = torch.sparse_coo_tensor(torch.from_numpy(sg["Q_indices"]), torch.from_numpy(sg["Q_values"]), (3 * N, 3 * N))
    return P, Q


def _feats(sg, ig):
    P, Q = _ops(sg)
    return {"rand": (torch.from_numpy(sg["x_C3"]), P, None), "mix": (torch.from_numpy(ig["x_mix"]), P, None),
            "vec": (torch.from_numpy(sg["x_v"]).reshape(-1, 1), Q, 3)}


def test_powers_restatement_bit_exact(sg, ig):
    P, _ = _ops(sg)
    assert np.array_equal(io_.powers(P, torch.from_numpy(ig["x_mix"]), 16).numpy(), ig["Ptxs_mix"])


@pytest.mark.parametrize("name", ["rand", "mix", "vec"])
def test_divergences_vs_reference(sg, ig, name):
    """per-graph divergences.  fp32 sums of ~n terms in a different association than torch's vectorised sum: the
    tolerance is 1e-5 of the largest divergence of the table plus 2e-6 relative."""
    x, op, vfd = _feats(sg, ig)[name]
    gp = ig["graph_ptr"] * (1 if vfd is None else 3)
    Ptxs = io_.powers(op, x, 16)
    for metric in ("kl", "l1"):
        for floor, ftag in ((None, "auto"), (1e-2, "f0p01")):
            for start in (2, 0):
                want = ig[f"div_{name}_{metric}_{ftag}_s{start}"]
                got = torch.stack([io_.divergence_by_channel(Ptxs[:, gp[g]:gp[g + 1]], start, vfd, metric, floor)
                                   for g in range(len(gp) - 1)]).numpy()
                assert got.shape == want.shape
                assert np.abs(got - want).max() <= 1e-5 * np.abs(want).max() + 2e-6 * np.abs(want).max(), (metric, ftag, start)


def _batches(sg, ig, feat, op, split, y, vfd):
    P, Q = _ops(sg)
    gp = ig["graph_ptr"]
    x = {"x": torch.from_numpy(sg["x_C3"]), "x_mix": torch.from_numpy(ig["x_mix"]), "x_v": torch.from_numpy(sg["x_v"])}[feat]
    A = (P if op == "P" else Q).to_dense()
    d = 1 if op == "P" else 3
    out = []
    for lo, hi in split:
        r0, r1 = int(gp[lo]), int(gp[hi])
        out.append({"x": x[r0:r1], "op": A[d * r0:d * r1, d * r0:d * r1].to_sparse(), "graph_ptr": gp[lo:hi + 1] - gp[lo],
                    "y": torch.from_numpy(y[lo:hi])})
    return out


CFGS = {
    "a": dict(feat="x", op="P", vfd=None, split=[(0, 3)], y="reg", task="regression", kw={}),
    "b": dict(feat="x_mix", op="P", vfd=None, split=[(0, 2), (2, 3)], y="bin", task="binary_classification", kw={}),
    "c": dict(feat="x_mix", op="P", vfd=None, split=[(0, 3)], y="reg", task="regression",
              kw=dict(T=8, start_from_t2=False, metric="l1", quantiles=(0.25, 0.5, 0.75))),
    "d": dict(feat="x_v", op="Q", vfd=3, split=[(0, 1), (1, 3)], y="reg", task="regression", kw={}),
    "e": dict(feat="x", op="P", vfd=None, split=[(0, 3)], y="bin", task="binary_classification",
              kw=dict(above_zero_floor=1e-2, auto_process=False, quantiles=(0.5, 0.9, 0.99))),
    "f": dict(feat="x_mix", op="P", vfd=None, split=[(0, 3)], y="reg", task="regression", kw=dict(T=32, strategy="average")),
}


@pytest.mark.parametrize("name", list(CFGS))
def test_whole_selection_vs_reference(sg, ig, name):
    c = CFGS[name]
    y = ig["y_bin"] if c["y"] == "bin" else ig["y_reg"]
    un, ts = io_.calc_infogain_wavelet_scales(_batches(sg, ig, c["feat"], c["op"], c["split"], y, c["vfd"]), c["task"],
                                              vector_feat_dim=c["vfd"], **c["kw"])
    assert np.array_equal(ts.numpy(), ig[f"scales_{name}"])
    assert (un is None) == bool(ig[f"uninf_none_{name}"])
    if un is not None:
        assert np.array_equal(un.numpy(), ig[f"uninf_{name}"])


def test_degenerate_column_raises_like_the_reference():
    x = torch.ones(3, 5, 1)
    with pytest.raises(TypeError):
        io_.divergence_by_channel(x, 0, None, "kl", None)


def test_uninformative_channel_handling(ig):
    """oracle and product host logic against the reference's process_uninformative_channels."""
    from esc_gnn_amd import infogain as prod
    tab = torch.from_numpy(ig["uninf_table"])
    for mod in (io_.process_uninformative, prod.process_uninformative_channels):
        for strat in ("drop", "average", "dyadic"):
            src = tab.clone().float() if strat == "dyadic" else tab.clone()
            with contextlib.redirect_stdout(io.StringIO()):
                un, ts = mod(src, strat, 16)
            assert np.array_equal(un.numpy(), ig[f"uninf_{strat}_idx"])
            assert np.array_equal(ts.numpy(), ig[f"uninf_{strat}_table"])
        with pytest.raises(RuntimeError):          # a long table cannot take the float dyadic row (reference behaviour)
            with contextlib.redirect_stdout(io.StringIO()):
                mod(tab.clone(), "dyadic", 16)
    assert np.array_equal(prod.get_avg_infogain_wavelet_scales(tab).numpy(), ig["avg_scales"])
    with pytest.raises(NotImplementedError):
        prod.get_avg_infogain_wavelet_scales(tab, "mean")


def test_select_scales_host_logic(sg, ig):
    """esc_gnn_amd.infogain.select_scales (torch, device-agnostic) on the reference's own per-graph divergences."""
    from esc_gnn_amd import infogain as prod
    divs = torch.from_numpy(ig["div_rand_kl_auto_s2"])
    with pytest.warns(UserWarning):
        ts = prod.select_scales(divs, torch.from_numpy(ig["y_reg"]), "regression", 16, 3, "cpu")
    assert np.array_equal(ts.numpy(), ig["scales_a"])
    divs = torch.from_numpy(ig["div_rand_kl_f0p01_s2"])
    ts = prod.select_scales(divs, torch.from_numpy(ig["y_bin"]), "binary_classification", 16, 3, "cpu", (0.5, 0.9, 0.99))
    assert np.array_equal(ts.numpy(), ig["scales_e"])
    assert prod.all_powers_plan(4).events == ((0, 0, 1, -1), (1, 1, 1, -1), (2, 2, 1, -1), (3, 3, 1, -1), (4, 4, 1, -1))
    with pytest.raises(NotImplementedError):
        prod.divergence_by_graph(torch.zeros(4, 1, 5), torch.tensor([0, 4]), 2, "emd")
    with pytest.raises(ValueError):
        prod.divergence_by_graph(torch.zeros(4, 1, 5), torch.tensor([0, 4]), 2, "hellinger")
