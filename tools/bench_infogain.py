"""Infogain fit at BASELINE cfg2's size (4096 graphs x 500 nodes, k=10, T=16): all-powers chain + divergence kernel per
batch, scalar (C=3) and vector tracks; CPU restatement (oracle/infogain_oracle.py, the reference's per-graph loops) on
a bounded sample beside it.   python tools/bench_infogain.py [--graphs 4096] [--cpu-graphs 16]"""
import argparse
import json
import os
import sys
import time
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--graphs", type=int, default=4096)
    ap.add_argument("--nodes", type=int, default=500)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--T", type=int, default=16)
    ap.add_argument("--cpu-graphs", type=int, default=16)
    ap.add_argument("--iters", type=int, default=20)
    a = ap.parse_args()
    from esc_gnn_amd import synthetic
    from esc_gnn_amd.process_pyg_data import build_PQ_batch
    from esc_gnn_amd.infogain import diffusion_powers, divergence_by_graph
    dev = torch.device("cuda:0")
    B, n, T = a.graphs, a.nodes, a.T
    pos = torch.from_numpy(synthetic.ellipsoid_clouds(B, n, seed=31).reshape(-1, 3)).to(dev)
    gp = torch.arange(B + 1, dtype=torch.int32, device=dev) * n
    b = build_PQ_batch(pos, gp, a.k, {"kernel": "gaussian", "gaussian_eps": 0.05})
    res = {"graphs": B, "nodes": n, "T": T, "nnz": b["nnz"]}

    def timed(fn):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / a.iters

    for name, x, op, d in (("scalar_C3", pos, b["P"], 1), ("vector", pos.reshape(-1, 1), b["Q"], 3)):
        W = diffusion_powers(x, op, T)
        t_pow = timed(lambda: diffusion_powers(x, op, T))
        t_div = timed(lambda: divergence_by_graph(W, gp * d, 2, "kl", None, check=False, max_graph_rows=n * d))
        t_div3 = timed(lambda: divergence_by_graph(W, gp * d, 2, "kl", None, check=False, max_graph_rows=0))
        bytes_w = W.numel() * 4
        res[name] = {"powers_ms": round(t_pow, 4), "divergence_ms": round(t_div, 4), "divergence_three_pass_ms": round(t_div3, 4),
                     "divergence_GBs_of_one_read_of_W": round(bytes_w / t_div / 1e6, 1),
                     "graphs_per_s": round(B / (t_pow + t_div) * 1e3, 1)}
    # CPU: the reference's structure (T sparse products on the batch, then Python loops over graphs x channels x steps)
    from oracle import infogain_oracle as io_
    m = a.cpu_graphs
    torch.set_num_threads(1)
    pc = b["P"].csr()
    rp = pc.rowptr[: m * n + 1].long().cpu(); col = pc.col[: int(rp[-1])].long().cpu(); val = pc.val[: int(rp[-1])].cpu()
    Pm = torch.sparse_csr_tensor(rp, col, val, (m * n, m * n)).to_sparse_coo()
    xs = pos[: m * n].cpu()
    t0 = time.perf_counter()
    Pt = io_.powers(Pm, xs, T)
    for g in range(m):
        io_.divergence_by_channel(Pt[:, g * n:(g + 1) * n], 2, None, "kl", None)
    dt = time.perf_counter() - t0
    res["cpu_port_scalar_C3"] = {"graphs_per_s": round(m / dt, 2), "cores": 1, "sample": f"{m} graphs"}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
