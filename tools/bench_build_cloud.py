#!/usr/bin/env python3
"""cfg3's operator built on the GPU: one ellipsoid cloud (default 2 M points, k = 16) through build_PQ_batch's grid
route (escat_build_knn_P_grid + frames + Q blocks), timed, and its pattern compared with scipy's cKDTree + the host
symmetrisation (synthetic.knn_sym_csr) that bench.py's cfg3 leg uses.  One JSON line."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=2_000_000)
    ap.add_argument("--k", type=int, default=16)
    ap.add_argument("--no-check", action="store_true")
    args = ap.parse_args()
    from esc_gnn_amd import synthetic
    rng = np.random.RandomState(457892)
    phi = rng.uniform(0, 2 * np.pi, args.n)
    theta = np.arccos(2 * rng.uniform(0, 1, args.n) - 1)
    pos = np.column_stack([3.0 * np.sin(theta) * np.cos(phi), np.sin(theta) * np.sin(phi), np.cos(theta)]).astype(np.float32)
    ref = None
    if not args.no_check:
        t0 = time.time()
        ref = synthetic.knn_sym_csr(pos, args.k, workers=min(16, os.cpu_count()))
        host_s = time.time() - t0
    import torch
    from esc_gnn_amd.process_pyg_data import build_PQ_batch
    dev = torch.device("cuda:0")
    p = torch.from_numpy(pos).to(dev)
    gp = torch.tensor([0, args.n], dtype=torch.int32)
    kw = {'kernel': 'epanechnikov', 'r_cut': 5.0}
    times = []
    for r in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        b = build_PQ_batch(p, gp, args.k, kw)
        torch.cuda.synchronize(); times.append(time.perf_counter() - t0)
    out = {"workload": f"one ellipsoid cloud, {args.n} points, k={args.k}: kNN (grid) + symmetrise + P + frames + Q blocks + ingest",
           "nnz": b["nnz"], "gpu_build_s_first": times[0], "gpu_build_s": min(times[1:]), "points_per_s": args.n / min(times[1:])}
    if ref is not None:
        out["host_ckdtree_sym_s"] = host_s
        out["pattern_equal_to_ckdtree"] = bool(np.array_equal(b["rowptr"].cpu().numpy(), ref[0]) and np.array_equal(b["col"].cpu().numpy(), ref[1]))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
