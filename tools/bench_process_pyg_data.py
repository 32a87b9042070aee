#!/usr/bin/env python3
"""per-call cost of the function-level drop-in process_pyg_data (one graph per call, as the reference's dataset scripts
use it: generate_ellipsoid_dataset.py:1167) against the batched builder.  One JSON line."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from esc_gnn_amd import synthetic
from esc_gnn_amd.process_pyg_data import process_pyg_data, build_PQ_batch


class D(dict):
    __getattr__ = dict.get


def main():
    n, k, G = 128, 5, 200
    clouds = synthetic.ellipsoid_clouds(G, n, seed=457892)
    kw = {"kernel": "gaussian", "gaussian_eps": 0.05}
    gck = {'r_cutoff': 5.0, 'gaussian_eps': 4.0, 'max_num_neighbors': k}
    res = {}
    for name, args in (("knn_dict", dict(return_data_object=False)), ("knn_data_object", dict(return_data_object=True))):
        for r in range(2):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for g in range(G):
                process_pyg_data(D(pos=torch.from_numpy(clouds[g])), graph_construction='k-nn', graph_construction_kwargs=gck,
                                 local_pca_kernel_fn_kwargs=kw, **args)
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
        res[name + "_ms_per_graph"] = dt / G * 1e3
    dev = torch.device("cuda:0")
    pos = torch.from_numpy(clouds.reshape(-1, 3)).to(dev)
    gp = torch.arange(G + 1, dtype=torch.int32) * n
    for r in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        build_PQ_batch(pos, gp, k, kw)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    res["batched_ms_per_graph"] = dt / G * 1e3
    print(json.dumps(res))


if __name__ == "__main__":
    main()
