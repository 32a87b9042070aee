#!/usr/bin/env python3
"""cfg5 end to end: build P/Q on the GPU for a dataset of ellipsoid graphs AND write the reference's HDF5 layout
(esc_gnn_amd.process_pyg_data.write_dataset_hdf5), then load it back as an operator store."""
import argparse, json, os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("--graphs", type=int, default=20000); ap.add_argument("--nodes", type=int, default=128); ap.add_argument("--k", type=int, default=5)
args = ap.parse_args()
import numpy as np, torch
from esc_gnn_amd import synthetic
from esc_gnn_amd.process_pyg_data import write_dataset_hdf5
from esc_gnn_amd.store import load_operator_store
dev = torch.device("cuda:0")
clouds = synthetic.ellipsoid_clouds(args.graphs, args.nodes, seed=457892)
pos = torch.from_numpy(clouds.reshape(-1, 3)).to(dev)
gp = torch.arange(args.graphs + 1, dtype=torch.int32) * args.nodes
kw = {"kernel": "gaussian", "gaussian_eps": 0.05}
path = os.path.join(tempfile.mkdtemp(), "pq.h5")
write_dataset_hdf5(path, pos[:args.nodes * 64], gp[:65], args.k, kw)            # warm-up
res = {"graphs": args.graphs}
for comp in ("gzip", None):      # gzip = the reference's create_dataset(..., compression='gzip') (the default); None = contiguous
    t0 = time.perf_counter(); info = write_dataset_hdf5(path, pos, gp, args.k, kw, compression=comp); dt = time.perf_counter() - t0
    size = os.path.getsize(path)
    t0 = time.perf_counter(); store = load_operator_store(path, dev); torch.cuda.synchronize(); dl = time.perf_counter() - t0
    res[str(comp)] = {"build_and_write_s": round(dt, 2), "graphs_per_s": round(args.graphs / dt), "file_MB": round(size / 1e6, 1),
                      "load_store_s": round(dl, 2), "load_graphs_per_s": round(args.graphs / dl)}
    del store
print(json.dumps(res))
