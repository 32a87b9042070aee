import cProfile, pstats, os, sys, tempfile, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from esc_gnn_amd import synthetic
from esc_gnn_amd.process_pyg_data import write_dataset_hdf5
dev = torch.device("cuda:0")
G, n = 8000, 128
clouds = synthetic.ellipsoid_clouds(G, n, seed=457892)
pos = torch.from_numpy(clouds.reshape(-1, 3)).to(dev)
gp = torch.arange(G + 1, dtype=torch.int32) * n
kw = {"kernel": "gaussian", "gaussian_eps": 0.05}
path = os.path.join(tempfile.mkdtemp(), "pq.h5")
write_dataset_hdf5(path, pos[:n * 64], gp[:65], 5, kw)
pr = cProfile.Profile(); pr.enable()
t0 = time.perf_counter(); write_dataset_hdf5(path, pos, gp, 5, kw); dt = time.perf_counter() - t0
pr.disable()
print("graphs/s", G / dt)
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
