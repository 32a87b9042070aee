#!/usr/bin/env python3
"""BASELINE cfg5: GPU P/Q operator construction (kNN + local-PCA frames + O(3) alignment blocks) for a dataset of
ellipsoid graphs; graphs/s of build_PQ_batch (positions resident on the GPU, CSR/BSR-3 out) and the CPU oracle's
restatement of process_pyg_data beside it on a small sample."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--graphs", type=int, default=100000)
    ap.add_argument("--nodes", type=int, default=128)
    ap.add_argument("--k", type=int, default=5)
    ap.add_argument("--chunk", type=int, default=25000, help="graphs per build_PQ_batch call")
    ap.add_argument("--cpu-sample", type=int, default=8)
    args = ap.parse_args()
    import numpy as np, torch
    from esc_gnn_amd import synthetic
    from esc_gnn_amd.process_pyg_data import build_PQ_batch
    dev = torch.device("cuda:0")
    t0 = time.time()
    clouds = synthetic.ellipsoid_clouds(args.graphs, args.nodes, seed=457892)
    gen_s = time.time() - t0
    pos = torch.from_numpy(clouds.reshape(-1, 3)).to(dev)
    kw = {"kernel": "gaussian", "gaussian_eps": 0.05}
    def run():
        nnz = 0
        for g0 in range(0, args.graphs, args.chunk):
            g1 = min(args.graphs, g0 + args.chunk)
            gp = (torch.arange(g1 - g0 + 1, dtype=torch.int32, device=dev) * args.nodes)
            b = build_PQ_batch(pos[g0 * args.nodes:g1 * args.nodes], gp, args.k, kw, max_graph_nodes=args.nodes)
            nnz += b["nnz"]
        torch.cuda.synchronize()
        return nnz
    run()
    t0 = time.perf_counter(); nnz = run(); dt = time.perf_counter() - t0
    out = {"workload": f"cfg5: build P (CSR) + Q (BSR-3) for {args.graphs} ellipsoid graphs x {args.nodes} nodes, kNN k={args.k}",
           "graphs_per_s": args.graphs / dt, "seconds": dt, "nnz": nnz, "input_gen_s": gen_s}
    if args.cpu_sample:
        from oracle import builder_oracle as bo
        torch.set_num_threads(1)
        t0 = time.perf_counter()
        for g in range(args.cpu_sample):
            ei = bo.knn_sym_edges(clouds[g], args.k)
            bo.build_PQ(clouds[g], ei, kw)
        cs = time.perf_counter() - t0
        out["cpu_oracle_graphs_per_s_1core"] = args.cpu_sample / cs
    print(json.dumps(out))

if __name__ == "__main__":
    main()
