#!/usr/bin/env python3
"""K1 / K2 of SURVEY 8d: ONE product Y = P X (CSR, C = 3) and y = Q x (BSR-3) -- what one torch.sparse.mm of the
reference is replaced by -- on the cfg2 batch (4096 graphs x 500 nodes, k = 10), HIP-event timed, against the
algorithmic bytes 4(N+1) + 8 nnz + 8 N C and 4(N+1) + 40 nnz + 24 N.  One JSON line."""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--graphs", type=int, default=4096)
    ap.add_argument("--nodes", type=int, default=500)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--reps", type=int, default=20)
    args = ap.parse_args()
    from esc_gnn_amd import synthetic
    b = synthetic.ellipsoid_batch(args.graphs, args.nodes, args.k, seed=457893, workers=min(16, os.cpu_count()))
    import torch
    from esc_gnn_amd import _lib
    lib = _lib.load()
    dev = torch.device("cuda:0")
    t = lambda a: torch.from_numpy(a).to(dev)
    rp, ci, pv, blk = t(b["rowptr"]), t(b["col"]), t(b["pval"]), t(b["blk"])
    N, nnz, C = b["N"], b["nnz"], 3
    x = torch.randn(N, C, device=dev); y = torch.empty_like(x)
    xv = torch.randn(3 * N, device=dev); yv = torch.empty_like(xv)
    st = torch.cuda.current_stream().cuda_stream

    def timed(fn):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.reps):
            fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / args.reps

    k1 = timed(lambda: _lib.check(lib.escat_csr_spmm_f32(rp.data_ptr(), ci.data_ptr(), pv.data_ptr(), N, nnz, x.data_ptr(), C,
                                                         y.data_ptr(), C, C, st), "spmm"))
    k2 = timed(lambda: _lib.check(lib.escat_bsr3_spmv_f32(rp.data_ptr(), ci.data_ptr(), blk.data_ptr(), N, nnz, xv.data_ptr(),
                                                          yv.data_ptr(), st), "spmv"))
    b1, b2 = 4 * (N + 1) + 8 * nnz + 8 * N * C, 4 * (N + 1) + 40 * nnz + 24 * N
    print(json.dumps({"workload": f"{args.graphs} graphs x {args.nodes} nodes, k={args.k}: one product", "N": N, "nnz": nnz,
                      "K1_csr_spmm_C3": {"ms": k1, "bytes": b1, "GBs": b1 / k1 * 1e-6, "frac_of_8TBs": b1 / k1 * 1e-6 / 8000},
                      "K2_bsr3_spmv": {"ms": k2, "bytes": b2, "GBs": b2 / k2 * 1e-6, "frac_of_8TBs": b2 / k2 * 1e-6 / 8000}}))


if __name__ == "__main__":
    main()
