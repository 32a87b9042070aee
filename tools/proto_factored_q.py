"""Numerical check of the factored form of the vector operator (design note in DESIGN.md, 'next').

process_pyg_data builds Q_ij = p_ij O_i S_ij O_j^T with S_ij = diag(sign<O_i[:,c], O_j[:,c]>) (column_dot alignment,
process_pyg_data.py:306-313, :492).  With D = blockdiag(O_i) (orthogonal) that is Q = D Pt D^T, Pt = three signed scalar
operators on P's pattern, Pt_c[i,j] = p_ij s_ij,c -- hence Q^t = D Pt^t D^T: the whole vector chain is one rotation into
the local frames, three SCALAR chains (4 B + 3 sign bits per nonzero instead of 36 B), and one rotation back per band.
This script runs that with the existing kernels (three C=1 CSR chains) and compares with the BSR-3 chain."""
import json
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from esc_gnn_amd import synthetic, get_Batch_Wjxs, ScatterOperator
    from esc_gnn_amd.process_pyg_data import build_PQ_batch
    dev = torch.device("cuda:0")
    out = {}
    for tag, B, n, k, J in (("cfg2-like", 512, 500, 10, 4), ("long chain", 64, 500, 16, 6)):
        pos = torch.from_numpy(synthetic.ellipsoid_clouds(B, n, seed=5).reshape(-1, 3)).to(dev)
        gp = torch.arange(B + 1, dtype=torch.int32, device=dev) * n
        b = build_PQ_batch(pos, gp, k, {"kernel": "gaussian", "gaussian_eps": 0.05})
        N = B * n
        rp = b["rowptr"].long()
        rows = torch.repeat_interleave(torch.arange(N, device=dev), rp[1:] - rp[:-1])
        cols = b["col"].long()
        O = b["frames"]                                                   # (N, 3, 3), columns = frame vectors
        s = torch.where((O[rows] * O[cols]).sum(1) >= 0, 1.0, -1.0)       # (nnz, 3): sign per frame column
        # the factorisation reproduces the builder's blocks
        blk = b["pval"][:, None, None] * torch.einsum("eac,ec,ebc->eab", O[rows], s, O[cols])
        out[tag + ": max |p O_i S O_j^T - block|"] = float((blk - b["blk"]).abs().max())
        x = pos                                                           # vector feature (N, 3)
        Wq = get_Batch_Wjxs(x.reshape(-1, 1), b["Q"], J=J).reshape(N, 3, -1)
        u = torch.einsum("nac,na->nc", O, x)                              # D^T x: local coordinates
        Wl = []
        for c in range(3):
            opc = ScatterOperator.from_csr(b["rowptr"], b["col"], (b["pval"] * s[:, c]).contiguous(), N, validate=False)
            Wl.append(get_Batch_Wjxs(u[:, c:c + 1].contiguous(), opc, J=J)[:, 0])          # (N, F)
        Wf = torch.einsum("nac,ncf->naf", O, torch.stack(Wl, dim=1))      # rotate every band back
        err = (Wf - Wq).abs().amax(dim=(0, 1)) / Wq.abs().amax(dim=(0, 1))
        out[tag + ": max rel err per band, factored vs BSR-3 chain"] = [float(e) for e in err]
    print(json.dumps(out))


if __name__ == "__main__":
    main()
