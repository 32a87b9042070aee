"""
Operator store: a whole dataset's P and Q kept canonical + packed in HBM, one part per sample, so that every epoch's
shuffled mini-batches are views (ScatterOperator.select) instead of a per-step COO upload + coalesce
(the reference re-reads /{P,Q}/{idx} per sample and collates int64 COO every step:
models/escgnn_data_classes.py:113-143, 20-27).
"""
from typing import Dict, Optional, Sequence
import numpy as np
import torch

from .operators import ScatterOperator, FactoredVectorOperator


def load_operator_store(path: str, device, indices: Optional[Sequence[int]] = None, pack: bool = True,
                        factored: bool = False) -> Dict:
    """
    Reads the reference's HDF5 operator layout (/{P,Q}/{idx}/{indices,values,size},
    scripts/python/generate_ellipsoid_dataset.py:1185-1206) -- both the version-2 files esc_gnn_amd.hdf5_min writes and
    the classic files h5py writes (superblock 0, symbol-table groups, chunked + gzip: esc_gnn_amd.hdf5_classic, written
    from the format specification; no HDF5 library is available here to cross-check: parity unpinned) -- for the
    samples `indices` (default: all, in numeric order)
    and returns {"P": ScatterOperator, "Q": ScatterOperator, "graph_ptr": int64 (B+1,) node offsets, "ids": list}.
    Values stored as fp16 are widened to fp32 as the reference loader does (escgnn_data_classes.py:137-142).
    The operators carry one part per sample; with pack=True the SELL packs of all four forms are built now.
    factored=True (files written with store_frames=True, esc_gnn_amd.process_pyg_data.write_dataset_hdf5) also returns
    "Q_factored": the vector operator as frames + signed scalar operator built from P's stored values.  It equals the
    stored Q to fp32 rounding only if the file holds fp32 values: fp16 storage rounds every Q entry on its own, which no
    factorisation reproduces (the factored operator is then the closer one to the operator before rounding).
    """
    from .hdf5_min import open_h5
    f = open_h5(path)
    try:
        ids = sorted((int(k) for k in f["P"].keys())) if indices is None else [int(i) for i in indices]
        out = {}
        gP, gQ = f["P"], f["Q"]
        sizes = [int(gP[str(i)]["size"][...][0]) for i in ids]
        gp = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
        for key, unit in (("P", 1), ("Q", 3)):
            rows, cols, vals = [], [], []
            for g, i in enumerate(ids):
                grp = (gP if key == "P" else gQ)[str(i)]
                idx = np.asarray(grp["indices"][...], dtype=np.int64)
                if tuple(int(v) for v in grp["size"][...]) != (unit * sizes[g], unit * sizes[g]):
                    raise ValueError(f"escat: {key}/{i} has size {grp['size'][...]} but P/{i} has {sizes[g]} nodes")
                rows.append(idx[0] + unit * gp[g]); cols.append(idx[1] + unit * gp[g])
                vals.append(np.asarray(grp["values"][...]).astype(np.float32))
            n = unit * int(gp[-1])
            A = torch.sparse_coo_tensor(torch.from_numpy(np.stack([np.concatenate(rows), np.concatenate(cols)])).to(device),
                                        torch.from_numpy(np.concatenate(vals)).to(device), (n, n))
            op = ScatterOperator.from_torch(A, graph_ptr=torch.from_numpy(unit * gp))
            if pack:
                forms = (op.bsr3(), op.bsr3_t()) if (key == "Q" and op.has_bsr3) else (op.csr(), op.csr_t())
                for form in forms:
                    form.sell()
            out[key] = op
        if factored:
            if "frames" not in f.keys():
                raise KeyError("escat: this file has no /frames group (write it with store_frames=True)")
            frames = np.concatenate([np.asarray(f["frames"][str(i)]["values"][...], dtype=np.float32) for i in ids])
            pc = out["P"].csr()
            out["Q_factored"] = FactoredVectorOperator.from_frames(
                pc.rowptr, pc.col, pc.val, torch.from_numpy(frames).to(device),
                parts=ScatterOperator.parts_from_graph_ptr(torch.from_numpy(gp).to(device)))
            if pack:
                out["Q_factored"].signed.csr().sell(); out["Q_factored"].signed.csr_t().sell()
        out["graph_ptr"] = torch.from_numpy(gp).to(device)
        out["ids"] = ids
        return out
    finally:
        f.close()
