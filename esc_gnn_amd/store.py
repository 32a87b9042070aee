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


class ScatteringStore:
    """
    The scattering coefficients of a WHOLE dataset, computed once and kept in HBM; a mini-batch is then a gather.

    ESCGNNModular.forward runs _scatter on both tracks at every training step (/root/reference/models/escgnn_modular.py:674,
    :685), outside the autograd graph (training/train.py:303-321), on inputs that never change: the node features of the
    ellipsoid experiment are the positions (config/yaml_files/ellipsoids/experiment.yaml: scalar_feat_key = vector_feat_key
    = pos), the operators come from the file, and the transform has no parameter.  The coefficients of a sample do not depend
    on which other samples share its batch (the operators are block-diagonal), so they can be computed for the whole
    dataset in a few large launches -- where the kernels run at their throughput, instead of 32-graph batches
    (batch_size: 32) that fill a sixteenth of the GPU -- and every later step, of every epoch, reads its batch out of HBM.
    ScatteringCache (esc_gnn_amd.scatter) is the per-batch form of the same observation.

        ops = load_operator_store("pq_tensor_data_512.h5", "cuda")           # P, Q (+ Q_factored), graph_ptr
        sstore = ScatteringStore(ops, pos_all, pos_all, {"diffusion_kwargs": {...}})
        Ws, Wv = sstore.batch(sample_ids)                                    # (n, C, W), (n, 1, 3, W) as _scatter returns them
        # or, behind the model's own two _scatter calls:
        model.escat_store = sstore;  ESCGNNModular._scatter = escat_scatter
        for sample_ids, batch in loader:  sstore.set_batch(sample_ids);  out = model(batch)

    ops: {"P": ScatterOperator, "Q": ScatterOperator (or "Q_factored"), "graph_ptr": (B + 1,) node offsets} with one part
    per sample (load_operator_store, or from_csr / from_bsr3 with parts=ScatterOperator.parts_from_graph_ptr).
    x_scalar (N, C), x_vector (N, 3): the node features of all samples in the store's node order.  Gradient-free by
    construction (like the reference's use); a caller that needs d/dx calls scatter() itself.
    """

    def __init__(self, ops: Dict, x_scalar: torch.Tensor, x_vector: torch.Tensor, kwargs: Dict,
                 ablate_second_order_wavelets: bool = False, chunk_graphs: int = 4096):
        from .scatter import scatter_both
        P = ops["P"]
        Q = ops.get("Q_factored", None) or ops["Q"]
        gp = ops["graph_ptr"].to(device=P.device, dtype=torch.int64)
        B = int(gp.numel()) - 1
        if x_scalar.shape[0] != int(gp[-1]) or x_vector.shape[0] != int(gp[-1]) or x_vector.shape[1] != 3:
            raise ValueError("escat: x_scalar (N, C) and x_vector (N, 3) must cover the store's nodes")
        self.graph_ptr = gp
        self._gp_host = gp.cpu().numpy()
        self.n_samples = B
        self.kwargs = kwargs
        self.ablate = bool(ablate_second_order_wavelets)
        self.Ws = self.Wv = None
        self._active = None
        xs = x_scalar.detach().to(device=P.device, dtype=torch.float32)
        xv = x_vector.detach().to(device=P.device, dtype=torch.float32)
        with torch.no_grad():
            for g0 in range(0, B, chunk_graphs):
                g1 = min(B, g0 + chunk_graphs)
                ids = np.arange(g0, g1)
                r0, r1 = int(gp[g0]), int(gp[g1])
                Pv = P.select(ids)
                ws, wv = scatter_both(x0_scalar=xs[r0:r1].contiguous(), P=Pv, x0_vector=xv[r0:r1].contiguous(),
                                      Q=Q.select(Pv), kwargs=kwargs, ablate_second_order_wavelets=self.ablate)
                if self.Ws is None:
                    self.Ws = torch.empty((int(gp[-1]),) + tuple(ws.shape[1:]), dtype=torch.float32, device=P.device)
                    self.Wv = torch.empty((int(gp[-1]),) + tuple(wv.shape[1:]), dtype=torch.float32, device=P.device)
                self.Ws[r0:r1] = ws
                self.Wv[r0:r1] = wv

    def rows_of(self, sample_ids) -> torch.Tensor:
        """node rows of the samples, in batch order (what PyG's collate concatenates)"""
        from .operators import rows_of_ranges
        gp = self.graph_ptr
        if torch.is_tensor(sample_ids):
            ids_h = sample_ids.detach().to("cpu", torch.int64).numpy().reshape(-1)
            ids_d = sample_ids.detach().to(device=gp.device, dtype=torch.int64).flatten()
        else:
            ids_h = np.asarray(sample_ids, dtype=np.int64).reshape(-1)
            ids_d = None
        if ids_h.size and (int(ids_h.min()) < 0 or int(ids_h.max()) >= self.n_samples):
            raise IndexError("escat: sample id outside the store")
        if ids_d is None and int((self._gp_host[1:] - self._gp_host[:-1])[ids_h].sum()) > (1 << 17):
            ids_d = torch.from_numpy(ids_h).to(gp.device)
        return rows_of_ranges(self._gp_host, ids_h, gp.device, 1, gp if ids_d is not None else None, ids_d)

    def batch(self, sample_ids):
        """(scatter(track='scalar') (n, C, W), scatter(track='vector') (n, 1, 3, W)) of the batch made of these samples"""
        rows = self.rows_of(sample_ids)
        return self.Ws[rows], self.Wv[rows]

    def set_batch(self, sample_ids):
        """the samples of the batch the model is about to see: escat_scatter (a model with `escat_store` set) then answers
        its two _scatter calls from the store; None switches that off"""
        self._active = None if sample_ids is None else self.batch(sample_ids)

    def lookup(self, track: str, x0: torch.Tensor):
        """the active batch's coefficients for a track, or None (no active batch, or x0 is not that batch's size)"""
        if self._active is None:
            return None
        W = self._active[0 if track == 'scalar' else 1]
        return W if W.shape[0] == x0.shape[0] else None
