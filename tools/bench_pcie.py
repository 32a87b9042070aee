"""PCIe-inclusive rates for cfg2: the product boundary takes device tensors, like the reference's module
(the trainer moves the batch with batch.to(device) first).  This measures what a step costs when the batch's dense
features, or also its operators, have to cross the bus every step -- it is never the bench's `value`."""
import json
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from esc_gnn_amd import synthetic, get_Batch_Wjxs
    from esc_gnn_amd.process_pyg_data import build_PQ_batch
    dev = torch.device("cuda:0")
    B, n, k, J = 4096, 500, 10, 4
    pos = torch.from_numpy(synthetic.ellipsoid_clouds(B, n, seed=457893).reshape(-1, 3)).to(dev)
    gp = torch.arange(B + 1, dtype=torch.int32, device=dev) * n
    b = build_PQ_batch(pos, gp, k, {"kernel": "gaussian", "gaussian_eps": 0.05})
    F = J + 2
    gs, gv = torch.ones(B * n, 3, F, device=dev), torch.ones(3 * B * n, 1, F, device=dev)
    host_x = pos.cpu().pin_memory()
    host_ops = [t.cpu().pin_memory() for t in (b["rowptr"], b["col"], b["pval"], b["blk"])]
    op_bytes = sum(t.numel() * t.element_size() for t in host_ops)

    def step(upload_x, upload_ops):
        if upload_ops:
            for t in host_ops:
                t.to(dev, non_blocking=True)
        x = host_x.to(dev, non_blocking=True) if upload_x else pos
        xs = x.clone().requires_grad_(True)
        xv = x.reshape(-1, 1).clone().requires_grad_(True)
        get_Batch_Wjxs(xs, b["P"], J=J).backward(gs)
        get_Batch_Wjxs(xv, b["Q"], J=J).backward(gv)

    def timed(fn, it=10):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(it):
            fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / it

    res = {"graphs": B, "x_bytes": host_x.numel() * 4, "operator_bytes_int32_csr_bsr3": op_bytes}
    for name, ux, uo in (("resident", False, False), ("x_over_pcie", True, False), ("x_and_operators_over_pcie", True, True)):
        ms = timed(lambda: step(ux, uo))
        res[name] = {"ms_per_step": round(ms, 3), "graphs_per_s": round(B / ms * 1e3)}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
