#!/usr/bin/env python3
"""host time to ENQUEUE one both-tracks step (forward call + autograd backward) against the GPU time of the step: how much
slack the launch path has before a busy host (8 ranks on one node) would make the step launch-bound."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    graphs = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    from esc_gnn_amd import synthetic
    b = synthetic.ellipsoid_batch(graphs, 500, 10, seed=457893, workers=16)
    import torch
    from esc_gnn_amd import ScatterOperator, get_Batch_Wjxs_both
    dev = torch.device("cuda:0")
    t = lambda a: torch.from_numpy(a).to(dev)
    opP = ScatterOperator.from_csr(t(b["rowptr"]), t(b["col"]), t(b["pval"]), b["N"])
    opQ = ScatterOperator.from_bsr3(t(b["rowptr"]), t(b["col"]), t(b["blk"]), b["N"])
    xs = t(b["pos"]).clone().requires_grad_(True)
    xv = t(b["pos"]).reshape(-1, 1).clone().requires_grad_(True)
    gs = torch.randn(b["N"], 3, 6, device=dev); gv = torch.randn(3 * b["N"], 1, 6, device=dev)

    def step():
        xs.grad = None; xv.grad = None
        Ws, Wv = get_Batch_Wjxs_both(xs, opP, xv, opQ, J=4)
        torch.autograd.backward([Ws, Wv], [gs, gv])
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    K = 200
    t0 = time.perf_counter()
    for _ in range(K):
        step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    res = {"graphs": graphs, "enqueue_ms_per_step": (t1 - t0) / K * 1e3, "total_ms_per_step": (t2 - t0) / K * 1e3}
    # the clock ramp after an idle spell: per-step GPU time of the first 80 steps after 0.5 s of idle
    time.sleep(0.5)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(81)]
    ev[0].record()
    for k in range(80):
        step(); ev[k + 1].record()
    torch.cuda.synchronize()
    res["ms_per_step_after_idle"] = [round(ev[k].elapsed_time(ev[k + 1]), 3) for k in range(80)]
    # the same after 0.5 s of idle followed by N ms of device-to-device copies (does a busy device of any kind do?)
    src = torch.empty(1 << 28, dtype=torch.float32, device=dev); dst = torch.empty_like(src)
    for reps in (25, 75, 150):
        time.sleep(0.5)
        c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        c0.record()
        for _ in range(reps):
            dst.copy_(src)
        c1.record()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(31)]
        ev[0].record()
        for k in range(30):
            step(); ev[k + 1].record()
        torch.cuda.synchronize()
        res[f"after_{reps}_copies"] = {"copy_ms": round(c0.elapsed_time(c1), 1),
                                        "ms_per_step": [round(ev[k].elapsed_time(ev[k + 1]), 3) for k in range(30)]}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
