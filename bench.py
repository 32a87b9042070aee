#!/usr/bin/env python3
"""
bench.py -- graphs/sec of P+Q wavelet scattering, forward + backward, J=4
(BASELINE.json metric), on synthetic ellipsoid batches.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (default = BASELINE.json configs[1], "cfg2"): per GPU a block-diagonal
batch of 4096 graphs x 500 nodes, symmetrised kNN k=10, scalar track x = pos
(C=3) on P (CSR), vector track x = pos on Q (BSR-3), dyadic J=4 + low-pass
(T=16 products, F=6 filters).  One "step" = one pass of the hot path over that
batch: P-chain fwd, Q-chain fwd, then both backward w.r.t. x for a fixed upstream
gradient (what autograd of the reference's get_Batch_Wjxs does).  Operators are
canonical (int32 CSR / BSR-3) and resident in HBM before the timed region.
Since round 3 the step goes through the both-tracks call (esc_gnn_amd.get_Batch_Wjxs_both:
ESCGNNModular.forward's two _scatter calls, models/escgnn_modular.py:674, :685, as one call):
when P pairs with the factored Q -- it does for builder-shaped operators like these -- the two
chains of a direction are ONE launch (`route_step`, kernels_ms = both_fwd / both_bwd); otherwise,
and in the "four_launches" variant, one chain per track and direction as before.

Multi-GPU: graphs are independent diagonal blocks, so ranks take disjoint
batches of the same size and no collective is on the data path ("weak" scaling);
value = graphs all ranks processed / max-over-ranks time.

Prints ONE JSON line (rank 0) with the contract's fields plus
  roofline      dominant kernel (both-tracks forward; the Q-chain forward when the tracks do not pair): HBM bytes per
                launch (measured, profiles/hbm_traffic.json) / its mean HIP-event duration inside the timed region,
                against 8 TB/s HBM peak; SURVEY.md §8d's algorithmic figure beside it (see `regime`).
  cpu_baseline  the CPU oracle's torch.sparse.mm restatement of the reference
                (kind "port"), timed on a bounded sample of the same batch.
  route_Q       which kernel family the drop-in chose by itself for THIS Q (round 3): the blocks of the synthetic batch are
                builder-shaped (p_ij O_i S_ij O_j^T in fp32, synthetic.q_blocks_from_frames = process_pyg_data.py:306-313,
                :492), so the ingest recovers the factored form from the blocks (escat_recover_factored, verified to
                1e-6 x max|Q|) and the vector chain runs the factored kernel.  `value` is that default behaviour.
  variants      (N = 1 only, after the timed region) the same step on the same inputs with the routing forced:
                "bsr3_fp32" = the BSR-3 register kernel (round 2's headline path: what an operator without the structure
                gets), "fp16_stored_values" = the same blocks rounded to fp16 as the reference's ellipsoid pipeline
                stores them (half-valued pack, bit-identical to bsr3 on those values), "factored_from_frames" = the
                factored operator built from the true frames (the builder's own route).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.3 TB/s achievable copy)
LDS_PEAK_GBS = 150000.0   # MI355X_MICROARCH.md, LDS: ~150 TB/s aggregate for ds_read_b64/b128 with every CU streaming


def alg_bytes_scalar(N, nnz, C, T, F):
    return T * (4 * (N + 1) + 8 * nnz) + 4 * N * C + 4 * N * C * F


def alg_bytes_vector(N, nnz, T, F):
    return T * (4 * (N + 1) + 40 * nnz) + 12 * N + 12 * N * F


def compulsory_bytes_vector(N, nnz, F):
    """operator read once (packed copy: 36 B values + 2 B column per block), x in, bank out."""
    return 38 * nnz + 4 * (N + 1) + 12 * N + 12 * N * F


def lib_sha256():
    """hash of the built product library: the measured-traffic summary is only valid for the kernels it was taken with."""
    import hashlib
    h = hashlib.sha256()
    with open(os.path.join(ROOT, "esc_gnn_amd", "libescat.so"), "rb") as f:
        for chunk in iter(lambda: f.read(1 << 20), b""):
            h.update(chunk)
    return h.hexdigest()


def measured_traffic(args, kernel_key, kernel_end=""):
    """HBM bytes per launch from the committed PMC summary (profiles/hbm_traffic.json) -- only if it was taken on this
    workload AND with this very libescat.so (sha256 stamp written by tools/pmc_to_traffic.py); otherwise None."""
    try:
        t = json.load(open(os.path.join(ROOT, "profiles", "hbm_traffic.json")))
    except Exception:
        return None
    c = t.get("config", {})
    if (c.get("graphs"), c.get("nodes"), c.get("k"), c.get("J")) != (args.graphs, args.nodes, args.k, args.J):
        return None
    if t.get("libescat_sha256") != lib_sha256():
        return None
    for name, v in t.get("kernels", {}).items():
        if name.startswith(kernel_key) and name.endswith(kernel_end):
            return v.get("hbm_bytes_per_launch_corrected")
    return None


def cpu_baseline(b, n_graphs, n_per_graph, J, threads):
    """reference-equivalent CPU path (oracle/chain_oracle.ref_get_Batch_Wjxs: torch.sparse.mm on the
    un-coalesced int64 COO operand, as models/escgnn_data_classes.py:137-142 builds it), fwd+bwd of both tracks."""
    import numpy as np
    import torch
    from oracle import chain_oracle as co
    N = n_graphs * n_per_graph
    e1 = int(b["rowptr"][N])
    rows = np.repeat(np.arange(N), np.diff(b["rowptr"][:N + 1])).astype(np.int64)
    cols = b["col"][:e1].astype(np.int64)
    P = torch.sparse_coo_tensor(torch.from_numpy(np.stack([rows, cols])), torch.from_numpy(b["pval"][:e1]), (N, N),
                                dtype=torch.float32)
    a = np.arange(3)
    qr = (3 * rows)[:, None, None] + a[None, :, None] + 0 * a[None, None, :]
    qc = (3 * cols)[:, None, None] + 0 * a[None, :, None] + a[None, None, :]
    Q = torch.sparse_coo_tensor(torch.from_numpy(np.stack([qr.reshape(-1), qc.reshape(-1)])),
                                torch.from_numpy(b["blk"][:e1].reshape(-1)), (3 * N, 3 * N), dtype=torch.float32)
    torch.set_num_threads(threads)
    xs = torch.from_numpy(b["pos"][:N]).clone().requires_grad_(True)
    xv = torch.from_numpy(b["pos"][:N]).reshape(-1, 1).clone().requires_grad_(True)
    t0 = time.perf_counter()
    Ws = co.ref_get_Batch_Wjxs(xs, P, scales_type="dyadic", J=J)
    Wv = co.ref_get_Batch_Wjxs(xv, Q, scales_type="dyadic", J=J)
    Ws.backward(torch.ones_like(Ws))
    Wv.backward(torch.ones_like(Wv))
    dt = time.perf_counter() - t0
    return n_graphs / dt, dt


def bench_cfg3(c, dev, t, args, J=6, reps=3):
    """secondary, N = 1 only: BASELINE configs[2] -- one cloud (default 2 M points, k = 16), vector track Q given as its
    BSR-3 blocks, dyadic J = 6 (T = 64).  The graph does not fit LDS: one step kernel per power, the operator streamed from
    HBM every step, rows relabelled breadth-first once at ingest.  Two chains on the same operator, HIP-event times forward
    and backward:
      * on the blocks (k_bsr3_step_staged; ESCAT_CHAIN_PATH=multi pins the route): the top-level keys and `roofline`
        (SURVEY 8d's per-step bytes 4(N+1) + 40 nnz + 24 N) -- comparable with rounds 1-2;
      * `factored`: what the drop-in takes by itself since round 3 -- the frames recovered from the blocks across the
        whole device (esc_gnn_amd/recover_large.py, timed), every block checked, then rotation + the 3-channel chain on the
        signed scalar operator (8 B per nonzero and step) + rotation back."""
    import torch
    from esc_gnn_amd import get_Batch_Wjxs, ScatterOperator, plan_chain
    none = (torch.zeros(1, dtype=torch.int32, device=dev), 0, 0)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    op = ScatterOperator.from_bsr3(t(c["rowptr"]), t(c["col"]), t(c["blk"]), c["N"], parts=none)
    op.bsr3_t()
    torch.cuda.synchronize(); ingest_s = time.perf_counter() - t0
    plan = plan_chain(scales_type="dyadic", J=J)
    n, nnzb, T, F = c["N"], c["nnz"], plan.T, plan.F
    xv = t(c["pos"]).reshape(-1, 1).clone().requires_grad_(True)
    g = torch.randn(3 * n, 1, F, device=dev)

    def chain_times():
        tf = tb = 0.0
        for r in range(reps + 1):
            e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
            xv.grad = None
            e[0].record(); W = get_Batch_Wjxs(xv, op, scales_type="dyadic", J=J); e[1].record(); W.backward(g); e[2].record()
            torch.cuda.synchronize()
            if r >= 1:
                tf += e[0].elapsed_time(e[1]) / reps; tb += e[1].elapsed_time(e[2]) / reps
        return tf, tb, W.detach()
    prev = os.environ.get("ESCAT_CHAIN_PATH")
    os.environ["ESCAT_CHAIN_PATH"] = "multi"          # the blocks themselves (the route a forced chain path keeps)
    try:
        tf, tb, W = chain_times()
    finally:
        if prev is None:
            del os.environ["ESCAT_CHAIN_PATH"]
        else:
            os.environ["ESCAT_CHAIN_PATH"] = prev
    tele = float((W.sum(-1).reshape(-1) - xv.detach().reshape(-1)).abs().max())
    # the default route: frames from the blocks alone, once per operator
    torch.cuda.synchronize(); t0 = time.perf_counter()
    fop = op.factored()
    torch.cuda.synchronize(); recover_s = time.perf_counter() - t0
    fac = {"route": "blocks (no factored form found)", "recover_s": recover_s, "factor_err_rel": op.factor_err}
    if fop is not None:
        ff, fb, Wf = chain_times()
        fstep = 4 * (n + 1) + 8 * nnzb + 24 * n
        fac = {"route": "factored form recovered from the blocks alone (recover_large: tree by pointer doubling, gauge from the "
                        "holonomies, level walk; every block checked) -> rotation, 3-channel chain on the signed scalar operator "
                        "(k_csr_step, |value| + sign bits), rotation back",
               "recover_s": recover_s, "factor_err_rel": op.factor_err, "fwd_ms": ff, "bwd_ms": fb, "fwd_ms_per_step": ff / T,
               "chains_per_s_fwd_bwd": 1e3 / (ff + fb), "speedup_vs_blocks": (tf + tb) / (ff + fb),
               "max_rel_diff_vs_block_chain": float((Wf - W).abs().max() / W.abs().max()),
               "bytes_per_step": fstep, "achieved_GBs": fstep / (ff / T * 1e-3) / 1e9,
               "frac_of_hbm_peak": fstep / (ff / T * 1e-3) / 1e9 / HBM_PEAK_GBS,
               "note": "the step is bound by the gather of x rows out of L2 (12 B per nonzero), not by the 8-B operator stream"}
        del Wf
    del W, g
    # the same cloud's operators built by the product's own builder (grid kNN + P + local-PCA frames + Q blocks, all on
    # the GPU), timed, its pattern checked against the host kd-tree's that fed the chains above.  The builder knows the
    # frames its blocks were made from (ScatterOperator.hint_frames): its Q needs the check of every block only.
    from esc_gnn_amd.process_pyg_data import build_PQ_batch
    gp = torch.tensor([0, n], dtype=torch.int32)
    pos_d = t(c["pos"])
    build_s = []
    for r in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        bb = build_PQ_batch(pos_d, gp, 16, {'kernel': 'epanechnikov', 'r_cut': 5.0})
        torch.cuda.synchronize(); build_s.append(time.perf_counter() - t0)
    same = bool(torch.equal(bb["rowptr"].cpu(), torch.from_numpy(c["rowptr"])) and torch.equal(bb["col"].cpu(), torch.from_numpy(c["col"])))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    hinted = bb["Q"].factored() is not None
    torch.cuda.synchronize(); check_s = time.perf_counter() - t0
    hint = {"factored_form_from_the_builders_frames": hinted, "check_s": check_s, "factor_err_rel": bb["Q"].factor_err}
    del bb
    step_bytes = 4 * (n + 1) + 40 * nnzb + 24 * n
    return {"workload": f"cfg3: one ellipsoid cloud, {n} points, kNN k=16, Q (BSR-3) dyadic J={J} (T={T}, F={F}), fwd and bwd "
                        f"chains; multi-launch path (k_bsr3_step_staged), breadth-first relabelling at ingest",
            "nodes": n, "nnz_blocks": nnzb, "fwd_ms": tf, "bwd_ms": tb, "fwd_ms_per_step": tf / T,
            "chains_per_s_fwd_bwd": 1e3 / (tf + tb),
            "roofline": {"bound": "hbm", "kernel": "k_bsr3_step_staged (one diffusion step)", "bytes_per_step": step_bytes,
                         "achieved": step_bytes / (tf / T * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": step_bytes / (tf / T * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "note": "algorithmic bytes of one step (operator + x in + y out) / (forward chain time / T)"},
            "telescoping_max_abs_err": tele, "host_knn_s": c["gen_s"], "ingest_and_relabel_s": ingest_s,
            "gpu_build": {"what": "escat_build_knn_P_grid + frames + Q blocks + operator ingest of the same cloud (process_pyg_data's "
                                  "builder on the GPU)", "seconds": build_s[1], "seconds_first_call": build_s[0],
                          "pattern_equal_to_host_kdtree": same, "frames_hint": hint},
            "factored": fac}


def bench_fresh_coo(b, dev, t, args, xs, xv, reps=5):
    """secondary, N = 1 only: the FUNCTION-level drop-in on what the reference's DataLoader hands over -- fresh torch sparse
    COO tensors every step (int64 indices, flagged un-coalesced; P row-major sorted, Q in block-emission order:
    models/escgnn_data_classes.py:137-142, process_pyg_data.py:839-857), forward only, both tracks, no gradient (how the
    reference trains: the scattering sits outside the autograd graph).  COO in -> bank out, everything included: ingest,
    parts, packs, routing, the two chains.  Under the three ingest policies (esc_gnn_amd.operators.INGEST_POLICY)."""
    import torch
    from esc_gnn_amd import get_Batch_Wjxs, operators as _ops
    N = b["N"]
    rp = t(b["rowptr"]).long()
    rows = torch.repeat_interleave(torch.arange(N, device=dev), rp[1:] - rp[:-1])
    cols = t(b["col"]).long()
    Pi = torch.stack([rows, cols]).contiguous()
    a = torch.arange(3, device=dev)
    Qi = torch.stack([((3 * rows)[:, None, None] + a[None, :, None] + 0 * a[None, None, :]).reshape(-1),
                      ((3 * cols)[:, None, None] + 0 * a[None, :, None] + a[None, None, :]).reshape(-1)]).contiguous()
    Pv, Qv = t(b["pval"]), t(b["blk"]).reshape(-1)
    del rows, cols, rp
    res = {}
    for policy in ("auto", "full", "single_use"):
        _ops.INGEST_POLICY = policy
        ts = []
        for r in range(reps + 1):
            # a DataLoader's next batch has other graphs: nothing of the previous batch's plan or parts may be reused
            # (here the pattern repeats, and the library would recognise it by content)
            _ops._PLAN_CACHE.clear(); _ops._PARTS_CACHE.clear()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            P = torch.sparse_coo_tensor(Pi, Pv, (N, N), check_invariants=False)             # fresh tensor objects: no cache hit
            Q = torch.sparse_coo_tensor(Qi, Qv, (3 * N, 3 * N), check_invariants=False)
            with torch.no_grad():
                Ws = get_Batch_Wjxs(xs, P, scales_type="dyadic", J=args.J)
                Wv = get_Batch_Wjxs(xv, Q, scales_type="dyadic", J=args.J)
            torch.cuda.synchronize()
            if r >= 1:
                ts.append((time.perf_counter() - t0) * 1e3)
            del P, Q, Ws, Wv
        ts.sort()
        res[policy] = {"ms_per_step": ts[len(ts) // 2], "min_ms": ts[0], "max_ms": ts[-1]}
    _ops.INGEST_POLICY = "auto"
    res["what"] = ("fresh int64 COO P and Q every step -> get_Batch_Wjxs forward for both tracks (no gradient), wall clock with a "
                   "synchronise at both ends, median of %d; 'auto' = the default (a tensor object seen for the first time gets "
                   "unplanned packs and no recovery, the full treatment from its second use on: every tensor here is new, so "
                   "this is the single-use cost), 'full' = ESCAT_INGEST=full (matching slot plan, factored form recovered from "
                   "Q's blocks, at once), 'single_use' = ESCAT_INGEST=single_use (never)" % reps)
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--graphs", type=int, default=4096, help="graphs per GPU (cfg2: 4096)")
    ap.add_argument("--nodes", type=int, default=500)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--J", type=int, default=4)
    ap.add_argument("--cpu-sample-graphs", type=int, default=128)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-steady-state", action="store_true", help="skip the settled-device measurement (260 more steps: profiling runs)")
    ap.add_argument("--no-fresh-coo", action="store_true", help="skip the function-level drop-in measurement on fresh COO tensors")
    ap.add_argument("--no-cfg3", action="store_true", help="skip the secondary cfg3 measurement (one large cloud, N = 1 only)")
    ap.add_argument("--cfg3-nodes", type=int, default=2_000_000)
    ap.add_argument("--gen-workers", type=int, default=-1,
                    help="processes for input generation (-1: up to 16; use 1 under rocprofv3: no fork after GPU init)")
    args = ap.parse_args()

    rehearsal = os.environ.get("ESCAT_BENCH_REHEARSAL", "") == "1"
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher around us: start the N ranks ourselves, as child processes, BEFORE anything here touches the GPU
        # (torch.cuda.device_count() does not initialise it), and leave with the launcher's exit code
        import subprocess
        import socket
        import torch
        have = torch.cuda.device_count()
        if have < args.gpus and not rehearsal:
            print(f"bench.py: --gpus {args.gpus} but only {have} GPU(s) visible", file=sys.stderr)
            raise SystemExit(2)
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.run(cmd).returncode)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
        raise SystemExit(2)

    # ---- inputs (host, before the GPU is touched: the generator forks workers)
    from esc_gnn_amd import synthetic
    workers = min(16, os.cpu_count() or 1) if args.gen_workers < 0 else args.gen_workers
    b = synthetic.ellipsoid_batch(args.graphs, args.nodes, args.k, seed=457892 + 1 + rank, workers=workers)
    N, nnz = b["N"], b["nnz"]
    c3 = None
    if world == 1 and not args.no_cfg3:
        # secondary workload (BASELINE configs[2]): one connected cloud, kNN k=16 on the host -- like the batch above it
        # is generated BEFORE the GPU is touched (the kNN forks workers)
        tg0 = time.perf_counter()
        c3 = synthetic.single_cloud(args.cfg3_nodes, 16, workers=workers)
        c3["gen_s"] = time.perf_counter() - tg0

    import torch
    import torch.distributed as dist
    from esc_gnn_amd import get_Batch_Wjxs, ScatterOperator, plan_chain, load_library
    load_library()                      # raises if libescat.so is missing: no fallback
    # ESCAT_BENCH_REHEARSAL=1: exercise the N > 1 code path on a box with fewer GPUs than ranks (ranks share devices,
    # gloo instead of RCCL).  For checking the script, not for numbers.
    if torch.cuda.device_count() < (1 if rehearsal else (local_rank + 1 if world > 1 else 1)):
        print(f"bench.py: rank {rank} needs cuda:{local_rank}, {torch.cuda.device_count()} GPU(s) visible", file=sys.stderr)
        raise SystemExit(2)
    dev_index = local_rank % max(torch.cuda.device_count(), 1) if rehearsal else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    t = lambda a: torch.from_numpy(a).to(dev)
    rowptr, col = t(b["rowptr"]), t(b["col"])
    pval_d, blk_d = t(b["pval"]), t(b["blk"])
    def canonicalise():
        # row ranges, transposes for the backward walk, SELL packs + slot plans: once per batch, outside the timed
        # region, reported separately
        torch.cuda.synchronize(); tc0 = time.perf_counter()
        p_ = ScatterOperator.from_csr(rowptr, col, pval_d, N)
        q_ = ScatterOperator.from_bsr3(rowptr, col, blk_d, N)
        for form in (p_.csr(), p_.csr_t()):
            form.sell()
        qf = q_.factored()                   # round 3: the factored form recovered from the blocks, if they have one
        if qf is not None and qf.fits_register_path(plan_chain(scales_type="dyadic", J=args.J).F):
            qf.signed.csr().sell(); qf.signed.csr_t().sell()
        else:
            q_.bsr3().sell(); q_.bsr3_t().sell()
        from esc_gnn_amd import BothTracksOperator
        bo = BothTracksOperator.pair(p_, q_)       # round 3: P and the factored Q as ONE operator (both tracks in one launch)
        if bo is not None and bo.fits(plan_chain(scales_type="dyadic", J=args.J)):
            bo.fop.signed.csr().sell(); bo.fop.signed.csr_t().sell()
        torch.cuda.synchronize()
        return p_, q_, (time.perf_counter() - tc0) * 1e3
    _, _, canon_first_ms = canonicalise()          # includes loading the library's code objects and torch's first calls
    opP, opQ, canon_ms = canonicalise()

    # measured copy ceiling of this GPU (1 GiB device-to-device copy: read + write bytes / time)
    src = torch.empty(1 << 28, dtype=torch.float32, device=dev); dst = torch.empty_like(src)
    for _ in range(2):
        dst.copy_(src)
    ce = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ce[0].record()
    for _ in range(5):
        dst.copy_(src)
    ce[1].record(); torch.cuda.synchronize()
    copy_gbs = 5 * 2 * src.numel() * 4 / (ce[0].elapsed_time(ce[1]) * 1e-3) / 1e9
    del src, dst
    plan = plan_chain(scales_type="dyadic", J=args.J)
    T, F, C = plan.T, plan.F, 3
    xs = t(b["pos"]).clone().requires_grad_(True)
    xv = t(b["pos"]).reshape(-1, 1).clone().requires_grad_(True)
    gs = torch.randn(N, C, F, device=dev)
    gv = torch.randn(3 * N, 1, F, device=dev)

    names = ("P_fwd", "Q_fwd", "P_bwd", "Q_bwd")
    evs = []

    def step(record, opQ_=None, evs=evs):
        """the step as four launches: one chain per track and direction"""
        opQ_ = opQ if opQ_ is None else opQ_
        e = [torch.cuda.Event(enable_timing=True) for _ in range(5)] if record else None
        xs.grad = None; xv.grad = None
        if record: e[0].record()
        Ws = get_Batch_Wjxs(xs, opP, scales_type="dyadic", J=args.J)
        if record: e[1].record()
        Wv = get_Batch_Wjxs(xv, opQ_, scales_type="dyadic", J=args.J)
        if record: e[2].record()
        Ws.backward(gs)
        if record: e[3].record()
        Wv.backward(gv)
        if record: e[4].record()
        if record: evs.append(e)

    # round 3: both tracks in one launch per direction (esc_gnn_amd/both.py) when P and Q pair -- what the drop-in's
    # both-tracks call takes by itself for this batch
    from esc_gnn_amd import get_Batch_Wjxs_both, BothTracksOperator
    from esc_gnn_amd import both as _both
    bop = BothTracksOperator.pair(opP, opQ) if os.environ.get("ESCAT_BOTH", "auto") != "0" else None
    fused = bop is not None and bop.fits(plan) and _both._fusable(xs, xv, plan)
    names2 = ("both_fwd", "both_bwd")

    def step_both(record, evs=evs):
        e = [torch.cuda.Event(enable_timing=True) for _ in range(3)] if record else None
        xs.grad = None; xv.grad = None
        if record: e[0].record()
        Ws, Wv = get_Batch_Wjxs_both(xs, opP, xv, opQ, scales_type="dyadic", J=args.J)
        if record: e[1].record()
        torch.autograd.backward([Ws, Wv], [gs, gv])
        if record: e[2].record()
        if record: evs.append(e)

    main_step = step_both if fused else step

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def variant(opQ_, note):
        """the same step with another form of the vector operator (N = 1, after the timed region)"""
        ev2 = []
        for _ in range(args.warmup):
            step(False, opQ_)
        torch.cuda.synchronize(); tv0 = time.perf_counter()
        for _ in range(args.steps):
            step(True, opQ_, ev2)
        torch.cuda.synchronize(); dtv = time.perf_counter() - tv0
        perv = {nm: sum(e[i].elapsed_time(e[i + 1]) for e in ev2) / len(ev2) for i, nm in enumerate(names)}
        return {"value": args.graphs * args.steps / dtv, "unit": "graphs/s", "ms_per_step": dtv / args.steps * 1e3,
                "kernels_ms": perv, "note": note}

    for _ in range(args.warmup):
        main_step(False)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        main_step(True)
    barrier()
    dt = time.perf_counter() - t0
    from esc_gnn_amd.dist_utils import max_over_ranks
    dt = max_over_ranks(dt, None if rehearsal else dev)
    # what the collective backend itself saw: every rank adds 1 (RCCL all-reduce on the GPU; gloo in rehearsal)
    ranks_seen, backend = 1, "none"
    if world > 1:
        one = torch.ones(1, dtype=torch.float32, device="cpu" if rehearsal else dev)
        dist.all_reduce(one)
        ranks_seen, backend = int(one.item()), dist.get_backend()
        if ranks_seen != args.gpus or dist.get_world_size() != args.gpus:
            print(f"bench.py: {ranks_seen} ranks answered the all-reduce, --gpus {args.gpus}", file=sys.stderr)
            raise SystemExit(3)

    if rank == 0:
        use = names2 if fused else names
        per = {nm: sum(e[i].elapsed_time(e[i + 1]) for e in evs) / len(evs) for i, nm in enumerate(use)}
        bytes_q = alg_bytes_vector(N, nnz, T, F)
        bytes_p = alg_bytes_scalar(N, nnz, C, T, F)
        comp_q = compulsory_bytes_vector(N, nnz, F)
        from esc_gnn_amd.operators import FactoredVectorOperator as _FVO
        from esc_gnn_amd.pyg_utilities import vector_route
        routed = vector_route(opQ, xv.detach(), F)
        factored_route = isinstance(routed, _FVO)
        half_route = (not factored_route) and opQ.bsr3().sell().half_val is not None
        if fused:                   # both tracks: |p| + LDS row with sign bits per nonzero, lane records, frames (prologue +
            # one re-read per written band pair), both x in, both banks out
            qkey, qend = "k_chain_both<", ", false>"
            qname = "k_chain_both<.., 512, false> (both tracks forward: P and the factored Q in one launch)"
            comp_q = 6 * nnz + 12 * N + 36 * N + 24 * N + 24 * N * F
            dom_ms = per["both_fwd"]
            bytes_dom = bytes_q + bytes_p           # SURVEY 8d's figure for the two chains this launch replaces
            lds_bytes = T * (2 * 16 * nnz + 4 * 16 * N)
        else:
            qend = ", false, false>"
            dom_ms = per["Q_fwd"]
            bytes_dom = bytes_q
            lds_bytes = T * (16 * nnz + 2 * 16 * N)
            if factored_route:      # |p| + LDS row per nonzero, lane records, frames, x in, bank out
                qkey, qname = "k_chain_sell<OpCsrV3", "k_chain_sell<OpCsrV3<..>, 512, false, false> (Q chain forward, factored form)"
                comp_q = 6 * nnz + 12 * N + 36 * N + 12 * N + 12 * N * F
            elif half_route:
                qkey, qname = "k_chain_qh<", "k_chain_qh<.., 512, false, false> (Q chain forward, half-valued pack)"
                comp_q = 20 * nnz + 12 * N + 12 * N + 12 * N * F
            else:
                qkey, qname = "k_chain_sell<OpBsr3", "k_chain_sell<OpBsr3<..>, 512, false, false> (Q chain forward)"
        alg_gbs = bytes_dom / (dom_ms * 1e-3) / 1e9
        traffic = measured_traffic(args, qkey, qend)
        hbm_bytes = traffic if traffic is not None else comp_q
        achieved = hbm_bytes / (dom_ms * 1e-3) / 1e9
        # LDS side of the same kernel: every step gathers 16 B per stored entry and track and stores each node's new vector twice
        lds_gbs = lds_bytes / (dom_ms * 1e-3) / 1e9
        if fused:
            step_bytes = sum((measured_traffic(args, "k_chain_both<", e) or 0) for e in (", false>", ", true>"))
        else:
            step_bytes = sum((measured_traffic(args, k, e) or 0) for k, e in
                             ((qkey, ", false, false>"), (qkey, ", true, false>"),
                              ("k_chain_sell<OpCsr<3", ", false, false>"), ("k_chain_sell<OpCsr<3", ", true, false>")))
        route_q = ("factored form recovered from the blocks at ingest (escat_recover_factored), relative reconstruction "
                   f"error {opQ.factor_err:.2e}" if factored_route else
                   "half-valued BSR-3 pack (every block value exact in fp16)" if half_route else
                   "BSR-3 register kernel (fp32 pack)" + (f"; recovery refused, error {opQ.factor_err:.2e}" if opQ.factor_err is not None else ""))
        out = {
            "metric": "graphs/sec for P+Q scattering fwd+bwd (J=4)",
            "value": args.graphs * world * args.steps / dt,
            "unit": "graphs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"cfg2: {args.graphs} ellipsoid graphs x {args.nodes} nodes per GPU, kNN k={args.k}, "
                                   f"P (CSR, C=3) + Q (BSR-3) dyadic J={args.J} + lowpass, fwd+bwd",
                       "graphs_per_gpu": args.graphs, "nodes_per_graph": args.nodes, "k": args.k, "J": args.J,
                       "N": N, "nnz_P": nnz, "T": T, "F": F, "parallelism": f"dp{world} (graph shards, no data-path collective)"},
            "collective_backend": backend, "ranks_seen_by_allreduce": ranks_seen,
            "route_Q": route_q,
            "route_step": ("both tracks in one launch per direction (escat_chain_both_f32 / _bwd_f32): P pairs with the factored Q "
                           f"(|Pt| = P to {bop.mag_err:.1e} of max P); get_Batch_Wjxs_both takes it by itself" if fused else
                           "one chain per track and direction (four launches)"),
            "roofline": {"bound": "hbm", "kernel": qname,
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic,
                         "bytes_basis": "measured (rocprofv3 FETCH_SIZE/WRITE_SIZE, profiles/hbm_traffic.json, same libescat.so)"
                                        if traffic is not None else
                                        "compulsory bytes (operator read once + x in + bank out): no PMC summary for this build",
                         "ms_per_launch": dom_ms,
                         "compulsory_bytes_per_launch": comp_q,
                         "frac_of_measured_copy": achieved / copy_gbs,
                         "algorithmic_bytes_per_launch": bytes_dom, "algorithmic_GBs": alg_gbs,
                         "whole_step_hbm_frac": (step_bytes / (dt / args.steps) / 1e9 / HBM_PEAK_GBS) if step_bytes and traffic else None,
                         "lds": {"bytes": lds_bytes, "achieved": lds_gbs, "peak": LDS_PEAK_GBS, "unit": "GB/s",
                                 "frac": lds_gbs / LDS_PEAK_GBS,
                                 "note": "T*(16 B gathered per stored entry and track + two 16-B vector stores per node and track) vs ~150 TB/s aggregate "
                                         "ds_read_b128 rate (MI355X_MICROARCH.md, LDS)"},
                         "regime": "the kernel keeps each graph's operator in registers for all T steps: HBM carries the "
                                   "operator once per chain (achieved/frac = those bytes / time vs the 8 TB/s peak); "
                                   "algorithmic_GBs is SURVEY.md 8d's figure (BSR-3 operator re-streamed every step) for "
                                   "reference only and is not a bandwidth: the kernels move far fewer bytes than that model "
                                   "(registers instead of re-streaming; 6 B instead of 40 B per nonzero in the factored form), "
                                   "so the time is bounded by LDS gathers and fp32 FMAs, not by HBM -- see roofline.lds and DESIGN.md 3."},
            "hbm_copy_measured_GBs": copy_gbs,
            "canonicalise_ms_once_per_batch": canon_ms,
            "canonicalise_ms_first_call_in_process": canon_first_ms,
            "kernels_ms": per,
            "kernels_alg_GBs": ({"both_fwd": alg_gbs, "both_bwd": (bytes_q + bytes_p) / (per["both_bwd"] * 1e-3) / 1e9} if fused else
                                {"P_fwd": bytes_p / (per["P_fwd"] * 1e-3) / 1e9, "Q_fwd": alg_gbs,
                                 "P_bwd": bytes_p / (per["P_bwd"] * 1e-3) / 1e9,
                                 "Q_bwd": bytes_q / (per["Q_bwd"] * 1e-3) / 1e9}),
        }
        if world == 1:
            # NOT the headline: the same step once the device has settled.  After an idle spell (or after other kernels) an
            # MI355X takes ~40 of these steps (30 ms) to reach its steady clock for this kernel -- per-step times 0.74 ->
            # 0.84 -> 0.68 ms, profiles/r3c_step_time_after_idle.json, tools/host_overhead.py -- so a window of K = 20 steps
            # after W = 5 sits inside that ramp; a training loop runs in the settled state.
            if not args.no_steady_state:
                for _ in range(60):
                    main_step(False)
                torch.cuda.synchronize(); ts0 = time.perf_counter()
                for _ in range(200):
                    main_step(False)
                torch.cuda.synchronize(); dts = time.perf_counter() - ts0
                out["steady_state"] = {"value": args.graphs * 200 / dts, "unit": "graphs/s", "ms_per_step": dts / 200 * 1e3, "steps": 200,
                                       "untimed_steps_before": 60 + args.warmup + args.steps,
                                       "note": "the same step after the device's power management has settled on this kernel (the "
                                               "headline window of --steps after --warmup lies inside the ~40-step ramp when both are small)"}
            # secondary, NOT the headline: the same step with the routing of the vector operator forced (same inputs)
            from esc_gnn_amd import FactoredVectorOperator, operators as _ops
            out["variants"] = {}
            if fused:
                out["variants"]["four_launches"] = variant(opQ, "the same operators, one chain per track and direction (the routing of the "
                                                                "vector operator as the drop-in picks it: " + route_q.split(" (")[0] + "): "
                                                                "round 3's headline before the both-tracks launch")
            _ops.FACTOR_MODE = "0"
            opQb = ScatterOperator.from_bsr3(rowptr, col, blk_d, N)
            opQb.bsr3().sell(); opQb.bsr3_t().sell()
            out["variants"]["bsr3_fp32"] = variant(opQb, "BSR-3 register kernel on the fp32 blocks (k_chain_sell<OpBsr3>): what an "
                                                         "operator without the factored structure runs; round 2's headline path")
            del opQb
            blk16 = blk_d.to(torch.float16).to(torch.float32)
            opQh = ScatterOperator.from_bsr3(rowptr, col, blk16, N)
            opQh.bsr3().sell(); opQh.bsr3_t().sell()
            assert opQh.bsr3().sell().half_val is not None
            out["variants"]["fp16_stored_values"] = variant(opQh, "the same blocks rounded to fp16 as the reference's ellipsoid HDF5 stores "
                                                                  "them (process_pyg_data.py:676): half-valued pack, k_chain_qh, two workgroups "
                                                                  "per CU, bit-identical to the BSR-3 kernel on these values")
            del opQh, blk16
            _ops.FACTOR_MODE = "auto"
            opQf = FactoredVectorOperator.from_frames(rowptr, col, pval_d, t(b["frames"]))
            opQf.signed.csr().sell(); opQf.signed.csr_t().sell()
            out["variants"]["factored_from_frames"] = variant(opQf, "factored operator built from the true frames (the builder's own route, "
                                                                    "FactoredVectorOperator.from_frames)")
            del opQf
        if world == 1 and not args.no_fresh_coo:
            out["dropin_fresh_coo"] = bench_fresh_coo(b, dev, t, args, xs.detach(), xv.detach())
        if c3 is not None:
            out["cfg3"] = bench_cfg3(c3, dev, t, args)
        if not args.no_cpu_baseline and world == 1:
            sample = min(args.cpu_sample_graphs, args.graphs)
            gps1, s1 = cpu_baseline(b, sample, args.nodes, args.J, 1)
            many = min(16, os.cpu_count() or 1)      # the box's CPU share for one GPU (os.cpu_count() is the whole host)
            gpsn, sn = cpu_baseline(b, sample, args.nodes, args.J, many)
            out["cpu_baseline"] = {"value": gps1, "unit": "graphs/s", "cores": 1, "kind": "port",
                                   "sample": f"first {sample} graphs of the same batch, fwd+bwd of both tracks, "
                                             f"torch.sparse.mm COO restatement of the reference, {s1:.1f} s",
                                   "value_torch_threads_16": gpsn, "threads_16_seconds": sn,
                                   "note": "torch's CPU COO product is serial per nonzero: more threads do not help (SURVEY 6)",
                                   "host_cpus": os.cpu_count()}
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
