#!/usr/bin/env python3
"""debug: wall-clock stamps of the phases of every part in the persistent chain kernel.
needs `make -C esc_gnn_amd/csrc stamps` (libescat_stamps.so, built with -DESCAT_STAMPS).
usage: python tools/stamps.py [P|Q] [fwd|bwd]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from esc_gnn_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), "libescat_stamps.so")
from esc_gnn_amd import synthetic
half = "--half" in sys.argv          # Q with fp16-exact blocks: the half-valued pack (escat_sell_bsr3h.hip)
sys.argv = [a for a in sys.argv if a != "--half"]
which = sys.argv[1] if len(sys.argv) > 1 else "Q"
bwd = len(sys.argv) > 2 and sys.argv[2] == "bwd"
G = 4096
b = synthetic.ellipsoid_batch(G, 500, 10, seed=457893, workers=8)
if half:
    b["blk"] = b["blk"].astype(np.float16).astype(np.float32)
import torch
from esc_gnn_amd import get_Batch_Wjxs, ScatterOperator
dev = torch.device("cuda:0")
t = lambda a: torch.from_numpy(a).to(dev)
if which == "Q":
    op = ScatterOperator.from_bsr3(t(b["rowptr"]), t(b["col"]), t(b["blk"]), b["N"]); x = t(b["pos"]).reshape(-1, 1).clone()
else:
    op = ScatterOperator.from_csr(t(b["rowptr"]), t(b["col"]), t(b["pval"]), b["N"]); x = t(b["pos"]).clone()
_lib.load()
raw = ctypes.CDLL(_lib.LIB_PATH)
us = lambda a: a / 100.0      # wall_clock64: 100 MHz
for J in (0, 4):
    def run():
        xx = x.clone().requires_grad_(bwd)
        W = get_Batch_Wjxs(xx, op, J=J)
        if bwd:
            g = torch.ones_like(W)
            torch.cuda.synchronize()
            return lambda: W.backward(g)
        return None
    for _ in range(2):
        f = run(); f and f()
    st = torch.zeros(G * 16, dtype=torch.int64, device=dev)
    f = run()
    raw.escat_debug_set_stamps(ctypes.c_void_p(st.data_ptr()))
    if f: f()
    else: get_Batch_Wjxs(x, op, J=J)
    torch.cuda.synchronize()
    raw.escat_debug_set_stamps(ctypes.c_void_p(0))
    s = st.cpu().numpy().reshape(G, 16).astype(np.int64)
    names = ["operator loads issued", "previous epilogue issued", "input wait + barrier", "prologue", "step 1 (waits for the operator)",
             "steps 2..T", "post-step wait"]
    if half:    # k_chain_qh: 0 part start, 1 lane records arrived, 2 input DMA + previous epilogue + operator loads issued, ...
        names = ["lane records (HBM round trip)", "DMA + prev epilogue + op loads issued", "input wait + barrier", "prologue + operator wait",
                 "step 1", "steps 2..T", "-"]
    print(f"== {which} {'bwd' if bwd else 'fwd'} J={J}: span {us(s[:, 6].max() - s[:, 0].min()):.1f} us")
    for k, nm in enumerate(names):
        a, bb = s[:, k], s[:, k + 1]
        ok = (a > 0) & (bb > 0)
        d = us(bb[ok] - a[ok])
        print(f"   {nm:34s} +{d.mean():7.2f} us (p10 {np.percentile(d, 10):6.2f}, p90 {np.percentile(d, 90):6.2f})")
    if J > 0:
        for nm, o in (("wave 0 (longest slice)", 8), ("last wave (shortest slice)", 12)):
            ok = (s[:, o] > 0) & (s[:, o + 3] > 0)
            g, e, bq = us(s[ok, o + 1] - s[ok, o]), us(s[ok, o + 2] - s[ok, o + 1]), us(s[ok, o + 3] - s[ok, o + 2])
            print(f"   step 5, {nm:28s}: gather {g.mean():5.2f}  events+store {e.mean():5.2f}  barrier wait {bq.mean():5.2f} us")
        ok = (s[:, 8] > 0) & (s[:, 12] > 0)
        print(f"   step 5 start skew last-first wave: {us(s[ok, 12] - s[ok, 8]).mean():5.2f} us")
    if half:
        ok = (s[:, 8] > 0) & (s[:, 9] > 0)
        print(f"   1 -> op loads issued {us(s[ok, 8] - s[ok, 1]).mean():6.2f} us; first-part input wait {us(s[ok, 9] - s[ok, 8]).mean():6.2f}; epilogue {us(s[ok, 2] - s[ok, 9]).mean():6.2f} us")
    if "--timeline" in os.environ.get("STAMPS_OPTS", "") and J > 0:
        # how many workgroups are inside their step phase (stamp 4 -> 6) at each microsecond of the launch: lockstep shows
        # as a count that swings between 0 and all of them
        t0 = s[:, 0][s[:, 0] > 0].min()
        a = ((s[:, 4] - t0) // 100).astype(np.int64); bq = ((s[:, 6] - t0) // 100).astype(np.int64)
        ok = (s[:, 4] > 0) & (s[:, 6] > 0)
        n = int(bq[ok].max()) + 2
        cnt = np.zeros(n, dtype=np.int64)
        np.add.at(cnt, a[ok], 1); np.add.at(cnt, bq[ok] + 1, -1)
        cnt = np.cumsum(cnt)
        print("   workgroups in their step phase, per us:", " ".join(str(int(v)) for v in cnt[:-1]))
    ok = s[:, 7] > 0
    print(f"   part total (0 -> 7)                {us(s[ok, 7] - s[ok, 0]).mean():7.2f} us")
