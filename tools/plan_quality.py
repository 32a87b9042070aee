#!/usr/bin/env python3
"""modelled LDS bank-slot multiplicity (cycles per 16-lane ds_read_b128 group, 1 = conflict-free) of the SELL slot plan
over whole parts of a cfg2-shaped batch: natural entry order vs the pack's plan (tests/test_gpu_parity.py holds the
same model for one slice)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from esc_gnn_amd import ScatterOperator, synthetic

G0 = np.array([0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27]); G1 = np.setdiff1d(np.arange(32), G0)
def cycles(tab):
    tot = cnt = 0
    for grp in (G0, G1, G0 + 32, G1 + 32):
        for row in tab[:, grp]:
            u = np.unique(row); tot += np.bincount(u & 15, minlength=16).max(); cnt += 1
    return tot, cnt

dev = torch.device("cuda:0")
b = synthetic.ellipsoid_batch(8, 500, 10, seed=21)
t = lambda a: torch.from_numpy(a).to(dev)
form = ScatterOperator.from_bsr3(t(b["rowptr"]), t(b["col"]), t(b["blk"]), b["N"]).bsr3()
sell = form.sell(keep_plan=True)
slot = sell.slot_of_e.cpu().numpy(); lrow = sell.lc_of_e.cpu().numpy(); pos = sell.pos_of_row.cpu().numpy().astype(np.int64)
rp, col, pp = b["rowptr"].astype(np.int64), b["col"].astype(np.int64), form.part_ptr.cpu().numpy().astype(np.int64)
perm = sell.perm.cpu().numpy(); doff = sell.slice_doff.cpu().numpy()
nsl = perm.shape[0] // 64 // form.n_parts
tn = cn = tp = cp = 0
by_col = {}
for s in range(perm.shape[0] // 64):
    part = s // nsl; gs = pp[part]; rows0 = int(pp[part + 1] - gs); w = int(doff[s + 1] - doff[s])
    if w == 0: continue
    plan = np.full((w, 64), rows0, dtype=np.int64); nat = plan.copy()
    for lane, r in enumerate(perm[s * 64:(s + 1) * 64]):
        if r < 0: continue
        r += gs
        for d, e in enumerate(range(rp[r], min(rp[r + 1], rp[r] + sell.dmax))):
            nat[d, lane] = pos[col[e]]; plan[slot[e], lane] = lrow[e]
    a, c = cycles(nat); tn += a; cn += c
    a, c = cycles(plan); tp += a; cp += c
    for d in range(w):
        a, c = cycles(plan[d:d + 1]); x = by_col.setdefault(d, [0, 0]); x[0] += a; x[1] += c
print(json.dumps({"natural": tn / cn, "planned": tp / cp, "planned_by_slot_column": {d: round(v[0] / v[1], 3) for d, v in sorted(by_col.items())}}))
