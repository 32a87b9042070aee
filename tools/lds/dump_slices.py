"""writes the LDS-row tables (slots x 64 lanes) of the first slices of a cfg2-like pack, natural entry order vs planned,
for lds_gather_bench <waves> <file>.  usage: python tools/lds/dump_slices.py <outdir>"""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from esc_gnn_amd import ScatterOperator, synthetic

out = sys.argv[1]; os.makedirs(out, exist_ok=True)
b = synthetic.ellipsoid_batch(2, 500, 10, seed=3)
dev = torch.device("cuda:0")
t = lambda a: torch.from_numpy(a).to(dev)
form = ScatterOperator.from_bsr3(t(b["rowptr"]), t(b["col"]), t(b["blk"]), b["N"]).bsr3()
sell = form.sell()
slot = sell.slot_of_e.cpu().numpy(); lrow = sell.lc_of_e.cpu().numpy()
perm = sell.perm.cpu().numpy(); doff = sell.slice_doff.cpu().numpy()
rp, col = b["rowptr"], b["col"]
FC = 24
nat, plan = [], []
for s in range(8):                      # the slices of part 0 (rows 0..499, gs = 0)
    w = int(doff[s + 1] - doff[s])
    tn = np.full((FC, 64), 500, dtype=np.int32); tp = tn.copy()
    for lane in range(64):
        r = perm[s * 64 + lane]
        if r < 0:
            continue
        for d, e in enumerate(range(rp[r], min(rp[r + 1], rp[r] + sell.dmax))):
            tn[d, lane] = col[e]
            tp[slot[e], lane] = lrow[e]
    nat.append(tn); plan.append(tp)
    print("slice", s, "w", w)
for name, tabs in (("natural", nat), ("planned", plan)):
    with open(os.path.join(out, name + ".bin"), "wb") as f:
        f.write(np.int32(len(tabs)).tobytes())
        for tb in tabs:
            f.write(tb.tobytes())
