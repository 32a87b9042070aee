#!/usr/bin/env python3
"""summarise a rocprofv3 kernel_trace.csv: per kernel name (escat kernels only) count, mean/min duration in us."""
import csv, sys, collections
d = collections.defaultdict(list)
for row in csv.DictReader(open(sys.argv[1])):
    n = row["Kernel_Name"]
    if not any(k in n for k in ("k_chain", "k_csr", "k_bsr", "k_scatter", "k_events", "k_build", "k_knn")):
        continue
    d[n.split("(")[0]].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
for n, v in sorted(d.items()):
    print(f"{n[:70]:70s} n={len(v):4d} mean={sum(v)/len(v):9.1f}us min={min(v):9.1f}us")
