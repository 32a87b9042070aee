#!/usr/bin/env python3
"""print the kernels of the last fresh-COO step of a rocprofv3 kernel trace, in time order (start offset us, duration us)"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/*/*kernel_trace.csv')[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'k_coo_sorted_to_csr' in r['Kernel_Name']]
i0 = idx[-1]
t0 = int(rows[i0]['Start_Timestamp'])
busy = 0
for r in rows[i0:]:
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    busy += d
    if d >= float(sys.argv[2]) if len(sys.argv) > 2 else True:
        print(f"{(int(r['Start_Timestamp']) - t0) / 1e3:9.1f} {d:8.1f}  {r['Kernel_Name'][:100]}")
print("span us", (int(rows[-1]['End_Timestamp']) - t0) / 1e3, "kernel-busy us", busy, "launches", len(rows) - i0)
