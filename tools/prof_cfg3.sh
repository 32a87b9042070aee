#!/bin/bash
# usage: tools/prof_cfg3.sh <tag> [n]  -- HBM traffic (FETCH_SIZE / WRITE_SIZE, separate --pmc passes) of the multi-launch
# step kernels on one large cloud (cfg3 shape; default 1M nodes so that the single-process kNN stays under a minute)
tag=$1
n=${2:-1000000}
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/cfg3_$tag
mkdir -p $out/fetch $out/write
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -- python3 $root/tools/bench_cfg3.py --n $n --reps 1 --gen-workers 1 > $out/fetch/bench.json 2> $out/fetch/err.log || { tail -5 $out/fetch/err.log; exit 1; }
timeout -s KILL 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -- python3 $root/tools/bench_cfg3.py --n $n --reps 1 --gen-workers 1 > $out/write/bench.json 2> $out/write/err.log || { tail -5 $out/write/err.log; exit 1; }
python3 - <<PY
import csv, glob, json, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for ctr, sub in (("FETCH_SIZE", "fetch"), ("WRITE_SIZE", "write")):
    for f in glob.glob("$out/" + sub + "/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"]
            if "_step" in n and r["Counter_Name"] == ctr:
                acc[n.split("(")[0].replace("void ", "").strip()][ctr].append(float(r["Counter_Value"]))
b = json.loads(open("$out/fetch/bench.json").read().strip().splitlines()[-1])
res = {"bench": b, "kernels": {}}
for k, v in acc.items():
    f, w = v.get("FETCH_SIZE", []), v.get("WRITE_SIZE", [])
    if f and w:
        res["kernels"][k] = {"FETCH_SIZE_KB_mean": sum(f) / len(f), "WRITE_SIZE_KB_mean": sum(w) / len(w), "dispatches": len(f),
                             "hbm_bytes_per_launch_fetch_doubled": int((2 * sum(f) / len(f) + sum(w) / len(w)) * 1024),
                             "hbm_bytes_per_launch_raw": int((sum(f) / len(f) + sum(w) / len(w)) * 1024)}
json.dump(res, open("$out/summary.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
