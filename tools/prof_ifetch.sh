#!/bin/bash
# usage: tools/prof_ifetch.sh <tag>  -- instruction-fetch counters of the chain kernels over tools/bench_both.py
tag=$1
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/ifetch_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --pmc SQ_IFETCH SQ_IFETCH_LEVEL SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES \
    --output-format csv -d $out -- python3 $root/tools/bench_both.py --reps 3 --gen-workers 1 > $out/bench.json 2> $out/err.log || { tail -5 $out/err.log; exit 1; }
python3 - <<PY
import csv, glob, collections, json
d = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$out/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "k_chain" not in n:
            continue
        d[n.split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {k: {c: sum(x) / len(x) for c, x in v.items()} for k, v in d.items()}
json.dump(res, open("$out/summary.json", "w"), indent=1)
for k, m in res.items():
    print(k)
    for c, x in sorted(m.items()):
        print("   %-44s %.5g" % (c, x))
PY
