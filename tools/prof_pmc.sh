#!/bin/bash
# usage: tools/prof_pmc.sh <tag> "<counters>" [microbench args] -- one --pmc pass over the microbench at J=4
tag=$1; ctrs=$2; shift 2
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/pmc_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --pmc $ctrs --output-format csv -d $out -- python3 $root/tools/microbench.py --gen-workers 1 --reps 2 --Jmin 4 --Jmax 4 "$@" > $out/mb.json 2> $out/err.log || { tail -5 $out/err.log; exit 1; }
python3 - <<PY
import csv, glob, collections
d=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$out/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        n=r["Kernel_Name"]
        if "k_chain" in n: d[n.split("(")[0].replace("void ","")][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in d.items():
    print(k)
    for c,vals in sorted(v.items()): print("   %-28s %.4g" % (c, sum(vals)/len(vals)))
PY
