#!/bin/bash
# usage: tools/prof_wait.sh <tag>  -- where the waves' cycles go in the chain kernels (one --pmc pass over the default bench)
tag=$1
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/wait_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS \
    --output-format csv -d $out -- python3 $root/bench.py --steps 3 --warmup 1 --gen-workers 1 --no-cpu-baseline --no-cfg3 --no-steady-state > $out/bench.json 2> $out/err.log || { tail -5 $out/err.log; exit 1; }
python3 - <<PY
import csv, glob, collections, json
d = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$out/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "k_chain" not in n:
            continue
        d[n.split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {}
for k, v in d.items():
    m = {c: sum(x) / len(x) for c, x in v.items()}
    wc = m.get("SQ_WAVE_CYCLES", 0.0)
    if wc:
        for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS"):
            m[c + "_frac_of_wave_cycles"] = m.get(c, 0.0) / wc
    res[k] = m
json.dump(res, open("$out/summary.json", "w"), indent=1)
for k, m in res.items():
    print(k)
    for c, x in sorted(m.items()):
        print("   %-44s %.5g" % (c, x))
PY
