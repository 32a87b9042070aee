#!/bin/bash
# usage: tools/prof_both.sh <tag>  -- LDS and wait counters (two --pmc passes) of the both-tracks kernel and of the two
# separate chains over tools/bench_both.py, summarised per kernel into gpurun_out/both_<tag>/{lds,wait}.json
tag=$1
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/both_$tag
mkdir -p $out/lds $out/wait
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE \
    --output-format csv -d $out/lds -- python3 $root/tools/bench_both.py --reps 3 --gen-workers 1 > $out/lds/bench.json 2> $out/lds/err.log || { tail -5 $out/lds/err.log; exit 1; }
timeout -s KILL 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU \
    --output-format csv -d $out/wait -- python3 $root/tools/bench_both.py --reps 3 --gen-workers 1 > $out/wait/bench.json 2> $out/wait/err.log || { tail -5 $out/wait/err.log; exit 1; }
python3 - <<PY
import csv, glob, collections, json
for which in ("lds", "wait"):
    d = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob("$out/%s/*/*counter_collection.csv" % which):
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"]
            if "k_chain" not in n:
                continue
            d[n.split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
    res = {}
    for k, v in d.items():
        m = {c: sum(x) / len(x) for c, x in v.items()}
        m["launches"] = len(next(iter(v.values())))
        wc = m.get("SQ_WAVE_CYCLES", 0.0)
        for c in list(m):
            if c.startswith("SQ_") and c not in ("SQ_WAVE_CYCLES", "SQ_WAVES", "SQ_BUSY_CYCLES") and wc and "INSTS" not in c and "LDS_" not in c:
                m[c + "_frac_of_wave_cycles"] = m[c] / wc
        if m.get("SQ_LDS_IDX_ACTIVE"):
            m["lds_conflict_frac_of_active"] = m.get("SQ_LDS_BANK_CONFLICT", 0.0) / m["SQ_LDS_IDX_ACTIVE"]
        res[k] = m
    json.dump(res, open("$out/%s.json" % which, "w"), indent=1)
    for k, m in res.items():
        print(which, k)
        for c, x in sorted(m.items()):
            print("   %-44s %.5g" % (c, x))
PY
