#!/bin/bash
# usage: tools/prof_lds.sh <tag>  -- LDS / occupancy counters of the chain kernels over the default bench (cfg2), one --pmc
# pass (8 SQ + 1 GRBM counter: the per-pass limits), summarised per kernel into gpurun_out/lds_<tag>/summary.json
tag=$1
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/lds_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE \
    --output-format csv -d $out -- python3 $root/bench.py --steps 3 --warmup 1 --gen-workers 1 --no-cpu-baseline --no-cfg3 --no-steady-state > $out/bench.json 2> $out/err.log || { tail -5 $out/err.log; exit 1; }
python3 - <<PY
import csv, glob, collections, json
d = collections.defaultdict(lambda: collections.defaultdict(list))
meta = {}
for f in glob.glob("$out/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "k_chain" not in n and "k_bsr3_step" not in n and "k_csr_step" not in n:
            continue
        k = n.split("(")[0].replace("void ", "")
        d[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        meta[k] = {"vgpr": r.get("VGPR_Count"), "accum_vgpr": r.get("Accum_VGPR_Count"), "sgpr": r.get("SGPR_Count"),
                   "lds_bytes": r.get("LDS_Block_Size"), "workgroup": r.get("Workgroup_Size"), "grid": r.get("Grid_Size")}
res = {}
for k, v in d.items():
    m = {c: sum(x) / len(x) for c, x in v.items()}
    m["launches"] = len(next(iter(v.values())))
    if m.get("SQ_LDS_IDX_ACTIVE"):
        m["lds_conflict_frac_of_active"] = m.get("SQ_LDS_BANK_CONFLICT", 0.0) / m["SQ_LDS_IDX_ACTIVE"]
    if m.get("SQ_BUSY_CYCLES"):
        m["lds_active_per_busy_cycle"] = m.get("SQ_LDS_IDX_ACTIVE", 0.0) / m["SQ_BUSY_CYCLES"]
    if m.get("SQ_WAVE_CYCLES"):
        m["wait_inst_lds_frac_of_wave_cycles"] = m.get("SQ_WAIT_INST_LDS", 0.0) / m["SQ_WAVE_CYCLES"]
    m.update(meta[k])
    res[k] = m
json.dump(res, open("$out/summary.json", "w"), indent=1)
for k, m in res.items():
    print(k)
    for c, x in sorted(m.items()):
        print("   %-36s %s" % (c, ("%.5g" % x) if isinstance(x, float) else x))
PY
