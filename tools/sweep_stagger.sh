#!/bin/bash
# experiment: de-phase co-resident workgroups (ESCAT_DBG_STAGGER, 10-ns ticks) / one workgroup per CU (ESCAT_DBG_PER_CU)
out=gpurun_out/r3
mkdir -p $out
for s in 0 400 700 1000 -700; do
  ESCAT_DBG_STAGGER=$s timeout -k 10 120 python3 tools/microbench.py --Jmin 4 --Jmax 4 --gen-workers 16 > $out/stg_$s.json 2> $out/stg_$s.err || exit 1
  echo "stagger $s: $(cat $out/stg_$s.json)"
done
ESCAT_DBG_PER_CU=1 timeout -k 10 120 python3 tools/microbench.py --Jmin 3 --Jmax 5 --gen-workers 16 > $out/percu1.json 2> $out/percu1.err || exit 1
echo "per_cu 1: $(cat $out/percu1.json)"
timeout -k 10 120 python3 tools/microbench.py --Jmin 3 --Jmax 5 --gen-workers 16 > $out/percu2.json 2> $out/percu2.err || exit 1
echo "default: $(cat $out/percu2.json)"
