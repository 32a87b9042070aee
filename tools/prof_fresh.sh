#!/bin/bash
# kernel trace of the fresh-COO drop-in loop (tools/prof_fresh_coo.py); usage: tools/prof_fresh.sh <tag>  (ESCAT_INGEST selects the policy)
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/r3/prof_fresh_$1
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 $root/tools/prof_fresh_coo.py > $out/out.json 2> $out/err.log || { tail -5 $out/err.log; exit 1; }
cat $out/out.json
