#!/bin/bash
# usage: tools/prof_mb.sh <tag> [microbench args]  -- true kernel durations (rocprofv3 kernel trace) at T=1 and T=16
tag=$1; shift
root=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for J in 0 4; do
  out=$root/gpurun_out/prof_$tag/J$J
  mkdir -p $out
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $out -- python3 $root/tools/microbench.py --gen-workers 1 --reps 5 --Jmin $J --Jmax $J "$@" > $out/mb.json 2> $out/err.log || exit 1
  echo "== J=$J"; python3 $root/tools/kstats.py $out/*/*kernel_trace.csv
done
