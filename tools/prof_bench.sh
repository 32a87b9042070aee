#!/bin/bash
# usage: tools/prof_bench.sh <tag>  -- kernel trace + stats of the default bench, then FETCH_SIZE / WRITE_SIZE in separate --pmc passes
tag=$1
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/prof_$tag
mkdir -p $out/trace $out/fetch $out/write
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $root/bench.py --steps 10 --warmup 3 --gen-workers 1 --no-cpu-baseline --no-cfg3 --no-steady-state > $out/trace/bench.json 2> $out/trace/err.log || exit 1
python3 $root/tools/kstats.py $out/trace/*/*kernel_trace.csv
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -- python3 $root/bench.py --steps 3 --warmup 1 --gen-workers 1 --no-cpu-baseline --no-cfg3 --no-steady-state > $out/fetch/bench.json 2> $out/fetch/err.log || exit 1
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -- python3 $root/bench.py --steps 3 --warmup 1 --gen-workers 1 --no-cpu-baseline --no-cfg3 --no-steady-state > $out/write/bench.json 2> $out/write/err.log || exit 1
ls $out/fetch/* $out/write/*
