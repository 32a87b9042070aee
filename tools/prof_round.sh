#!/bin/bash
# usage: tools/prof_round.sh <tag>   (on the GPU box, from the repo root)
# Everything the round's numbers come from, in one call: kernel trace + stats of the default bench, FETCH_SIZE and
# WRITE_SIZE in separate --pmc passes (-> profiles/hbm_traffic.json via tools/pmc_to_traffic.py, stamped with the hash of
# libescat.so), the LDS and wait counters of the chain kernels, and the full default bench line.  Summaries are collected
# under gpurun_out/profiles_<tag>/ ; copy them to profiles/ .
tag=$1
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/profiles_$tag
mkdir -p $out
bash $root/tools/prof_bench.sh $tag > $out/${tag}_kernel_durations.txt 2>&1 || { tail -5 $out/${tag}_kernel_durations.txt; exit 1; }
cp $root/gpurun_out/prof_$tag/trace/bench.json $out/${tag}_bench_under_rocprof.json
cp $root/gpurun_out/prof_$tag/trace/*/*kernel_stats.csv $out/${tag}_kernel_stats.csv 2>/dev/null
python3 $root/tools/pmc_to_traffic.py $root/gpurun_out/prof_$tag > $out/${tag}_traffic_stdout.txt 2>&1 || { tail -5 $out/${tag}_traffic_stdout.txt; exit 1; }
cp $root/profiles/hbm_traffic.json $out/hbm_traffic.json
bash $root/tools/prof_lds.sh $tag > $out/${tag}_lds_counters.txt 2>&1 || { tail -5 $out/${tag}_lds_counters.txt; exit 1; }
cp $root/gpurun_out/lds_$tag/summary.json $out/${tag}_lds_counters.json
bash $root/tools/prof_wait.sh $tag > $out/${tag}_wait_counters.txt 2>&1 || { tail -5 $out/${tag}_wait_counters.txt; exit 1; }
cp $root/gpurun_out/wait_$tag/summary.json $out/${tag}_wait_counters.json
cd $root && timeout -k 10 600 python3 bench.py > $out/${tag}_bench.json 2> $out/${tag}_bench.err || { tail -5 $out/${tag}_bench.err; exit 1; }
tail -c 400 $out/${tag}_bench.json
