#!/bin/bash
# usage: tools/sweep_env.sh VAR "v1 v2 ..." out_prefix command...   (one run of the command per value of VAR)
var=$1; vals=$2; out=$3; shift 3
for v in $vals; do
  env "$var=$v" timeout -k 10 300 "$@" > "${out}_${v}.json" 2> "${out}_${v}.err" || { echo "value $v failed"; exit 1; }
  echo "$var=$v: $(cat ${out}_${v}.json)"
done
