#!/bin/bash
# tools/bench_matrix.sh OUTDIR "ENV_ASSIGNMENTS|bench args" ... — run bench.py variants on the GPU box.
OUT=$1; shift
mkdir -p "$OUT"
i=0
for V in "$@"; do
  i=$((i+1))
  envs="${V%%|*}"; args="${V#*|}"
  tag=$(echo "$V" | tr ' =|,' '____' | tr -cd 'A-Za-z0-9_-')
  ( [ -n "$envs" ] && export $envs; timeout -k 10 200 python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline $args > "$OUT/$tag.json" 2> "$OUT/$tag.err" )
  echo "[$i] $V rc=$?"
done
