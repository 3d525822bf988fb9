#!/bin/bash
# tools/sweep_knobs.sh — on the GPU box: time the bench under the library's tuning knobs and
# collect L2 hit/miss counters per variant.  Usage: bash tools/sweep_knobs.sh OUTDIR "VAR=val ..." ...
# Each remaining argument is one variant (a space-separated list of env assignments).
OUT=$1; shift
mkdir -p "$OUT"
R=$PWD
i=0
for V in "$@"; do
  i=$((i+1))
  tag=$(echo "$V" | tr ' =' '__' | tr -cd 'A-Za-z0-9_')
  [ -z "$tag" ] && tag=default
  ( export $V; timeout -k 10 150 python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline $BENCH_ARGS > "$OUT/bench_$tag.json" 2>"$OUT/bench_$tag.err" )
  ( cd /tmp && export TMPDIR=/tmp $V && timeout -k 10 150 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum --kernel-trace --output-format csv -d "$R/$OUT/pmc_$tag" -- python3 "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline $BENCH_ARGS > "$R/$OUT/pmc_$tag.log" 2>&1 )
  echo "variant $i [$V] done"
done
