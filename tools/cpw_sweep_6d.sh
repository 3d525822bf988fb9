#!/bin/bash
# tools/cpw_sweep_6d.sh OUTFILE   (GPU box) — round 6: chunks per workgroup of the 25^6 sweeps under the strip schedule
# (pi_create's defaults, 256 threads x 4 chunks for evaluation and x 3 for improvement, were picked under the slab schedule:
# profiles/r03/block_cpw_sweep_6d.txt).  Product path on the bench state, class memory orders, one process per variant.
OUT=$1
: > "$OUT"
for cfg in "double_cartpole 25" "double_cartpole_swingup 25"; do
  set -- $cfg
  for v in "256 4 3" "256 2 3" "256 3 3" "256 6 3" "256 8 3" "256 4 2" "256 4 4" "256 4 6" "512 2 2" "512 4 3"; do
    set -- $cfg $v
    PI_MI355_EVAL_BLOCK=$3 PI_MI355_IMPROVE_BLOCK=$3 PI_MI355_EVAL_CPW=$4 PI_MI355_IMPROVE_CPW=$5 PI_MI355_KERNEL_CACHE=/tmp/kc_sweep \
      timeout -k 10 200 python3 tools/eval_states.py --env $1 --bins $2 --state bench --groups 3 --sweeps 10 --improve 2 2>/dev/null | \
      python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1@$2 block $3 eval cpw $4 improve cpw $5  eval %.4f ms  improve %.4f ms' % (min(d['eval_ms_per_sweep']), d['improve_ms_per_sweep']))" >> "$OUT"
  done
done
cat "$OUT"
