#!/bin/bash
# tools/block_cpw_sweep.sh OUTFILE — evaluation / improvement sweep time of the BASELINE configs for
# every (threads per workgroup, chunks per workgroup) pair (GPU box); picks pi_create's defaults.
OUT=$1
: > "$OUT"
for cfg in "cartpole_swingup 50 100" "double_pendulum_swingup 80 100" "double_cartpole 25 20" "double_cartpole_swingup 25 10"; do
  set -- $cfg
  for b in 256 512 1024; do for c in 1 2 3; do
    PI_MI355_EVAL_BLOCK=$b PI_MI355_IMPROVE_BLOCK=$b PI_MI355_EVAL_CPW=$c PI_MI355_IMPROVE_CPW=$c PI_MI355_KERNEL_CACHE=/tmp/kc_sweep \
      python3 tools/eval_states.py --env $1 --bins $2 --state bench --groups 2 --sweeps $3 --improve 2 2>/dev/null | \
      python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1@$2 block $b cpw $c  eval %.4f ms  improve %.4f ms' % (min(d['eval_ms_per_sweep']), d['improve_ms_per_sweep']))" >> "$OUT"
  done; done
done
