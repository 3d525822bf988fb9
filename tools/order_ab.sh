#!/bin/bash
# tools/order_ab.sh OUT ENV BINS SWEEPS IMPROVE ORDER...   (GPU box; run from the repository root)
# Sweep time of one config with its device arrays in the given memory orders (PI_MI355_ORDER: comma-separated, memory
# dimension k = the env's dimension ORDER[k], slowest first; "class" = the env class's own MEMORY_ORDER) — the product
# path (envs.make -> solver -> C ABI) on the bench state through tools/eval_states.py, with the live-state lists, the
# per-evaluation lists and the strip schedule as a real run has them.  One process per order, the list run twice so
# that box drift shows.  One line per run into OUT (jsonl), a table on stdout.  How MEMORY_ORDER of an env class is chosen
# since round 6 (tools/dim_order_sweep.py permutes the ENV instead and predates pi_set_option 4).
O=$1; E=$2; B=$3; S=$4; I=$5; shift 5
mkdir -p "$(dirname "$O")"
for rep in 1 2; do
  for P in "$@"; do
    if [ "$P" = "class" ]; then unset PI_MI355_ORDER; else export PI_MI355_ORDER=$P; fi
    timeout -k 10 280 python3 tools/eval_states.py --env $E --bins $B --state bench --sweeps $S --groups 3 --improve $I 2>/dev/null \
      | python3 -c "import sys, json; d = json.loads(sys.stdin.readline()); d['order'] = '$P'; d['rep'] = $rep; print(json.dumps(d))" >> "$O" || echo "order $P failed"
  done
done
unset PI_MI355_ORDER
python3 - "$O" <<'PY'
import json, sys
rows = [json.loads(l) for l in open(sys.argv[1])]
print("%-28s %5s %-14s %4s  %-30s %s" % ("env", "bins", "order", "rep", "eval ms per sweep (groups)", "improve ms"))
for d in rows:
    print("%-28s %5d %-14s %4d  %-30s %s" % (d["env"], d["bins"], d["order"], d["rep"],
          " ".join("%.4f" % m for m in d["eval_ms_per_sweep"]), "%.4f" % d.get("improve_ms_per_sweep", float("nan"))))
PY
