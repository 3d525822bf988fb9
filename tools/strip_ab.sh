#!/bin/bash
# tools/strip_ab.sh OUT ENV BINS SWEEPS IMPROVE PERIOD...   (GPU box; run from the repository root)
# Sweep time of one config under the slab schedule (PERIOD 0) and the strip schedule with the given periods in
# states ("auto" = the library's choice) — tools/eval_states.py on the bench state, one process per variant, the
# variants interleaved twice so that box drift shows.  One line per run into OUT (jsonl) and a table on stdout.
O=$1; E=$2; B=$3; S=$4; I=$5; shift 5
mkdir -p "$(dirname "$O")"
for rep in 1 2; do
  for P in "$@"; do
    PI_MI355_STRIP=$P timeout -k 10 280 python3 tools/eval_states.py --env $E --bins $B --state bench --sweeps $S --groups 3 --improve $I 2>/dev/null \
      | python3 -c "import sys, json; d = json.loads(sys.stdin.readline()); d['strip'] = '$P'; d['rep'] = $rep; print(json.dumps(d))" >> "$O" || echo "variant $P failed"
  done
done
python3 - "$O" <<'PY'
import json, sys
rows = [json.loads(l) for l in open(sys.argv[1])]
print("%-28s %5s %-10s %4s  %-30s %s" % ("env", "bins", "strip", "rep", "eval ms per sweep (groups)", "improve ms"))
for d in rows:
    print("%-28s %5d %-10s %4d  %-30s %s" % (d["env"], d["bins"], d["strip"], d["rep"],
          " ".join("%.4f" % m for m in d["eval_ms_per_sweep"]), "%.4f" % d.get("improve_ms_per_sweep", float("nan"))))
PY
