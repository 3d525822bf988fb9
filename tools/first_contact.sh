#!/bin/bash
# tools/first_contact.sh   (a node with >= 2 MI355X; run from the repository root)
# RCCL with N > 1 ranks has never run on this code: every development box had ONE GPU.  Whoever gets
# a node should learn in two minutes whether the multi-GPU path works, not from inside the timed scaling run:
#   1. tests/test_gpu_rccl.py — fresh child ranks through torch.distributed.run, halo exchange and all-gather, a C4-shaped
#      and an odd 6-D grid, bit-identity with the single-rank run, and the shortened bench line; bounded (kills its group).
#   2. bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline — the scaling run's own command, shortened; its
#      check.exchange is printed (transport, plan, bytes per sweep per rank, sharded == unsharded before and after).
# Nothing here retries a GPU step; a failed step ends the script.  No transport work is expected from this script's user:
# if step 1 fails, the message names the rank, the case and the first differing quantity.
set -e -o pipefail
export HSA_ENABLE_IPC_MODE_LEGACY=0
N=$(python3 -c 'import torch; print(torch.cuda.device_count())')
echo "[first contact] $N GPU(s) visible"
if [ "$N" -lt 2 ]; then
  echo "[first contact] RCCL with N > 1 ranks needs at least 2 GPUs: nothing to do here (the gloo / in-process / peer-to-peer"
  echo "                rehearsals of the same code run in tests/test_distributed_gloo.py and tests/test_gpu_p2p.py)"
  exit 0
fi
echo "[first contact] 1/2  pytest tests/test_gpu_rccl.py"
timeout -k 10 900 python3 -m pytest tests/test_gpu_rccl.py -x -q
echo "[first contact] 2/2  bench.py --gpus 2 (3 steps)"
timeout -k 10 600 python3 bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline --attempt-timeout 200 > /tmp/first_contact_bench.json
python3 - <<'PY'
import json
line = [json.loads(l) for l in open("/tmp/first_contact_bench.json") if l.startswith("{") and '"metric"' in l][-1]
x = line["check"]["exchange"]
print(json.dumps({"n_gpus": line["n_gpus"], "value": line["value"], "ms_per_step": line["ms_per_step"],
                  "transport": x.get("transport"), "mode": x.get("mode"), "ladder_mode": x.get("ladder_mode"),
                  "attempts": x.get("attempts"), "bit_identical": x.get("bit_identical"),
                  "per_rank": x.get("per_rank")}, indent=1))
assert x.get("transport") == "rccl" and x["bit_identical"]["ok"], "the first rung over RCCL did not go through: see attempts"
print("[first contact] ok: 2 ranks over RCCL, sharded == unsharded before and after the timed steps")
PY
