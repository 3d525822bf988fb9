"""tools/full_run.py env bins — one full run() to convergence with progress on stderr; prints sweeps, seconds and digests of V / policy."""
import logging, sys, time, json
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
logging.basicConfig(level=logging.INFO, stream=sys.stderr, format="%(asctime)s %(message)s")
import torch
from dynamicprogramming_amd import envs
name, bins = sys.argv[1], int(sys.argv[2])
s = envs.make(name, bins)
torch.cuda.synchronize()
t0 = time.perf_counter()
s.run()
dt = time.perf_counter() - t0
st = s.stats
bk = s.n_states * (st["eval_sweeps"] + st["improve_sweeps"] * s.n_actions)
import hashlib
print(json.dumps({"env": name, "bins": bins, "states": s.n_states, "pi_iterations": st["pi_iterations"], "eval_sweeps": st["eval_sweeps"],
                  "improve_sweeps": st["improve_sweeps"], "stable": st.get("stable"), "seconds": round(dt, 2),
                  "backups_per_s": bk / dt, "us_per_eval_sweep": dt / max(st["eval_sweeps"], 1) * 1e6,
                  "sha256_V": hashlib.sha256(s.value_function.tobytes()).hexdigest()[:16],
                  "sha256_policy": hashlib.sha256(s.policy.tobytes()).hexdigest()[:16]}), flush=True)
