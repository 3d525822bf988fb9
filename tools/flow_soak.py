import hashlib, sys, time, json
sys.path.insert(0, '.')
import numpy as np, torch
from dynamicprogramming_amd import envs
out = {}   # tools/flow_soak.py — repeated full runs through the dataflow kernel: digests must agree, nothing may give up
for name, bins, reps in (("pendulum", 200, 40), ("mountain_car", 200, 30), ("double_pendulum_swingup", 15, 6), ("cartpole", 15, 40)):
    digests, secs, evals = set(), [], 0
    for r in range(reps):
        s = envs.make(name, bins, device="cuda:0")
        assert s._backend.engine.info(19) > 0
        t0 = time.perf_counter(); s.run(); secs.append(time.perf_counter() - t0)
        evals += s.stats["pi_iterations"]
        digests.add(hashlib.sha256(s.value_function.tobytes() + s.policy.tobytes()).hexdigest()[:16])
    out[f"{name}@{bins}"] = {"runs": reps, "evaluations": evals, "distinct_digests": len(digests), "seconds_min": min(secs), "seconds_max": max(secs)}
    print(json.dumps({f"{name}@{bins}": out[f"{name}@{bins}"]}), flush=True)
