"""tools/xcd_soak.py — the XCD-local kernel (pi_xcd_kernel) under repetition and under UNEVEN load (GPU box).

Repeated full run()s of launch-bound 2-D grids through the one-launch path (pi_policy_iteration) and round by round
(one pi_policy_evaluation launch per evaluation): every run of a config must end with the same sha256 of (V, policy) as
the sweep-by-sweep reference run, no launch may fall back.  Second half: the same while another stream keeps the chip
busy with a memory-streaming kernel of changing size (torch copies on a side stream), i.e. with the other XCDs' fabric
links, the HBM channels and part of XCD 0's CUs occupied at uneven times — the condition MI355X_MICROARCH.md asks every
hand-off to be tested under.  A launch that cannot be placed under that load falls back (counted, allowed); results
must not change.
"""
import hashlib, json, os, sys, time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch

from dynamicprogramming_amd import envs

CASES = (("pendulum", 200, 40), ("mountain_car", 200, 40), ("continuous_mountain_car", 200, 20), ("mountain_car", 113, 40))


def digest(s):
    return hashlib.sha256(s.value_function.tobytes() + s.policy.tobytes()).hexdigest()[:16]


def reference(name, bins):
    os.environ["PI_MI355_RESIDENT"] = "0"
    try:
        s = envs.make(name, bins, device="cuda:0")
        s.run()
        return digest(s), s.stats["eval_sweeps"], s.stats["pi_iterations"]
    finally:
        del os.environ["PI_MI355_RESIDENT"]


def soak(name, bins, reps, whole, noise):
    os.environ["PI_MI355_WHOLE_RUN"] = "1" if whole else "0"
    digests, secs, launches, fallbacks, evals = set(), [], 0, 0, 0
    side = torch.cuda.Stream()
    a = torch.empty(64 << 20, dtype=torch.float32, device="cuda:0")
    b = torch.empty_like(a)
    for r in range(reps):
        s = envs.make(name, bins, device="cuda:0")
        assert s._backend.whole_run == whole or not whole
        if noise:
            with torch.cuda.stream(side):
                for k in range(120):                       # ~20 ms of streaming copies of changing size (4 - 256 MB each)
                    m = (1 << 20) << ((r + k) % 7)
                    b[:m].copy_(a[:m])
        t0 = time.perf_counter()
        s.run()
        secs.append(time.perf_counter() - t0)
        digests.add(digest(s))
        launches += s._backend.whole_runs + s._backend.xcd_evaluations
        fallbacks += s._backend.xcd_fallbacks + (1 if whole and s._backend.xcd_evaluations > 0 else 0)
        evals += s.stats["pi_iterations"]
        torch.cuda.synchronize()
    return {"runs": reps, "policy_evaluations": evals, "launches": launches, "fallbacks": fallbacks, "digests": sorted(digests),
            "seconds_min": round(min(secs), 5), "seconds_median": round(sorted(secs)[len(secs) // 2], 5),
            "seconds_max": round(max(secs), 5)}


for name, bins, reps in CASES:
    ref = reference(name, bins)
    row = {"env": name, "bins": bins, "reference_digest": ref[0], "eval_sweeps": ref[1], "pi_iterations": ref[2]}
    for label, whole, noise in (("one_launch", True, False), ("round_by_round", False, False),
                                ("one_launch_under_load", True, True), ("round_by_round_under_load", False, True)):
        row[label] = soak(name, bins, reps, whole, noise)
        assert row[label]["digests"] == [ref[0]], (name, bins, label, row[label])
    print(json.dumps(row), flush=True)
