"""tools/flow_bench.py — the one-launch dataflow evaluation (pi_eval_flow_kernel) against the sweep-by-sweep path on
launch-bound grids (GPU box): full run() of BASELINE config C2 (pendulum 200 x 200 x 21) and of the 4-D default grids,
wall time, sweeps, microseconds per sweep, digests of V / policy (identical by construction; checked).
"""
import hashlib, json, os, sys, time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
import torch

from dynamicprogramming_amd import envs

cases = [("pendulum", 200), ("mountain_car", 200), ("double_pendulum_swingup", 15), ("cartpole", 15)]
if len(sys.argv) > 1:
    cases = [(a.split("@")[0], int(a.split("@")[1])) for a in sys.argv[1:]]
out = []
for name, bins in cases:
    row = {"env": name, "bins": bins}
    for label, flag, xcd, whole in (("whole_run", "1", "1", "1"), ("xcd_local", "1", "1", "0"), ("one_launch", "1", "0", "0"),
                                    ("sweep_by_sweep", "0", "0", "0")):
        os.environ["PI_MI355_RESIDENT"] = flag
        os.environ["PI_MI355_XCD"] = xcd
        os.environ["PI_MI355_WHOLE_RUN"] = whole
        best = None
        for rep in range(3):
            s = envs.make(name, bins, device="cuda:0")
            kind = {True: "flow" if s._backend.engine.info(19) else "lds", False: "graphs"}[bool(s._backend.resident)]
            if s._backend.engine.info(30) > 0:
                kind = "xcd"
            elif xcd == "1":
                continue                                   # this grid has no XCD-local kernel: nothing to time
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            s.run()
            dt = time.perf_counter() - t0
            if best is None or dt < best[0]:
                best = (dt, s)
        if best is None:
            continue
        dt, s = best
        row[label] = {"path": kind, "seconds": dt, "eval_sweeps": s.stats["eval_sweeps"], "pi_iterations": s.stats["pi_iterations"],
                      "us_per_sweep": dt / s.stats["eval_sweeps"] * 1e6, "eval_seconds": s.stats["eval_seconds"],
                      "V_sha": hashlib.sha256(s.value_function.tobytes()).hexdigest()[:16],
                      "policy_sha": hashlib.sha256(s.policy.tobytes()).hexdigest()[:16],
                      "xcd_evaluations_fallbacks_runs": [s._backend.xcd_evaluations, s._backend.xcd_fallbacks, s._backend.whole_runs]}
    row["identical"] = (row["one_launch"]["V_sha"] == row["sweep_by_sweep"]["V_sha"]
                        and row["one_launch"]["policy_sha"] == row["sweep_by_sweep"]["policy_sha"]
                        and row["one_launch"]["eval_sweeps"] == row["sweep_by_sweep"]["eval_sweeps"])
    row["speedup"] = row["sweep_by_sweep"]["seconds"] / row["one_launch"]["seconds"]
    for label in ("xcd_local", "whole_run"):
        if label in row:
            row["identical"] = row["identical"] and all(row[label][k] == row["sweep_by_sweep"][k]
                                                        for k in ("V_sha", "policy_sha", "eval_sweeps", "pi_iterations"))
            row["speedup_" + label] = row["sweep_by_sweep"]["seconds"] / row[label]["seconds"]
    print(json.dumps(row), flush=True)
    out.append(row)
