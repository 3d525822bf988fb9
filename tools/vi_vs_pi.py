"""tools/vi_vs_pi.py — value iteration (the fused sweep of SURVEY section 8f.4) against policy iteration, to
convergence, on one MI355X: sweeps, seconds, and how far apart the two fixed points are.

Policy iteration = the reference's run() (evaluation sweeps to theta, then one improvement sweep, until the policy
is stable).  Value iteration = solver.value_iteration(): V' = max_a Q and the argmax in ONE kernel per sweep until
the residual (looked at every 25 sweeps) is below the same theta.  A value-iteration sweep costs what an improvement
sweep costs (n_actions backups per state); an evaluation sweep costs one backup per state.
usage: python tools/vi_vs_pi.py [--vi-only] env@bins [env@bins ...]   (default: pendulum@200 cartpole_swingup@50)
--vi-only: value iteration alone (grids whose policy iteration does not fit one GPU call), progress on stderr.
"""
import json
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
import torch

from dynamicprogramming_amd import envs

cases = [(a, int(b)) for a, b in (x.split("@") for x in sys.argv[1:] if "@" in x)] or \
    [("pendulum", 200), ("cartpole_swingup", 50)]
VI_ONLY = "--vi-only" in sys.argv
for name, bins in cases:
    if VI_ONLY:
        vi = envs.make(name, bins)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        total, delta = 0, float("inf")
        while total < 400_000 and not delta < vi.config.theta:
            delta = vi.value_iteration(max_iter=500)         # 500-sweep slices: a progress line each
            total = vi.stats["value_sweeps"]
            print(f"{name}@{bins}: {total} sweeps, residual {delta:.3e}, {time.perf_counter() - t0:.1f} s", file=sys.stderr, flush=True)
        torch.cuda.synchronize()
        t_vi = time.perf_counter() - t0
        print(json.dumps({"grid": f"{name} {bins}^{vi._D}", "states": vi.n_states, "actions": vi.n_actions,
                          "theta": vi.config.theta, "gamma": vi.config.gamma,
                          "value_iteration": {"sweeps": total, "last_residual": delta, "converged": bool(delta < vi.config.theta),
                                              "seconds": round(t_vi, 3), "backups": vi.n_states * vi.n_actions * total,
                                              "backups_per_s": vi.n_states * vi.n_actions * total / t_vi}}), flush=True)
        continue
    pi = envs.make(name, bins)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pi.run()
    t_pi = time.perf_counter() - t0
    vi = envs.make(name, bins)
    limit = 400_000
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    delta = vi.value_iteration(max_iter=limit)
    torch.cuda.synchronize()
    t_vi = time.perf_counter() - t0
    sweeps_vi = vi.stats["value_sweeps"]
    # device tensors are in the solver's memory order; host-side results (pi.value_function) in the env's
    V_vi = np.ascontiguousarray(vi._to_user(vi.d_value_function[:vi.n_states].cpu().numpy()))
    P_vi = np.ascontiguousarray(vi._to_user(vi.d_policy[:vi.n_states].cpu().numpy()))
    st = pi.stats
    n, na = pi.n_states, pi.n_actions
    scale = max(1.0, float(np.abs(pi.value_function).max()))
    print(json.dumps({
        "grid": f"{name} {bins}^{pi._D}", "states": n, "actions": na, "theta": pi.config.theta, "gamma": pi.config.gamma,
        "policy_iteration": {"pi_iterations": st["pi_iterations"], "eval_sweeps": st["eval_sweeps"],
                             "improve_sweeps": st["improve_sweeps"], "stable": st.get("stable"),
                             "seconds": round(t_pi, 3),
                             "backups": n * (st["eval_sweeps"] + na * st["improve_sweeps"])},
        "value_iteration": {"sweeps": sweeps_vi, "last_residual": delta, "converged": bool(delta < vi.config.theta),
                            "seconds": round(t_vi, 3), "backups": n * na * sweeps_vi},
        "fixed_points": {"max_abs_dV": float(np.abs(V_vi - pi.value_function).max()),
                         "max_abs_dV_over_scale": float(np.abs(V_vi - pi.value_function).max() / scale),
                         "policy_agreement": float(np.mean(P_vi == pi.policy))}}), flush=True)
