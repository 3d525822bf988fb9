"""tools/stationary_census.py [--env E --bins B] — how much of a real run's evaluation sweeps recompute values that cannot
change?  (GPU box; experiment.)  A Jacobi sweep's output for a state is a function of the 2^D corner values of its successor
cell: where those did not change in the previous sweep — bit for bit — the state's new value is its old one.  This census
runs real policy-iteration rounds and, inside evaluations, reports per look the share of states whose value did not change
in the last sweep and the share of 1 024-state tiles (and of whole planes of memory dimension 0) in which NO state changed:
an upper bound on what a bit-exact "skip what cannot change" scheme could save (the tile's inputs lie in other tiles)."""
import argparse, json, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
import torch
from dynamicprogramming_amd import envs

ap = argparse.ArgumentParser()
ap.add_argument("--env", default="double_pendulum_swingup")
ap.add_argument("--bins", type=int, default=80)
ap.add_argument("--rounds", type=int, default=4)
ap.add_argument("--every", type=int, default=500)
args = ap.parse_args()
cls = envs.ENVS[args.env]
s = envs.make(args.env, args.bins, config=envs.CudaPIConfig(**cls.CONFIG), device="cuda:0")
n = s.n_states
gamma = float(np.float32(s.config.gamma))
plane = n // s._backend.engine._shape[0] if hasattr(s._backend.engine, "_shape") else n
for rnd in range(args.rounds):
    sweeps, out = 0, []
    while sweeps < s.config.max_eval_iter:
        k = 25
        s._evaluation_sweeps(k, gamma)
        sweeps += k
        delta = float(s._d_delta.item())
        if sweeps % args.every == 0 or delta < s.config.theta:
            a, b = s.d_value_function[:n], s.d_new_value_function[:n]        # the last two iterates
            same = a.view(torch.int32) == b.view(torch.int32)
            tiles = same[: n // 1024 * 1024].view(-1, 1024).all(dim=1)
            planes = same[: n // plane * plane].view(-1, plane).all(dim=1)
            out.append({"sweep": sweeps, "residual": delta, "states_unchanged": float(same.float().mean()),
                        "tiles_unchanged": float(tiles.float().mean()), "planes_unchanged": float(planes.float().mean())})
            print(json.dumps({"round": rnd, **out[-1]}), flush=True)
        if delta < s.config.theta:
            break
    s.stats["eval_sweeps"] += sweeps
    s.policy_improvement()
    print(json.dumps({"round": rnd, "sweeps": sweeps, "changed": s.stats.get("last_changed")}), flush=True)
