"""tools/eval_states.py — time (and let rocprofv3 profile) evaluation sweeps of one config on two
V/policy states (GPU box):

  bench : what bench.py times — V ~ N(0,1), policy ~ U{0..nA-1}, after two bench steps
          (10 evaluation sweeps + 1 improvement each), i.e. a greedy policy on a noise V.
  real  : a policy-iteration state — from V = 0, policy = 0 run `--pi-iters` (default 3) outer
          iterations with at most `--max-eval` (default 2000) sweeps each, i.e. what the sweeps of a
          real run() see (VERDICT r01 weak point 4: 742 us/sweep in the real C4 run vs 502 in bench).

--save PATH / --load PATH keep the prepared state in a torch file so the PMC passes need not
re-run the preparation under the profiler.  Prints one JSON line with the per-sweep time of
`--groups` groups of `--sweeps` sweeps (HIP events on the launch stream).
"""
import argparse, json, sys, time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
import torch

from dynamicprogramming_amd import envs

ap = argparse.ArgumentParser()
ap.add_argument("--env", default="double_pendulum_swingup")
ap.add_argument("--bins", type=int, default=80)
ap.add_argument("--state", choices=["bench", "real"], default="bench")
ap.add_argument("--sweeps", type=int, default=50)
ap.add_argument("--groups", type=int, default=5)
ap.add_argument("--pi-iters", type=int, default=3)
ap.add_argument("--max-eval", type=int, default=2000)
ap.add_argument("--save")
ap.add_argument("--load")
ap.add_argument("--improve", type=int, default=0, help="also time this many improvement sweeps")
args = ap.parse_args()

cls = envs.ENVS[args.env]
cfg = envs.CudaPIConfig(**{**cls.CONFIG, "max_eval_iter": args.max_eval})
solver = envs.make(args.env, args.bins, config=cfg, device="cuda:0")
n, nA = solver.n_states, solver.n_actions
gamma = float(np.float32(cfg.gamma))
t_prep = time.perf_counter()
if args.load:
    st = torch.load(args.load)
    solver.d_value_function[:n].copy_(st["V"])
    solver.d_policy[:n].copy_(st["P"])
elif args.state == "bench":
    gen = torch.Generator(device="cpu").manual_seed(0)
    solver.d_value_function[:n].copy_(torch.randn(n, generator=gen, dtype=torch.float32))
    solver.d_policy[:n].copy_(torch.randint(0, nA, (n,), generator=gen, dtype=torch.int32))
    solver.d_new_value_function.copy_(solver.d_value_function)
    for _ in range(2):
        solver._evaluation_sweeps(10, gamma)
        solver._improvement_sweep(gamma)
else:
    per_iter = []
    for _ in range(args.pi_iters):
        torch.cuda.synchronize()
        t_it, s_it = time.perf_counter(), solver.stats["eval_sweeps"]
        solver.policy_evaluation()
        torch.cuda.synchronize()
        dt_it, n_it = time.perf_counter() - t_it, solver.stats["eval_sweeps"] - s_it
        solver.policy_improvement()
        per_iter.append({"sweeps": n_it, "ms_per_sweep": dt_it / max(n_it, 1) * 1e3,
                         "changed": solver.stats.get("last_changed")})
        print(json.dumps(per_iter[-1]), file=sys.stderr, flush=True)
solver.d_new_value_function.copy_(solver.d_value_function)
torch.cuda.synchronize()
t_prep = time.perf_counter() - t_prep
if args.save:
    torch.save({"V": solver.d_value_function[:n].cpu(), "P": solver.d_policy[:n].cpu()}, args.save)

pol = solver.d_policy[:n]
hist = torch.bincount(pol.to(torch.int64), minlength=nA).cpu().tolist()
out = {"per_pi_iteration": per_iter if args.state == "real" and not args.load else None, "env": args.env, "bins": args.bins, "state": args.state, "states": n, "prep_seconds": t_prep,
       "policy_histogram": hist, "eval_sweeps_so_far": solver.stats["eval_sweeps"]}
ms = []
for g in range(args.groups):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    solver._evaluation_sweeps(args.sweeps, gamma)
    e1.record()
    e1.synchronize()
    ms.append(e0.elapsed_time(e1) / args.sweeps)
out["eval_ms_per_sweep"] = ms
out["residual"] = float(solver._d_delta.item())
if args.improve:
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.improve):
        solver._improvement_sweep(gamma)
    e1.record()
    e1.synchronize()
    out["improve_ms_per_sweep"] = e0.elapsed_time(e1) / args.improve
print(json.dumps(out), flush=True)
