"""
Shared command line and training entry point of the runner scripts.

The reference's runners share ONE argparse surface (README.md:323-347; e.g.
runners/pendulum_cuda.py:268-287) plus two crane-only flags (overhead_crane_cuda.py:588-591).
Every flag is accepted here with the reference's name, type and default, so a command line
written for the reference works unchanged:

  solver-side, honoured      --bins N, --retrain, --save-path PATH, crane: --target-x X
  rollout-side, accepted     --render, --record PATH, --random [N], --episodes N, --steps N,
  and ignored (with a note)  --seed N, --no-plot, crane: --start-x X

One flag is an extension (the reference has no counterpart; off by default): --value-iteration trains with
the fused max-backup sweep of SURVEY.md section 8f.4 instead of policy iteration — the same fixed point
(measured: identical V and policy on the 80^4 double pendulum) in fewer sweeps.

The ignored group drives the reference's gymnasium / pygame / matplotlib evaluation harness, which is
outside the scope of this package: training, saving and loading behave as in
the reference (train when the archive is missing or --retrain is given, otherwise load it), the
rollouts are simply not run.  `--random` skips training like the reference does (it only ever ran
random rollouts).  The hybrid runner of the reference accepts-and-ignores flags the same way
(README.md:346-347).
"""
from __future__ import annotations

import argparse
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]  # repository root: makes `dynamicprogramming_amd` importable
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def bins_space_for(env_name: str, bins: int) -> dict:
    """The env's reference grid ranges with `bins` points per dimension (the reference rebuilds
    BINS_SPACE the same way when --bins is given: pendulum_cuda.py:293-296)."""
    from dynamicprogramming_amd import envs
    return envs.ENVS[env_name].bins_space(bins)


def train(env_name: str, bins: int | None = None, save_path: Path | str | None = None,
          value_iteration: bool = False, **kw):
    """Build the env on its reference grid, run policy iteration on the GPU, save the .npz
    (reference: each runner's train(), e.g. pendulum_cuda.py:116-130).  value_iteration=True (extension):
    fused value-iteration sweeps to the same theta instead, at most max_pi_iter slices of max_eval_iter."""
    from dynamicprogramming_amd import envs
    solver = envs.make(env_name, bins, **kw)
    t0 = time.perf_counter()
    if value_iteration:
        delta = float("inf")
        for _ in range(solver.config.max_pi_iter):
            delta = solver.value_iteration()
            if delta < solver.config.theta:
                break
        solver.stats["stable"] = bool(delta < solver.config.theta)
        solver._pull_tensors_from_gpu()
        dt = time.perf_counter() - t0
        sweeps = solver.stats["value_sweeps"]
        print(f"[{env_name}] value iteration: {sweeps} sweeps in {dt:.2f} s  "
              f"({solver.n_states * solver.n_actions * sweeps / dt:.3e} backups/s), residual {delta:.3e}, "
              f"converged={solver.stats['stable']}")
    else:
        solver.run()
        dt = time.perf_counter() - t0
        st = solver.stats
        backups = solver.n_states * (st["eval_sweeps"] + st["improve_sweeps"] * solver.n_actions)
        print(f"[{env_name}] {st['pi_iterations']} PI iterations, {st['eval_sweeps']} eval sweeps, "
              f"{st['improve_sweeps']} improve sweeps in {dt:.2f} s  ({backups / dt:.3e} backups/s), "
              f"stable={st.get('stable')}")
    if save_path is not None:
        solver.save(save_path)
    return solver


def build_parser(env_name: str, default_save: str) -> argparse.ArgumentParser:
    from dynamicprogramming_amd import envs
    cls = envs.ENVS[env_name]
    p = argparse.ArgumentParser(description=(cls.__doc__ or env_name).strip().splitlines()[0])
    p.add_argument("--render", action="store_true", help="(rollout harness; accepted, ignored)")
    p.add_argument("--random", type=int, nargs="?", const=5, default=None, metavar="N",
                   help="(rollout harness) random-policy baseline: no training, like the reference")
    p.add_argument("--record", type=Path, default=None, metavar="PATH", help="(rollout harness; accepted, ignored)")
    p.add_argument("--episodes", type=int, default=5, help="(rollout harness; accepted, ignored)")
    p.add_argument("--steps", type=int, default=1000, help="(rollout harness; accepted, ignored)")
    if env_name == "overhead_crane":
        p.add_argument("--start-x", type=float, default=2.5, help="(rollout harness; accepted, ignored)")
        p.add_argument("--target-x", type=float, default=-2.5,
                       help="target trolley position (m), compiled into the dynamics (default: -2.5)")
    p.add_argument("--bins", type=int, default=cls.DEFAULT_BINS,
                   help=f"bins per dimension (default: {cls.DEFAULT_BINS})")
    p.add_argument("--seed", type=int, default=42, help="(rollout harness; accepted, ignored)")
    p.add_argument("--no-plot", action="store_true", help="(plots; accepted, ignored)")
    p.add_argument("--retrain", action="store_true", help="force retraining even if a saved policy exists")
    p.add_argument("--save-path", type=Path, default=Path(default_save))
    p.add_argument("--value-iteration", action="store_true",
                   help="(extension) train with fused value-iteration sweeps instead of policy iteration")
    return p


def main(env_name: str, default_save: str, argv=None):
    """Reference control flow (pendulum_cuda.py:289-309): --random -> no training; otherwise load
    the archive when it exists and --retrain is absent, else train and save.  Returns the solver /
    loaded instance (None for --random)."""
    from dynamicprogramming_amd import envs
    cls = envs.ENVS[env_name]
    args = build_parser(env_name, default_save).parse_args(argv)
    ignored = [f for f, on in (("--render", args.render), ("--record", args.record is not None),
                               ("--episodes", args.episodes != 5), ("--steps", args.steps != 1000),
                               ("--seed", args.seed != 42), ("--no-plot", args.no_plot),
                               ("--start-x", getattr(args, "start_x", 2.5) != 2.5)) if on]
    if ignored:
        print(f"[{env_name}] note: {', '.join(ignored)} belong to the rollout / plot harness, which this "
              "package does not include; accepted and ignored")
    if args.random is not None:
        print(f"[{env_name}] --random runs rollouts only (no training) in the reference; nothing to do here")
        return None
    kw = {"target_x": args.target_x} if env_name == "overhead_crane" else {}
    path = Path(args.save_path).with_suffix(".npz")
    if path.exists() and not args.retrain:
        print(f"[+] Loading existing policy from {path}")
        pi = cls.load(path)
        print(f"[{env_name}] {pi.n_states:,} states, {pi.n_actions} actions; use --retrain to recompute")
        return pi
    print("[*] Training new policy...")
    return train(env_name, args.bins, path, value_iteration=args.value_iteration, **kw)
