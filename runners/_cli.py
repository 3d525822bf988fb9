"""
Shared command line of the runner scripts (the reference's runners share one argparse
surface, README.md:323-347; the flags that concern the solver are kept: --bins, --retrain,
--save-path; rollout / rendering flags belong to the out-of-scope evaluation harness).
"""
from __future__ import annotations

import argparse
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def train(env_name: str, bins: int | None = None, save_path: Path | str | None = None, **kw):
    """Build the env on its reference grid, run policy iteration on the GPU, save the .npz."""
    from dynamicprogramming_amd import envs
    solver = envs.make(env_name, bins, **kw)
    t0 = time.perf_counter()
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


def main(env_name: str, default_save: str) -> None:
    from dynamicprogramming_amd import envs
    cls = envs.ENVS[env_name]
    ap = argparse.ArgumentParser(description=cls.__doc__)
    ap.add_argument("--bins", type=int, default=cls.DEFAULT_BINS, help="grid points per dimension")
    ap.add_argument("--retrain", action="store_true", help="train even if the policy file exists")
    ap.add_argument("--save-path", type=Path, default=Path(default_save))
    args = ap.parse_args()
    path = Path(args.save_path).with_suffix(".npz")
    if path.exists() and not args.retrain:
        pi = cls.load(path)
        print(f"[{env_name}] loaded {path} ({pi.n_states:,} states); use --retrain to recompute")
        return
    train(env_name, args.bins, path)
