"""tools/time_runs.py — wall time and sweeps-to-converge of full run() calls (GPU box)."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from dynamicprogramming_amd import envs
DEFAULTS = [] if "--only" in sys.argv else [("pendulum", 50), ("pendulum", 200), ("mountain_car", 200), ("continuous_mountain_car", 200),
                   ("cartpole", 30), ("double_pendulum_swingup", 15), ("double_pendulum_swingup", 25)]
REPEAT = int(sys.argv[sys.argv.index("--repeat") + 1]) if "--repeat" in sys.argv else 1   # best of N fresh solvers
for name, bins in DEFAULTS + [(a, int(b)) for a, b in (x.split("@") for x in sys.argv[1:] if "@" in x)]:
    cls = envs.ENVS[name]
    best = None
    for _ in range(REPEAT):
        if (name, bins) == ("pendulum", 50):
            import numpy as np
            solver = cls(cls.bins_space(50), np.linspace(-2, 2, 11, dtype=np.float32), envs.CudaPIConfig(**cls.CONFIG))
        else:
            solver = envs.make(name, bins)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        solver.run()
        dt = time.perf_counter() - t0
        if best is None or dt < best[0]:
            best = (dt, solver)
    dt, solver = best
    st = solver.stats
    bk = solver.n_states * (st["eval_sweeps"] + st["improve_sweeps"] * solver.n_actions)
    print(f"{name:26s} bins={bins:4d} n={solver.n_states:9d}  PI iters {st['pi_iterations']:3d}  eval sweeps {st['eval_sweeps']:6d}  "
          f"stable={st.get('stable')}  {dt:7.2f} s  {bk / dt:.3e} backups/s  {dt / max(st['eval_sweeps'], 1) * 1e6:.1f} us/sweep", flush=True)
