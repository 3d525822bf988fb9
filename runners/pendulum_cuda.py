"""
runners/pendulum_cuda.py — train Pendulum-v1 on (theta, theta_dot); reference runner runners/pendulum_cuda.py.

    python runners/pendulum_cuda.py [--bins N] [--retrain] [--save-path results/pendulum_cuda_policy.npz]

The env plugin (dynamics string, grid, actions, solver settings) is
``dynamicprogramming_amd.envs.PendulumCuda``; this script is only the entry point.
"""
from _cli import main, train  # noqa: F401  (runners/ is on sys.path when run as a script)

from dynamicprogramming_amd.envs import PendulumCuda  # noqa: E402,F401  re-exported for `from runners...`

ENV = "pendulum"

if __name__ == "__main__":
    main(ENV, "results/pendulum_cuda_policy.npz")
