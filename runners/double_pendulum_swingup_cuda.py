"""
runners/double_pendulum_swingup_cuda.py — train two-link pendulum swing-up (BASELINE headline grid); reference runner runners/double_pendulum_swingup_cuda.py.

    python runners/double_pendulum_swingup_cuda.py [--bins N] [--retrain] [--save-path results/double_pendulum_swingup_cuda_policy.npz]

The env plugin (dynamics string, grid, actions, solver settings) is
``dynamicprogramming_amd.envs.DoublePendulumSwingUpCuda``; this script is only the entry point.
"""
from _cli import main, train  # noqa: F401  (runners/ is on sys.path when run as a script)

from dynamicprogramming_amd.envs import DoublePendulumSwingUpCuda  # noqa: E402,F401  re-exported for `from runners...`

ENV = "double_pendulum_swingup"

if __name__ == "__main__":
    main(ENV, "results/double_pendulum_swingup_cuda_policy.npz")
