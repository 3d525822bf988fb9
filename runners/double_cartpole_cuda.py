"""
runners/double_cartpole_cuda.py — train double cart-pole balance (6-D); reference runner runners/double_cartpole_cuda.py.

    python runners/double_cartpole_cuda.py [--bins N] [--retrain] [--save-path results/double_cartpole_cuda_policy.npz]

The env plugin (dynamics string, grid, actions, solver settings) is
``dynamicprogramming_amd.envs.DoubleCartPoleCuda``; this script is only the entry point.
"""
from _cli import main, train  # noqa: F401  (runners/ is on sys.path when run as a script)

from dynamicprogramming_amd.envs import DoubleCartPoleCuda  # noqa: E402,F401  re-exported for `from runners...`

ENV = "double_cartpole"

if __name__ == "__main__":
    main(ENV, "results/double_cartpole_cuda_policy.npz")
