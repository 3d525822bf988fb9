"""
runners/cartpole_swingup_cuda.py — train cart-pole swing-up; reference runner runners/cartpole_swingup_cuda.py.

    python runners/cartpole_swingup_cuda.py [--bins N] [--retrain] [--save-path results/cartpole_swingup_cuda_policy.npz]

The env plugin (dynamics string, grid, actions, solver settings) is
``dynamicprogramming_amd.envs.CartPoleSwingUpCuda``; this script is only the entry point.
"""
from _cli import main, train  # noqa: F401  (runners/ is on sys.path when run as a script)

from dynamicprogramming_amd.envs import CartPoleSwingUpCuda  # noqa: E402,F401  re-exported for `from runners...`

ENV = "cartpole_swingup"

if __name__ == "__main__":
    main(ENV, "results/cartpole_swingup_cuda_policy.npz")
