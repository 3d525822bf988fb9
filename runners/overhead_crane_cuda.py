"""
runners/overhead_crane_cuda.py — train overhead crane anti-sway; reference runner runners/overhead_crane_cuda.py.

    python runners/overhead_crane_cuda.py [--bins N] [--retrain] [--save-path results/overhead_crane_cuda_policy.npz]

The env plugin (dynamics string, grid, actions, solver settings) is
``dynamicprogramming_amd.envs.OverheadCraneCuda``; this script is only the entry point.
"""
from _cli import main, train  # noqa: F401  (runners/ is on sys.path when run as a script)

from dynamicprogramming_amd.envs import OverheadCraneCuda  # noqa: E402,F401  re-exported for `from runners...`

ENV = "overhead_crane"

if __name__ == "__main__":
    main(ENV, "results/overhead_crane_cuda_policy.npz")
