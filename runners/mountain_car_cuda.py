"""
runners/mountain_car_cuda.py — train MountainCar-v0; reference runner runners/mountain_car_cuda.py.

    python runners/mountain_car_cuda.py [--bins N] [--retrain] [--save-path results/mountain_car_cuda_policy.npz]

The env plugin (dynamics string, grid, actions, solver settings) is
``dynamicprogramming_amd.envs.MountainCarCuda``; this script is only the entry point.
"""
from _cli import main, train  # noqa: F401  (runners/ is on sys.path when run as a script)

from dynamicprogramming_amd.envs import MountainCarCuda  # noqa: E402,F401  re-exported for `from runners...`

ENV = "mountain_car"

if __name__ == "__main__":
    main(ENV, "results/mountain_car_cuda_policy.npz")
