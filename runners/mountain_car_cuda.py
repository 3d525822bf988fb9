"""
runners/mountain_car_cuda.py — train MountainCar-v0 on (position, velocity); same entry point, flags and module-level names
as the reference runner (runners/mountain_car_cuda.py).

    python runners/mountain_car_cuda.py [--bins N] [--retrain] [--save-path results/mountain_car_cuda_policy.npz] [...]

Module surface kept from the reference runner: ``MountainCarCuda`` (the env plugin, defined in
``dynamicprogramming_amd.envs``), ``BINS_PER_DIM``, ``BINS_SPACE``, ``ACTION_SPACE`` and
``train(save_path)``.  The rollout / plot / render functions of the reference runner are not part
of this package; their flags are accepted and ignored (runners/_cli.py).
"""
from pathlib import Path

try:                      # imported as runners.<name>
    from . import _cli
except ImportError:       # run as a script: runners/ is on sys.path
    import _cli

from dynamicprogramming_amd.envs import CudaPIConfig, MountainCarCuda  # noqa: E402,F401

ENV = "mountain_car"
DEFAULT_SAVE = "results/mountain_car_cuda_policy.npz"
BINS_PER_DIM = MountainCarCuda.DEFAULT_BINS
BINS_SPACE = MountainCarCuda.bins_space(BINS_PER_DIM)
ACTION_SPACE = MountainCarCuda.ACTIONS


def train(save_path: Path = Path(DEFAULT_SAVE), **kw) -> MountainCarCuda:
    """Policy iteration on BINS_SPACE x ACTION_SPACE with the runner's own solver settings, then
    save (reference train(): config, construct, run(), save())."""
    pi = MountainCarCuda(BINS_SPACE, ACTION_SPACE, CudaPIConfig(**MountainCarCuda.CONFIG), **kw)
    pi.run()
    pi.save(save_path)
    return pi


if __name__ == "__main__":
    _cli.main(ENV, DEFAULT_SAVE)
