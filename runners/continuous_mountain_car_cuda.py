"""
runners/continuous_mountain_car_cuda.py — train MountainCarContinuous-v0; same entry point, flags and module-level names
as the reference runner (runners/continuous_mountain_car_cuda.py).

    python runners/continuous_mountain_car_cuda.py [--bins N] [--retrain] [--save-path results/continuous_mountain_car_cuda_policy.npz] [...]

Module surface kept from the reference runner: ``ContinuousMountainCarCuda`` (the env plugin, defined in
``dynamicprogramming_amd.envs``), ``BINS_PER_DIM``, ``BINS_SPACE``, ``ACTION_SPACE`` and
``train(save_path)``.  The rollout / plot / render functions of the reference runner are not part
of this package; their flags are accepted and ignored (runners/_cli.py).
"""
from pathlib import Path

try:                      # imported as runners.<name>
    from . import _cli
except ImportError:       # run as a script: runners/ is on sys.path
    import _cli

from dynamicprogramming_amd.envs import CudaPIConfig, ContinuousMountainCarCuda  # noqa: E402,F401

ENV = "continuous_mountain_car"
DEFAULT_SAVE = "results/continuous_mountain_car_cuda_policy.npz"
BINS_PER_DIM = ContinuousMountainCarCuda.DEFAULT_BINS
BINS_SPACE = ContinuousMountainCarCuda.bins_space(BINS_PER_DIM)
ACTION_SPACE = ContinuousMountainCarCuda.ACTIONS


def train(save_path: Path = Path(DEFAULT_SAVE), **kw) -> ContinuousMountainCarCuda:
    """Policy iteration on BINS_SPACE x ACTION_SPACE with the runner's own solver settings, then
    save (reference train(): config, construct, run(), save())."""
    pi = ContinuousMountainCarCuda(BINS_SPACE, ACTION_SPACE, CudaPIConfig(**ContinuousMountainCarCuda.CONFIG), **kw)
    pi.run()
    pi.save(save_path)
    return pi


if __name__ == "__main__":
    _cli.main(ENV, DEFAULT_SAVE)
