"""
runners/pendulum_cuda.py — train Pendulum-v1 on (theta, theta_dot); same entry point, flags and module-level names
as the reference runner (runners/pendulum_cuda.py:40-49, :116-130).

    python runners/pendulum_cuda.py [--bins N] [--retrain] [--save-path results/pendulum_cuda_policy.npz] [...]

Module surface kept from the reference runner: ``PendulumCuda`` (the env plugin, defined in
``dynamicprogramming_amd.envs``), ``BINS_PER_DIM``, ``BINS_SPACE``, ``ACTION_SPACE`` and
``train(save_path)``.  The rollout / plot / render functions of the reference runner are not part
of this package; their flags are accepted and ignored (runners/_cli.py).
"""
from pathlib import Path

try:                      # imported as runners.<name>
    from . import _cli
except ImportError:       # run as a script: runners/ is on sys.path
    import _cli

from dynamicprogramming_amd.envs import CudaPIConfig, PendulumCuda  # noqa: E402,F401

ENV = "pendulum"
DEFAULT_SAVE = "results/pendulum_cuda_policy.npz"
BINS_PER_DIM = PendulumCuda.DEFAULT_BINS
BINS_SPACE = PendulumCuda.bins_space(BINS_PER_DIM)
ACTION_SPACE = PendulumCuda.ACTIONS


def train(save_path: Path = Path(DEFAULT_SAVE), **kw) -> PendulumCuda:
    """Policy iteration on BINS_SPACE x ACTION_SPACE with the runner's own solver settings, then
    save (reference train(): config, construct, run(), save())."""
    pi = PendulumCuda(BINS_SPACE, ACTION_SPACE, CudaPIConfig(**PendulumCuda.CONFIG), **kw)
    pi.run()
    pi.save(save_path)
    return pi


if __name__ == "__main__":
    _cli.main(ENV, DEFAULT_SAVE)
