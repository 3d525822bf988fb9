"""
runners/double_pendulum_swingup_cuda.py — train base-actuated double pendulum swing-up (BASELINE config C4 at --bins 80); same entry point, flags and module-level names
as the reference runner (runners/double_pendulum_swingup_cuda.py:53-66, :285-297).

    python runners/double_pendulum_swingup_cuda.py [--bins N] [--retrain] [--save-path results/double_pendulum_swingup_cuda_policy.npz] [...]

Module surface kept from the reference runner: ``DoublePendulumSwingUpCuda`` (the env plugin, defined in
``dynamicprogramming_amd.envs``), ``BINS_PER_DIM``, ``BINS_SPACE``, ``ACTION_SPACE`` and
``train(save_path)``.  The rollout / plot / render functions of the reference runner are not part
of this package; their flags are accepted and ignored (runners/_cli.py).
"""
from pathlib import Path

try:                      # imported as runners.<name>
    from . import _cli
except ImportError:       # run as a script: runners/ is on sys.path
    import _cli

from dynamicprogramming_amd.envs import CudaPIConfig, DoublePendulumSwingUpCuda  # noqa: E402,F401

ENV = "double_pendulum_swingup"
DEFAULT_SAVE = "results/double_pendulum_swingup_cuda_policy.npz"
BINS_PER_DIM = DoublePendulumSwingUpCuda.DEFAULT_BINS
BINS_SPACE = DoublePendulumSwingUpCuda.bins_space(BINS_PER_DIM)
ACTION_SPACE = DoublePendulumSwingUpCuda.ACTIONS


def train(save_path: Path = Path(DEFAULT_SAVE), **kw) -> DoublePendulumSwingUpCuda:
    """Policy iteration on BINS_SPACE x ACTION_SPACE with the runner's own solver settings, then
    save (reference train(): config, construct, run(), save())."""
    pi = DoublePendulumSwingUpCuda(BINS_SPACE, ACTION_SPACE, CudaPIConfig(**DoublePendulumSwingUpCuda.CONFIG), **kw)
    pi.run()
    pi.save(save_path)
    return pi


if __name__ == "__main__":
    _cli.main(ENV, DEFAULT_SAVE)
