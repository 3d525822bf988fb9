"""
runners/double_cartpole_swingup_cuda.py — train double cart-pole swing-up, 6-D; same entry point, flags and module-level names
as the reference runner (runners/double_cartpole_swingup_cuda.py:62-74).

    python runners/double_cartpole_swingup_cuda.py [--bins N] [--retrain] [--save-path results/double_cartpole_swingup_cuda_policy.npz] [...]

Module surface kept from the reference runner: ``DoubleCartPoleSwingUpCuda`` (the env plugin, defined in
``dynamicprogramming_amd.envs``), ``BINS_PER_DIM``, ``BINS_SPACE``, ``ACTION_SPACE`` and
``train(save_path)``.  The rollout / plot / render functions of the reference runner are not part
of this package; their flags are accepted and ignored (runners/_cli.py).
"""
from pathlib import Path

try:                      # imported as runners.<name>
    from . import _cli
except ImportError:       # run as a script: runners/ is on sys.path
    import _cli

from dynamicprogramming_amd.envs import CudaPIConfig, DoubleCartPoleSwingUpCuda  # noqa: E402,F401

ENV = "double_cartpole_swingup"
DEFAULT_SAVE = "results/double_cartpole_swingup_cuda_policy.npz"
BINS_PER_DIM = DoubleCartPoleSwingUpCuda.DEFAULT_BINS
BINS_SPACE = DoubleCartPoleSwingUpCuda.bins_space(BINS_PER_DIM)
ACTION_SPACE = DoubleCartPoleSwingUpCuda.ACTIONS


def train(save_path: Path = Path(DEFAULT_SAVE), **kw) -> DoubleCartPoleSwingUpCuda:
    """Policy iteration on BINS_SPACE x ACTION_SPACE with the runner's own solver settings, then
    save (reference train(): config, construct, run(), save())."""
    pi = DoubleCartPoleSwingUpCuda(BINS_SPACE, ACTION_SPACE, CudaPIConfig(**DoubleCartPoleSwingUpCuda.CONFIG), **kw)
    pi.run()
    pi.save(save_path)
    return pi


if __name__ == "__main__":
    _cli.main(ENV, DEFAULT_SAVE)
