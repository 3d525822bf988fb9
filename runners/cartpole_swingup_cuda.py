"""
runners/cartpole_swingup_cuda.py — train cart-pole swing-up (BASELINE config C3 at --bins 50); same entry point, flags and module-level names
as the reference runner (runners/cartpole_swingup_cuda.py:45-53).

    python runners/cartpole_swingup_cuda.py [--bins N] [--retrain] [--save-path results/cartpole_swingup_cuda_policy.npz] [...]

Module surface kept from the reference runner: ``CartPoleSwingUpCuda`` (the env plugin, defined in
``dynamicprogramming_amd.envs``), ``BINS_PER_DIM``, ``BINS_SPACE``, ``ACTION_SPACE`` and
``train(save_path)``.  The rollout / plot / render functions of the reference runner are not part
of this package; their flags are accepted and ignored (runners/_cli.py).
"""
from pathlib import Path

try:                      # imported as runners.<name>
    from . import _cli
except ImportError:       # run as a script: runners/ is on sys.path
    import _cli

from dynamicprogramming_amd.envs import CudaPIConfig, CartPoleSwingUpCuda  # noqa: E402,F401

ENV = "cartpole_swingup"
DEFAULT_SAVE = "results/cartpole_swingup_cuda_policy.npz"
BINS_PER_DIM = CartPoleSwingUpCuda.DEFAULT_BINS
BINS_SPACE = CartPoleSwingUpCuda.bins_space(BINS_PER_DIM)
ACTION_SPACE = CartPoleSwingUpCuda.ACTIONS


def train(save_path: Path = Path(DEFAULT_SAVE), **kw) -> CartPoleSwingUpCuda:
    """Policy iteration on BINS_SPACE x ACTION_SPACE with the runner's own solver settings, then
    save (reference train(): config, construct, run(), save())."""
    pi = CartPoleSwingUpCuda(BINS_SPACE, ACTION_SPACE, CudaPIConfig(**CartPoleSwingUpCuda.CONFIG), **kw)
    pi.run()
    pi.save(save_path)
    return pi


if __name__ == "__main__":
    _cli.main(ENV, DEFAULT_SAVE)
