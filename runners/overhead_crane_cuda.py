"""
runners/overhead_crane_cuda.py — train overhead crane positioning with sway damping; same entry point, flags and module-level names
as the reference runner (runners/overhead_crane_cuda.py).

    python runners/overhead_crane_cuda.py [--bins N] [--retrain] [--save-path results/overhead_crane_cuda_policy.npz] [...]

Module surface kept from the reference runner: ``OverheadCraneCuda`` (the env plugin, defined in
``dynamicprogramming_amd.envs``), ``BINS_PER_DIM``, ``BINS_SPACE``, ``ACTION_SPACE`` and
``train(save_path)``.  The rollout / plot / render functions of the reference runner are not part
of this package; their flags are accepted and ignored (runners/_cli.py).
"""
from pathlib import Path

try:                      # imported as runners.<name>
    from . import _cli
except ImportError:       # run as a script: runners/ is on sys.path
    import _cli

from dynamicprogramming_amd.envs import CudaPIConfig, OverheadCraneCuda  # noqa: E402,F401

ENV = "overhead_crane"
DEFAULT_SAVE = "results/overhead_crane_cuda_policy.npz"
BINS_PER_DIM = OverheadCraneCuda.DEFAULT_BINS
BINS_SPACE = OverheadCraneCuda.bins_space(BINS_PER_DIM)
ACTION_SPACE = OverheadCraneCuda.ACTIONS


def train(save_path: Path = Path(DEFAULT_SAVE), target_x: float = -2.5, **kw) -> OverheadCraneCuda:
    """Policy iteration on BINS_SPACE x ACTION_SPACE with the runner's own solver settings, then
    save (reference train(save_path, target_x=-2.5): config, construct, run(), save() —
    runners/overhead_crane_cuda.py:252)."""
    pi = OverheadCraneCuda(BINS_SPACE, ACTION_SPACE, CudaPIConfig(**OverheadCraneCuda.CONFIG),
                           target_x=target_x, **kw)
    pi.run()
    pi.save(save_path)
    return pi


if __name__ == "__main__":
    _cli.main(ENV, DEFAULT_SAVE)
