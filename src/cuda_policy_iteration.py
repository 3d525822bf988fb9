"""
Drop-in module path of the reference solver (``from src.cuda_policy_iteration import
CudaPolicyIteration2D, CudaPolicyIteration4D, CudaPolicyIteration6D, CudaPIConfig``,
e.g. runners/pendulum_cuda.py:36 of the reference).  Everything is implemented in
``dynamicprogramming_amd.solver``; this file only re-exports the public names.
"""
from dynamicprogramming_amd.solver import (  # noqa: F401
    GPU_AVAILABLE,
    CudaPIConfig,
    CudaPolicyIteration2D,
    CudaPolicyIteration4D,
    CudaPolicyIteration6D,
)
