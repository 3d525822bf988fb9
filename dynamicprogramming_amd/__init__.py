"""
dynamicprogramming_amd — MI355X-native policy iteration on regular 2-D/4-D/6-D grids.

Host API mirrors nicoRomeroCuruchet/DynamicProgramming's ``src/cuda_policy_iteration.py``;
the Bellman-backup sweeps run as hand-written gfx950 kernels behind libpi_mi355.so.
"""
from .solver import (GPU_AVAILABLE, CudaPIConfig, CudaPolicyIteration2D, CudaPolicyIteration4D,
                     CudaPolicyIteration6D)

__all__ = ["GPU_AVAILABLE", "CudaPIConfig", "CudaPolicyIteration2D", "CudaPolicyIteration4D",
           "CudaPolicyIteration6D"]
