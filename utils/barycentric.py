"""
utils/barycentric.py — inference-time interpolation on the solver's grids (numpy, no numba).

Same module path and call signatures as the reference's CPU helper
(/root/reference/utils/barycentric.py): ``get_barycentric_weights_and_indices`` (:12-73)
and ``get_optimal_action`` (:76-108), so rollout code written against the reference
(``from utils.barycentric import get_optimal_action``) runs unchanged on policies trained
here or there.  Vectorised over the batch instead of JIT-compiled loops.

Semantics kept from the reference helper (they differ slightly from the training kernels):
the POINT is clamped to the bounds (not the cell coordinate), cell widths are float64
``(hi - lo) / (shape - 1)``, corners follow the rows of ``corner_bits`` (MSB-first
``itertools.product``), weights are float64 products stored as float32, indices int32.
"""
from __future__ import annotations

import numpy as np


def get_barycentric_weights_and_indices(points, bounds_low, bounds_high, grid_shape, strides,
                                        corner_bits):
    """
    points (n, D) float32 -> (weights (n, 2^D) float32 summing to 1, indices (n, 2^D) int32).
    """
    pts = np.asarray(points)
    lo = np.asarray(bounds_low)
    hi = np.asarray(bounds_high)
    shape = np.asarray(grid_shape)
    st = np.asarray(strides).astype(np.int64)
    bits = np.asarray(corner_bits).astype(np.int64)           # (C, D)
    step = (hi - lo) / (shape - 1)                             # float64, like the reference
    p = np.maximum(lo, np.minimum(pts, hi))                    # clamp the point
    cell = (p - lo) / step
    idx = cell.astype(np.int64)                                # truncation, cell >= 0
    idx = np.where(idx >= shape - 1, shape - 2, idx)
    t = ((p - (lo + idx * step)) / step).astype(np.float32)    # (n, D)
    t64 = t.astype(np.float64)
    # w[c] = prod_d (t_d if bit else 1 - t_d), multiplied in dimension order
    w = np.ones((pts.shape[0], bits.shape[0]), dtype=np.float64)
    for d in range(pts.shape[1]):
        w = w * np.where(bits[None, :, d] == 1, t64[:, None, d], 1.0 - t64[:, None, d])
    flat = ((idx[:, None, :] + bits[None, :, :]) * st[None, None, :]).sum(axis=2)
    return w.astype(np.float32), flat.astype(np.int32)


def get_optimal_action(state, policy, action_space, bounds_low, bounds_high, grid_shape, strides,
                       corner_bits):
    """Interpolated action at a continuous state: weights @ action VALUES of the surrounding
    grid nodes' greedy actions (reference :96-108)."""
    state_2d = np.atleast_2d(state).astype(np.float32)
    lambdas, flat = get_barycentric_weights_and_indices(state_2d, bounds_low, bounds_high,
                                                        grid_shape, strides, corner_bits)
    lambdas = lambdas.flatten()
    flat = flat.flatten()
    return lambdas @ np.asarray(action_space)[np.asarray(policy)[flat]]
