"""utils/barycentric.py (inference-side helper, next-row f2): weights/indices agree with the
training-side interpolation up to the documented differences, and get_optimal_action
interpolates action values."""
from __future__ import annotations

import numpy as np

import oracle
from tests import helpers as H
from utils.barycentric import get_barycentric_weights_and_indices, get_optimal_action
from itertools import product


def test_weights_and_indices_match_training_interpolation():
    for name, shape in (("pendulum", (11, 7)), ("cartpole", (5, 4, 6, 3)), ("double_cartpole", (3, 4, 2, 5, 3, 2))):
        bins = H.env_bins(name, shape)
        D = len(bins)
        lo, hi, gshape, strides = oracle.grid_metadata(bins)
        bits = np.array(list(product([0, 1], repeat=D)), dtype=np.int32)
        pts = H.sample_states(np.random.default_rng(D), bins, 2000)
        w, idx = get_barycentric_weights_and_indices(pts, lo, hi, gshape, strides, bits)
        assert w.dtype == np.float32 and idx.dtype == np.int32 and w.shape == (2000, 1 << D)
        np.testing.assert_allclose(w.sum(axis=1), 1.0, atol=1e-5)
        k_idx, k_w = H.oracle_for(name).interp(pts, lo, hi, gshape, strides)
        # same cell, same weights up to rounding; corner ORDER differs (MSB-first rows here)
        V = np.random.default_rng(1).standard_normal(int(np.prod(shape)))
        a = (w.astype(np.float64) * V[idx]).sum(axis=1)
        b = (k_w.astype(np.float64) * V[k_idx]).sum(axis=1)
        ok = np.abs(a - b) < 1e-4
        assert ok.mean() > 0.995          # cell-boundary points may pick the neighbouring cell
        same_cell = np.all(np.sort(idx, axis=1) == np.sort(k_idx, axis=1), axis=1)
        assert same_cell.mean() > 0.9         # exact nodes may sit in either adjacent cell


def test_get_optimal_action_interpolates_action_values():
    bins = [np.linspace(0, 1, 3, dtype=np.float32), np.linspace(0, 1, 3, dtype=np.float32)]
    lo, hi, gshape, strides = oracle.grid_metadata(bins)
    bits = np.array(list(product([0, 1], repeat=2)), dtype=np.int32)
    policy = np.arange(9, dtype=np.int32) % 3
    actions = np.array([-1.0, 0.0, 2.0], dtype=np.float32)
    assert get_optimal_action(np.array([0.0, 0.0]), policy, actions, lo, hi, gshape, strides, bits) == actions[0]
    mid = get_optimal_action(np.array([0.25, 0.25]), policy, actions, lo, hi, gshape, strides, bits)
    want = 0.25 * (actions[policy[0]] + actions[policy[1]] + actions[policy[3]] + actions[policy[4]])
    assert abs(mid - want) < 1e-6
    out = get_optimal_action(np.array([9.0, -9.0]), policy, actions, lo, hi, gshape, strides, bits)
    assert out == actions[policy[6]]          # clamped to the (1, 0) corner node
