"""utils/barycentric.py (inference-side helper): pinned bit for bit to
vectors produced by the reference's own function (tests/golden/barycentric_utils.npz, generated
by tests/golden/make_barycentric_golden.py in the build container); agrees with the training-side
interpolation up to the documented differences; get_optimal_action interpolates action values."""
from __future__ import annotations

import numpy as np

import oracle
from tests import helpers as H
from utils.barycentric import get_barycentric_weights_and_indices, get_optimal_action
from itertools import product


GOLD = np.load(H.GOLDEN / "barycentric_utils.npz")


def test_matches_the_reference_function_bit_for_bit():
    """Same flat indices and bit-equal float32 weights as /root/reference/utils/barycentric.py:15-77
    (numba typing: float64 products rounded once) on 2 000 seeded points per dimension count,
    including out-of-range, exact-node and last-cell points; within 2 ulp of what the same function
    body gives under numpy-2 scalar promotion; get_optimal_action (:80-112) equal to 1e-12."""
    for D in (2, 4, 6):
        name, shape = str(GOLD[f"d{D}_env"]), tuple(int(x) for x in GOLD[f"d{D}_shape"])
        bins = H.env_bins(name, shape)
        lo, hi, gshape, strides = oracle.grid_metadata(bins)
        bits = np.array(list(product([0, 1], repeat=D)), dtype=np.int32)
        pts = GOLD[f"d{D}_points"]
        assert np.array_equal(pts.view(np.uint32), H.sample_states(np.random.default_rng(100 + D), bins, 2000).view(np.uint32))
        w, idx = get_barycentric_weights_and_indices(pts, lo, hi, gshape, strides, bits)
        assert np.array_equal(idx, GOLD[f"d{D}_indices"])
        assert np.array_equal(w.view(np.uint32), GOLD[f"d{D}_weights_numba"].view(np.uint32))
        plain = GOLD[f"d{D}_weights_plain"]
        ulp = np.spacing(np.maximum(np.abs(plain), np.float32(1e-30)))
        assert np.all(np.abs(w.astype(np.float64) - plain) <= 2.0 * D * ulp)
        policy, actions = GOLD[f"d{D}_policy"], GOLD[f"d{D}_actions"]
        got = np.array([get_optimal_action(pts[k], policy, actions, lo, hi, gshape, strides, bits)
                        for k in range(300)], dtype=np.float64)
        np.testing.assert_allclose(got, GOLD[f"d{D}_optimal_action"], rtol=0, atol=1e-12)


def test_package_reexports_like_the_reference():
    """`from utils import get_optimal_action` works as against the reference's utils/__init__.py."""
    import utils
    assert utils.get_optimal_action is get_optimal_action
    assert utils.get_barycentric_weights_and_indices is get_barycentric_weights_and_indices


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
