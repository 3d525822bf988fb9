"""
utils/barycentric.py — inference-time interpolation on the solver's grids (numpy, no numba).

Same module path and call signatures as the reference's CPU helper
(/root/reference/utils/barycentric.py): ``get_barycentric_weights_and_indices`` (:12-73)
and ``get_optimal_action`` (:76-108), so rollout code written against the reference
(``from utils.barycentric import get_optimal_action``) runs unchanged on policies trained
here or there.  Vectorised over the batch instead of JIT-compiled loops.

Batched GPU path (keyword opt-in): ``device="cuda:0"`` on either function —
or a ``DevicePolicy`` object that keeps the policy table on the GPU — runs the same arithmetic as
ONE hand-written HIP kernel over the whole batch (libpi_mi355.so, ``pi_infer_query``): indices and
weights bit-identical to the functions below, the action summed in float32 in ascending corner order.
Without the keyword nothing here touches the GPU or the native library.

Semantics kept from the reference helper (they differ slightly from the training kernels):
the POINT is clamped to the bounds (not the cell coordinate), cell widths are float64
``(hi - lo) / (shape - 1)``, corners follow the rows of ``corner_bits`` (MSB-first
``itertools.product``), weights are float64 products stored as float32, indices int32.
"""
from __future__ import annotations

import numpy as np


class DevicePolicy:
    """A trained policy resident on the GPU for batched queries: ``DevicePolicy(policy, action_space,
    bounds_low, bounds_high, grid_shape, strides, corner_bits, device="cuda:0")``; calling it with
    states (m, D) returns the m interpolated actions (float32 numpy array); ``weights_and_indices``
    returns what ``get_barycentric_weights_and_indices`` returns.  ``policy`` may be None when only
    weights and indices are wanted.  Raises if the native library or a GPU is missing (no fallback)."""

    def __init__(self, policy, action_space, bounds_low, bounds_high, grid_shape, strides, corner_bits,
                 device="cuda:0"):
        import torch
        from dynamicprogramming_amd import _native
        if not torch.cuda.is_available():
            raise RuntimeError("DevicePolicy needs a ROCm GPU (torch.cuda.is_available() is False)")
        self._torch = torch
        self.device = torch.device(device)
        self.D = len(np.asarray(grid_shape))
        self.n_corners = len(np.asarray(corner_bits))
        self._engine = _native.InferenceEngine(bounds_low, bounds_high, grid_shape, strides, corner_bits,
                                               device=self.device.index or 0)
        self._has_policy = policy is not None
        if self._has_policy:
            self._engine.set_policy(policy, action_space)

    def _points(self, states):
        pts = np.ascontiguousarray(np.atleast_2d(states), dtype=np.float32)
        assert pts.shape[1] == self.D, f"states must be (m, {self.D})"
        return self._torch.from_numpy(pts).to(self.device), len(pts)

    def _stream(self):
        return self._torch.cuda.current_stream(self.device).cuda_stream

    def __call__(self, states):
        """states (m, D): a numpy array -> numpy actions (m,), or a float32 torch tensor already on this
        device -> a torch tensor on the device (no host round trip: closed-loop rollouts on the GPU)."""
        if not self._has_policy:
            raise RuntimeError("this DevicePolicy was built without a policy table")
        on_device = self._torch.is_tensor(states)
        if on_device:
            if states.device != self.device or states.dtype != self._torch.float32 or states.dim() != 2 \
                    or states.shape[1] != self.D:
                raise ValueError(f"device states must be a float32 (m, {self.D}) tensor on {self.device}")
            d_pts, m = states.contiguous(), states.shape[0]
        else:
            d_pts, m = self._points(states)
        out = self._torch.empty(m, dtype=self._torch.float32, device=self.device)
        self._engine.query(d_pts.data_ptr(), m, d_actions=out.data_ptr(), stream=self._stream())
        return out if on_device else out.cpu().numpy()

    def weights_and_indices(self, states):
        d_pts, m = self._points(states)
        w = self._torch.empty((m, self.n_corners), dtype=self._torch.float32, device=self.device)
        idx = self._torch.empty((m, self.n_corners), dtype=self._torch.int32, device=self.device)
        self._engine.query(d_pts.data_ptr(), m, d_weights=w.data_ptr(), d_indices=idx.data_ptr(), stream=self._stream())
        return w.cpu().numpy(), idx.cpu().numpy()

    def close(self) -> None:
        self._engine.close()


def get_barycentric_weights_and_indices(points, bounds_low, bounds_high, grid_shape, strides,
                                        corner_bits, *, device=None):
    """
    points (n, D) float32 -> (weights (n, 2^D) float32 summing to 1, indices (n, 2^D) int32).
    ``device="cuda:0"``: the whole batch in one HIP kernel (same bits).
    """
    if device is not None:
        dp = DevicePolicy(None, None, bounds_low, bounds_high, grid_shape, strides, corner_bits, device=device)
        try:
            return dp.weights_and_indices(points)
        finally:
            dp.close()
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
                       corner_bits, *, device=None):
    """Interpolated action at a continuous state: weights @ action VALUES of the surrounding
    grid nodes' greedy actions (reference :96-108).  ``device="cuda:0"``: ``state`` may be a batch
    (m, D) and the m actions come from one HIP kernel launch (for repeated queries keep a
    ``DevicePolicy`` instead: it uploads the policy table once)."""
    if device is not None:
        dp = DevicePolicy(policy, action_space, bounds_low, bounds_high, grid_shape, strides, corner_bits,
                          device=device)
        try:
            out = dp(state)
        finally:
            dp.close()
        return out[0] if np.ndim(state) == 1 else out
    state_2d = np.atleast_2d(state).astype(np.float32)
    lambdas, flat = get_barycentric_weights_and_indices(state_2d, bounds_low, bounds_high,
                                                        grid_shape, strides, corner_bits)
    lambdas = lambdas.flatten()
    flat = flat.flatten()
    return lambdas @ np.asarray(action_space)[np.asarray(policy)[flat]]
