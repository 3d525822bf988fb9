"""Shared helpers for the test-suite (grids, masks, bit-level comparisons, checker backend)."""
from __future__ import annotations

from pathlib import Path

import numpy as np

import oracle
from dynamicprogramming_amd import envs
from dynamicprogramming_amd.solver import _CudaPolicyIterationBase

GOLDEN = Path(__file__).resolve().parent / "golden"
ENV_NAMES = list(envs.ENVS)


def env_bins(name: str, shape) -> list[np.ndarray]:
    """The env's reference grid ranges with shape[d] points in dimension d."""
    cls = envs.ENVS[name]
    out = []
    for d, g in enumerate(shape):
        space = cls.bins_space(int(g))
        out.append(np.asarray(list(space.values())[d], dtype=np.float32))
    return out


def env_bins_space(name: str, shape) -> dict:
    cls = envs.ENVS[name]
    keys = list(cls.bins_space(2).keys())
    return dict(zip(keys, env_bins(name, shape)))


def terminal_mask(name: str, states: np.ndarray):
    cls = envs.ENVS[name]
    inst = object.__new__(cls)
    if name == "overhead_crane":
        inst.target_x = 0.0
    if cls._terminal_fn is _CudaPolicyIterationBase._terminal_fn:
        return np.zeros(len(states), dtype=bool), 0.0
    mask, val = cls._terminal_fn(inst, states)
    return np.asarray(mask, dtype=bool), float(val)


def sample_states(rng, bins, m):
    """Seeded query points: inside the grid, beyond its borders, on nodes, in the edge cells."""
    D = len(bins)
    lo = np.array([b.min() for b in bins], dtype=np.float64)
    hi = np.array([b.max() for b in bins], dtype=np.float64)
    span = hi - lo
    pts = lo + span * rng.uniform(-0.15, 1.15, size=(m, D))
    k = m // 8
    nodes = np.stack([b[rng.integers(0, len(b), size=k)] for b in bins], axis=1)
    pts[:k] = nodes
    pts[k:2 * k] = hi - span * rng.uniform(0, 1e-3, size=(k, D))
    pts[2 * k:3 * k] = lo + span * rng.uniform(0, 1e-3, size=(k, D))
    pts[3 * k] = hi
    pts[3 * k + 1] = lo
    return pts.astype(np.float32)


def bits_equal(a, b) -> bool:
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    if a.shape != b.shape:
        return False
    if a.dtype == np.float32 and b.dtype == np.float32:      # bit patterns: +0 != -0, NaN payloads count
        return bool(np.array_equal(a.view(np.uint32), b.view(np.uint32)))
    return bool(np.array_equal(a, b))


def assert_bits_equal(a, b, what=""):
    a, b = np.asarray(a), np.asarray(b)
    if not bits_equal(a, b):
        bad = np.flatnonzero((a != b).ravel())
        raise AssertionError(f"{what}: {len(bad)} of {a.size} entries differ; first at {bad[:5]}: "
                             f"{a.ravel()[bad[:5]]} vs {b.ravel()[bad[:5]]}")


def golden(name: str):
    return np.load(GOLDEN / f"{name}.npz")


def oracle_for(name: str, libm: bool = False) -> oracle.OracleLib:
    return oracle.build(envs.ENVS[name]._D, envs.dynamics_source(name), libm=libm)


def with_checker_backend(cls):
    """Subclass of a solver class whose sweeps run on the CPU checker below instead of the HIP
    backend — for HOST-LOGIC tests only.  The product has exactly one backend and offers no hook to
    swap it: while such a solver is being constructed the test patches the module's own names
    (``solver.HipSweepBackend``, ``solver.GPU_AVAILABLE``) and restores them afterwards."""
    def __init__(self, *args, **kwargs):
        from dynamicprogramming_amd import solver as S
        saved = (S.HipSweepBackend, S.GPU_AVAILABLE)
        S.HipSweepBackend, S.GPU_AVAILABLE = OracleSweepBackend, True
        try:
            cls.__init__(self, *args, **kwargs)
        finally:
            S.HipSweepBackend, S.GPU_AVAILABLE = saved
    return type(cls.__name__ + "OnChecker", (cls,), {"__init__": __init__})


class OracleSweepBackend:
    """CPU stand-in for HipSweepBackend, for HOST-LOGIC tests only (run-loop semantics,
    sharding over gloo ranks).  Lives under tests/ so the product can never pick it up; tests
    install it with ``with_checker_backend``."""

    def __init__(self, D, grid_shape, lo, hi, bins, actions, dynamics_src, device=None, order=None):
        import torch
        self.torch = torch
        self.device = torch.device("cpu")
        # memory order of the dimensions (solver.MEMORY_ORDER): the "device" arrays the solver hands over are in that
        # order; the checker itself works in the user's order, whole grids only
        self.order = None if order is None else tuple(int(d) for d in order)
        self.lib = oracle.build(int(D), dynamics_src)
        self.lo, self.hi = np.asarray(lo, np.float32), np.asarray(hi, np.float32)
        self.shape = np.asarray(grid_shape, np.int32)
        st = np.ones(int(D), dtype=np.int64)
        for d in range(int(D) - 2, -1, -1):
            st[d] = st[d + 1] * self.shape[d + 1]
        self.strides = st.astype(np.int32)
        self.actions = np.asarray(actions, np.float32)
        self.states = oracle.states_from_bins(bins)
        self.n = len(self.states)
        self.calls = {"eval": 0, "improve": 0}

    def to_memory(self, a):
        if self.order is None:
            return a
        b = a.reshape([int(g) for g in self.shape])
        b = b.permute(*self.order).contiguous() if hasattr(b, "permute") else np.ascontiguousarray(b.transpose(self.order))
        return b.reshape(-1)

    def to_user(self, a):
        if self.order is None:
            return a
        inv = [self.order.index(d) for d in range(len(self.order))]
        b = a.reshape([int(self.shape[d]) for d in self.order])
        b = b.permute(*inv).contiguous() if hasattr(b, "permute") else np.ascontiguousarray(b.transpose(inv))
        return b.reshape(-1)

    def _u(self, t):
        """numpy view (user's order) of a whole-grid device tensor; a copy when the memory order differs."""
        a = t.numpy()[: self.n]
        return a if self.order is None else self.to_user(a)

    def _whole(self, s_begin, s_end):
        assert self.order is None or (s_begin == 0 and s_end == self.n), "memory orders: whole-grid sweeps only"

    def _mask(self, term):
        """The solver passes None for a grid without terminal states (the product then streams no mask)."""
        return np.zeros(self.n, dtype=np.uint8) if term is None else self._u(term)

    def eval_sweeps(self, Va, Vb, policy, term, s_begin, s_end, gamma, n_sweeps, d_delta,
                    rebuild=True):
        n = self.n
        self._whole(s_begin, s_end)
        for i in range(n_sweeps):
            src, dst = (Vb, Va) if (i & 1) else (Va, Vb)
            out = np.ascontiguousarray(self._u(dst))
            _, delta = self.lib.eval_sweep(self.states, self.actions, self._u(policy),
                                           self._u(src), self._mask(term), self.lo, self.hi,
                                           self.shape, self.strides, gamma, s_begin, s_end,
                                           out=out)
            dst.numpy()[:n] = self.to_memory(out)
            self.calls["eval"] += 1
            if d_delta is not None and i == n_sweeps - 1:
                d_delta[0] = delta

    def reach_planes(self, term, s_begin, s_end, n_planes):
        """CPU restatement of pi_reach_planes: planes of every successor cell (+1), any action."""
        n = self.n
        out = np.zeros(n_planes, dtype=bool)
        live = ~self._mask(term)[s_begin:s_end].astype(bool)
        st = self.states[s_begin:s_end][live]
        for a in self.actions:
            nxt, _, done = self.lib.step(st, a)
            idx, _ = self.lib.interp(nxt[~done], self.lo, self.hi, self.shape, self.strides)
            planes = np.unique(idx // int(self.strides[0]))
            out[planes] = True
        return out

    def reach_units(self, term, s_begin, s_end, depth):
        """CPU restatement of pi_reach_units: units of the leading `depth` dimensions (planes of
        dimension 0, rows (i0, i1)) holding any corner of any successor cell, any action."""
        n = self.n
        unit = int(self.strides[depth - 1])
        out = np.zeros(n // unit, dtype=bool)
        live = ~self._mask(term)[s_begin:s_end].astype(bool)
        st = self.states[s_begin:s_end][live]
        for a in self.actions:
            nxt, _, done = self.lib.step(st, a)
            idx, _ = self.lib.interp(nxt[~done], self.lo, self.hi, self.shape, self.strides)
            out[np.unique(idx // unit)] = True
        return out

    def improve_sweep(self, V, policy, term, s_begin, s_end, gamma, d_changed):
        n = self.n
        self._whole(s_begin, s_end)
        new_pol, changed = self.lib.improve_sweep(self.states, self.actions, self._u(policy),
                                                  self._u(V), self._mask(term), self.lo, self.hi,
                                                  self.shape, self.strides, gamma, s_begin, s_end)
        if self.order is None:
            policy.numpy()[:n][s_begin:s_end] = new_pol[s_begin:s_end]
        else:
            policy.numpy()[:n] = self.to_memory(new_pol)
        self.calls["improve"] += 1
        if d_changed is not None:
            d_changed[0] = changed

    def value_sweep(self, V, Vnew, policy, term, s_begin, s_end, gamma, d_delta, d_changed):
        n = self.n
        self._whole(s_begin, s_end)
        out = np.ascontiguousarray(self._u(Vnew))
        _, new_pol, delta, changed = self.lib.value_sweep(
            self.states, self.actions, self._u(policy), self._u(V), self._mask(term), self.lo,
            self.hi, self.shape, self.strides, gamma, s_begin, s_end, out=out)
        Vnew.numpy()[:n] = self.to_memory(out)
        if self.order is None:
            policy.numpy()[:n][s_begin:s_end] = new_pol[s_begin:s_end]
        else:
            policy.numpy()[:n] = self.to_memory(new_pol)
        if d_delta is not None:
            d_delta[0] = delta
        if d_changed is not None:
            d_changed[0] = changed

    def close(self):
        pass


# -- reference-EXECUTED dynamics (tests/golden/step_python.npz) -----------------------------------------
# The reference's own float64 `_step_python` mirrors, run in the build container on seeded (state, action)
# pairs (tests/golden/make_step_python_golden.py).  Tolerances of a float32 kernel against a float64 mirror,
# written out once for the CPU and the GPU test:
STEP_PYTHON_ENVS = {            # env -> wrapped angle dimensions (compared on the circle)
    "cartpole_swingup": (2,),
    "double_pendulum_swingup": (0, 2),
    "overhead_crane": (),
    "double_cartpole": (),
    "double_cartpole_swingup": (2, 4),
}
STEP_NEXT_TOL = 1e-5            # |d next| <= tol * max(1, |next|)        (measured 2.3e-6)
STEP_REWARD_TOL = 5e-5          # |d reward| <= tol * max(1, |reward|)    (measured 9.1e-6)
STEP_MARGIN = 1e-4              # flag / reward compared only this far from a comparison threshold


def check_against_step_python(name: str, nxt, rew, done, what: str) -> dict:
    """next state / reward / terminated of an implementation of env `name` on the fixture's inputs, against what
    the reference's own `_step_python` returned for them.  Returns the measured maxima."""
    g = np.load(GOLDEN / "step_python.npz")
    r_next, r_rew = g[f"{name}_next"].astype(np.float64), g[f"{name}_reward"]
    r_term, margin = g[f"{name}_term"], g[f"{name}_margin"]
    far = margin > STEP_MARGIN
    d = np.asarray(nxt, np.float64) - r_next
    for k in STEP_PYTHON_ENVS[name]:
        d[:, k] = (d[:, k] + np.pi) % (2.0 * np.pi) - np.pi
    e_next = float(np.abs(d / np.maximum(1.0, np.abs(r_next))).max())
    assert e_next <= STEP_NEXT_TOL, f"{what} {name}: next state off by {e_next:.3g} (relative)"
    done = np.asarray(done, bool)
    bad = np.flatnonzero((done != r_term) & far)
    assert len(bad) == 0, f"{what} {name}: terminated differs at {bad[:5]} away from every threshold"
    rew = np.asarray(rew, np.float64)
    if name == "double_cartpole":
        # the reference's kernel string has reward 1 - 0.0 * xn^2 (double_cartpole_cuda.py:98,:163)
        # while its Python mirror kept 1 - 0.5 * (nx / 2.4)^2 (:234).  The kernel string is the ground truth of the
        # sweep; the mirror pins next state and flag, and the reward through the mirror's own formula un-drifted.
        r_rew = r_rew + 0.5 * (r_next[:, 0] / 2.4) ** 2
    e_rew = float((np.abs(rew - r_rew) / np.maximum(1.0, np.abs(r_rew)))[far].max())
    assert e_rew <= STEP_REWARD_TOL, f"{what} {name}: reward off by {e_rew:.3g} (relative)"
    return {"next": e_next, "reward": e_rew, "flags_compared": int(far.sum()), "pairs": len(far)}


def schedule_groups(sched: dict, n_chunks: int) -> np.ndarray:
    """Host restatement of the kernels' workgroup -> group map (pi_first_chunk in csrc/pi_sweep_kernels.hip): the group
    every workgroup of the launch `sched` (Engine.plan_schedule) takes, in dispatch order (x fastest), -1 for a workgroup
    that leaves at once.  Column 1 is the XCD (dispatch index mod 8)."""
    gx, gy, T, phase, cpw = (sched[k] for k in ("grid_x", "grid_y", "period", "phase", "cpw"))
    assert gx % 8 == 0
    b = np.arange(gx * gy, dtype=np.int64)
    bx, p = b % gx, b // gx
    x, r = bx % 8, bx // 8
    if T == 0:                                  # slab schedule: XCD x walks the x-th contiguous run of groups
        assert gy == 1
        g = x * (gx // 8) + r
        valid = np.ones(len(b), dtype=bool)
    else:                                       # strip schedule: y = period, XCD x takes its eighth of it
        lo8 = x * T + ((p * 3) & 7)
        b0, b1 = lo8 >> 3, (lo8 + T) >> 3
        g = p * T + b0 + r - phase
        valid = (r < b1 - b0) & (g >= 0)
    valid &= g * cpw < n_chunks
    return np.stack([np.where(valid, g, -1), b % 8], axis=1)
