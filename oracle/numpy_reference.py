"""
oracle/numpy_reference.py — vectorised numpy restatement of the 2-D sweep path for BASELINE
config 1 ("Pendulum-v1 2D bins=50, 11 actions — numpy CPU reference sweep").  TEST INFRASTRUCTURE
ONLY, like the rest of oracle/: imported by tests/ and by bench.py's `cpu_baseline` leg.

Restates, over whole arrays instead of one thread per state:
  get_barycentric_2d      /root/reference/src/cuda_policy_iteration.py:183-210
  policy_eval_kernel      :212-242        policy_improve_kernel :244-283
  policy_evaluation / policy_improvement / run loop   :300-370 (residual looked at on sweeps 0, 25, ...)
and the pendulum plugin of runners/pendulum_cuda.py:81-108 (as restated in
dynamicprogramming_amd/envs.py).  float32 arrays throughout; numpy has no fused multiply-add and
its float32 sin is its own, so this path agrees with the C++ oracle to rounding (tested:
|dV| <= 2e-4, policy equal on >= 99.5 % of states), not bit for bit.
"""
from __future__ import annotations

import numpy as np

F = np.float32
PI = F(3.14159265358979323846)
TWO_PI = F(2.0) * PI


def wrap_angle(a):
    w = np.fmod(a + PI, TWO_PI)
    w = np.where(w < F(0.0), w + TWO_PI, w)
    return (w - PI).astype(F)


def pendulum_step(th, om, u):
    u = np.maximum(F(-2.0), np.minimum(F(2.0), u)).astype(F)
    err = wrap_angle(th)
    rew = -(err * err + F(0.1) * om * om + F(0.001) * u * u)
    alpha = F(15.0) * np.sin(th, dtype=F) + F(3.0) * u
    w = om + alpha * F(0.05)
    w = np.maximum(F(-8.0), np.minimum(F(8.0), w)).astype(F)
    return wrap_angle(th + w * F(0.05)), w, rew.astype(F), np.zeros(th.shape, dtype=bool)


def interp_2d(s0, s1, lo, hi, shape, strides):
    """Indices (n, 4) and weights (n, 4) in the reference's corner order (0,0),(0,1),(1,0),(1,1)."""
    out_i, out_f = [], []
    for s, d in ((s0, 0), (s1, 1)):
        top = F(shape[d] - 1)
        n = (s - lo[d]) / (hi[d] - lo[d]) * top
        n = np.maximum(F(0.0), np.minimum(n, top)).astype(F)
        i = np.minimum(n.astype(np.int32), np.int32(shape[d] - 2))
        out_i.append(i)
        out_f.append((n - i.astype(F)).astype(F))
    (i0, i1), (f0, f1) = out_i, out_f
    base = i0 * np.int32(strides[0]) + i1 * np.int32(strides[1])
    idx = np.stack([base, base + strides[1], base + strides[0], base + strides[0] + strides[1]], axis=1)
    w = np.stack([(F(1) - f0) * (F(1) - f1), (F(1) - f0) * f1, f0 * (F(1) - f1), f0 * f1], axis=1)
    return idx.astype(np.int32), w.astype(F)


class PendulumNumpy:
    def __init__(self, bins, actions):
        self.bins = [np.asarray(b, F) for b in bins]
        self.actions = np.asarray(actions, F)
        self.shape = np.array([len(b) for b in self.bins], np.int32)
        self.strides = np.array([self.shape[1], 1], np.int32)
        self.lo = np.array([b.min() for b in self.bins], F)
        self.hi = np.array([b.max() for b in self.bins], F)
        g0, g1 = np.meshgrid(*self.bins, indexing="ij")
        self.s0, self.s1 = g0.ravel().astype(F), g1.ravel().astype(F)
        self.n = len(self.s0)

    def q_values(self, V, u):
        t, w, rew, done = pendulum_step(self.s0, self.s1, u)
        idx, wt = interp_2d(t, w, self.lo, self.hi, self.shape, self.strides)
        e = np.zeros(self.n, dtype=F)
        for c in range(4):                      # ascending corner order, like the fmaf chain
            e = (wt[:, c] * V[idx[:, c]] + e).astype(F)
        e = np.where(done, F(0.0), e)
        return (rew + F(self.gamma) * e).astype(F)

    def eval_sweep(self, V, policy):
        newV = self.q_values(V, self.actions[policy])
        return newV, float(np.max(np.abs(newV - V)))

    def improve_sweep(self, V, policy):
        best_q = np.full(self.n, F(-1.0e30))
        best = np.zeros(self.n, np.int32)
        for a, u in enumerate(self.actions):
            q = self.q_values(V, np.full(self.n, u, F))
            better = q > best_q
            best_q = np.where(better, q, best_q)
            best = np.where(better, np.int32(a), best)
        return best, int((best != policy).sum())

    def run(self, gamma, theta, max_eval_iter, max_pi_iter):
        self.gamma = F(gamma)
        V = np.zeros(self.n, F)
        policy = np.zeros(self.n, np.int32)
        sweeps_per_iter, stable = [], False
        for _ in range(max_pi_iter):
            k = 0
            for i in range(max_eval_iter):
                V, delta = self.eval_sweep(V, policy)
                k += 1
                if (i % 25 == 0 or i == max_eval_iter - 1) and delta < theta:
                    break
            sweeps_per_iter.append(k)
            policy, changed = self.improve_sweep(V, policy)
            if changed == 0:
                stable = True
                break
        return {"value_function": V, "policy": policy, "sweeps_per_iter": sweeps_per_iter,
                "eval_sweeps": int(sum(sweeps_per_iter)), "outer_iterations": len(sweeps_per_iter),
                "stable": stable}
