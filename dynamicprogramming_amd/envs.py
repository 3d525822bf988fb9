"""
Env plugins for the MI355X policy-iteration engine.

Each class plugs one control problem into the generic Bellman-backup kernels through
the reference's plugin surface (src/cuda_policy_iteration.py:113-138): a C string that
defines ``step_dynamics`` with the arity for D, and an optional ``_terminal_fn`` mask.
The strings are compiled unchanged by hipRTC (product) and by g++ (test oracle).

The physics, reward shaping, grids, action sets and solver settings restate the
reference runners so that results are comparable state for state; every class cites
the runner it follows.  The arithmetic is kept in the same association order as the
reference's kernels (fp32 is not associative and a greedy argmax is sensitive to the
last bit); ``tests/golden/make_golden.py`` checks that against the reference's own
strings, bit for bit, whenever /root/reference is present.
"""
from __future__ import annotations

from pathlib import Path

import numpy as np

from .solver import (CudaPIConfig, CudaPolicyIteration2D, CudaPolicyIteration4D,
                     CudaPolicyIteration6D)

_PI_WRAP = r'''
#define ENV_PI      3.14159265358979323846f
#define ENV_TWO_PI  (2.0f * ENV_PI)
/* angle -> (-pi, pi]; fmodf keeps the dividend's sign, hence the fix-up */
__device__ float env_wrap_angle(float ang) {
    float w = fmodf(ang + ENV_PI, ENV_TWO_PI);
    if (w < 0.0f) w += ENV_TWO_PI;
    return w - ENV_PI;
}
'''


def _lin(lo, hi, n):
    return np.linspace(lo, hi, n, dtype=np.float32)


# ═════════════════════════════ 2-D ═══════════════════════════════════════════════

class PendulumCuda(CudaPolicyIteration2D):
    """Pendulum-v1 on (theta, omega); theta = 0 upright.  Follows
    runners/pendulum_cuda.py:43-49 (grid/actions), :81-108 (dynamics), :119-125 (config)."""

    DEFAULT_BINS = 200
    ACTIONS = _lin(-2.0, 2.0, 21)
    CONFIG = dict(gamma=0.99, theta=1e-4, max_eval_iter=5_000, max_pi_iter=50, log_interval=200)

    @staticmethod
    def bins_space(bins: int = DEFAULT_BINS) -> dict:
        return {"theta": _lin(-np.pi, np.pi, bins), "theta_dot": _lin(-8.0, 8.0, bins)}

    def _dynamics_cuda_src(self) -> str:
        return _PI_WRAP + r'''
        /* gymnasium PendulumEnv: g = 10, m = l = 1, dt = 0.05, |u| <= 2, |omega| <= 8 */
        __device__ void step_dynamics(float th, float om, float u,
                                      float* th_next, float* om_next,
                                      float* rew, bool* done) {
            u = fmaxf(-2.0f, fminf(2.0f, u));
            const float err = env_wrap_angle(th);          /* cost is paid on the current state */
            *rew = -(err * err + 0.1f * om * om + 0.001f * u * u);
            /* 3g/(2l) = 15, 3/(m l^2) = 3 */
            const float alpha = 15.0f * sinf(th) + 3.0f * u;
            float w = om + alpha * 0.05f;
            w = fmaxf(-8.0f, fminf(8.0f, w));
            *th_next = env_wrap_angle(th + w * 0.05f);
            *om_next = w;
            *done = false;
        }
        '''


class MountainCarCuda(CudaPolicyIteration2D):
    """MountainCar-v0.  Follows runners/mountain_car_cuda.py:53-68 (dynamics), :70-75
    (terminal mask), grid position [-1.2, 0.6] x velocity [-0.07, 0.07], 3 pushes."""

    DEFAULT_BINS = 200
    ACTIONS = np.array([-1.0, 0.0, 1.0], dtype=np.float32)
    CONFIG = dict(gamma=0.99, theta=1e-4, max_eval_iter=5_000, max_pi_iter=50, log_interval=200)

    @staticmethod
    def bins_space(bins: int = DEFAULT_BINS) -> dict:
        return {"position": _lin(-1.2, 0.6, bins), "velocity": _lin(-0.07, 0.07, bins)}

    def _dynamics_cuda_src(self) -> str:
        return r'''
        __device__ void step_dynamics(float p, float v, float push,
                                      float* p_next, float* v_next,
                                      float* rew, bool* done) {
            v += push * 0.001f - 0.0025f * cosf(3.0f * p);
            v = fmaxf(-0.07f, fminf(0.07f, v));
            p += v;
            p = fmaxf(-1.2f, fminf(0.6f, p));
            if (p <= -1.2f) v = 0.0f;                  /* inelastic left wall */
            *p_next = p;
            *v_next = v;
            *done = (p >= 0.5f) && (v >= 0.0f);
            *rew = -1.0f;
        }
        '''

    def _terminal_fn(self, states: np.ndarray):
        return (states[:, 0] >= 0.5) & (states[:, 1] >= 0.0), 0.0

    def _terminal_fn_axes(self, axes):
        return (axes[0] >= 0.5) & (axes[1] >= 0.0), 0.0


class ContinuousMountainCarCuda(CudaPolicyIteration2D):
    """MountainCarContinuous-v0.  Follows runners/continuous_mountain_car_cuda.py:57-76
    (dynamics), :79-85 (terminal mask); 21 forces in [-1, 1]."""

    DEFAULT_BINS = 200
    ACTIONS = _lin(-1.0, 1.0, 21)
    CONFIG = dict(gamma=0.99, theta=1e-4, max_eval_iter=5_000, max_pi_iter=50, log_interval=200)

    @staticmethod
    def bins_space(bins: int = DEFAULT_BINS) -> dict:
        return {"position": _lin(-1.2, 0.6, bins), "velocity": _lin(-0.07, 0.07, bins)}

    def _dynamics_cuda_src(self) -> str:
        return r'''
        __device__ void step_dynamics(float p, float v, float f,
                                      float* p_next, float* v_next,
                                      float* rew, bool* done) {
            f = fmaxf(-1.0f, fminf(1.0f, f));
            v += f * 0.0015f - 0.0025f * cosf(3.0f * p);
            v = fmaxf(-0.07f, fminf(0.07f, v));
            p += v;
            p = fmaxf(-1.2f, fminf(0.6f, p));
            if (p <= -1.2f) v = 0.0f;
            const bool flag = (p >= 0.45f) && (v >= 0.0f);
            *p_next = p;
            *v_next = v;
            *done = flag;
            *rew = -0.1f * f * f + (flag ? 100.0f : 0.0f);   /* bonus on the reaching step */
        }
        '''

    def _terminal_fn(self, states: np.ndarray):
        return (states[:, 0] >= 0.45) & (states[:, 1] >= 0.0), 0.0

    def _terminal_fn_axes(self, axes):
        return (axes[0] >= 0.45) & (axes[1] >= 0.0), 0.0


# ═════════════════════════════ 4-D ═══════════════════════════════════════════════

_CARTPOLE_CORE = r'''
#define CART_GRAV   9.8f
#define CART_MPOLE  0.1f
#define CART_MTOT   1.1f
#define CART_HALFL  0.5f
#define CART_ML     0.05f     /* pole mass x half length */
#define CART_DT     0.02f
/* Barto-Sutton-Anderson cart-pole accelerations for force F */
__device__ void cart_accel(float thd, float sn, float cs, float F, float* xacc, float* thacc) {
    const float t = (F + CART_ML * thd * thd * sn) / CART_MTOT;
    const float a = (CART_GRAV * sn - cs * t)
                    / (CART_HALFL * (4.0f / 3.0f - CART_MPOLE * cs * cs / CART_MTOT));
    *thacc = a;
    *xacc = t - CART_ML * a * cs / CART_MTOT;
}
'''


class CartPoleCuda(CudaPolicyIteration4D):
    """CartPole-v1 balance.  Follows runners/cartpole_cuda.py:45-53 (grid/actions), :80-112
    (dynamics), :114-123 (terminal mask), :131-137 (config)."""

    DEFAULT_BINS = 30
    ACTIONS = np.array([-10.0, 10.0], dtype=np.float32)
    CONFIG = dict(gamma=0.99, theta=1e-4, max_eval_iter=10_000, max_pi_iter=100, log_interval=500)
    _TH_LIMIT = 12.0 * 2.0 * np.pi / 360.0

    @staticmethod
    def bins_space(bins: int = DEFAULT_BINS) -> dict:
        return {"x": _lin(-2.5, 2.5, bins), "x_dot": _lin(-5.0, 5.0, bins),
                "theta": _lin(-0.25, 0.25, bins), "theta_dot": _lin(-5.0, 5.0, bins)}

    def _dynamics_cuda_src(self) -> str:
        return _CARTPOLE_CORE + r'''
        __device__ void step_dynamics(float x, float xd, float th, float thd, float F,
                                      float* x2, float* xd2, float* th2, float* thd2,
                                      float* rew, bool* done) {
            const float cs = cosf(th);
            const float sn = sinf(th);
            float xacc, thacc;
            cart_accel(thd, sn, cs, F, &xacc, &thacc);
            const float px = x + CART_DT * xd;
            const float pth = th + CART_DT * thd;
            *x2 = px;
            *xd2 = xd + CART_DT * xacc;
            *th2 = pth;
            *thd2 = thd + CART_DT * thacc;
            *rew = 1.0f;                       /* +1 per step survived */
            *done = (px < -2.4f) || (px > 2.4f) || (pth < -0.20943951f) || (pth > 0.20943951f);
        }
        '''

    def _terminal_fn(self, states: np.ndarray):
        x, th = states[:, 0], states[:, 2]
        lim = self._TH_LIMIT
        return (x < -2.4) | (x > 2.4) | (th < -lim) | (th > lim), 0.0

    def _terminal_fn_axes(self, axes):
        x, th = axes[0], axes[2]
        lim = self._TH_LIMIT
        return (x < -2.4) | (x > 2.4) | (th < -lim) | (th > lim), 0.0


class CartPoleSwingUpCuda(CudaPolicyIteration4D):
    """Cart-pole swing-up with an energy-shaped reward.  Follows
    runners/cartpole_swingup_cuda.py:45-53 (grid/actions), :88-126 (dynamics), :129-133
    (terminal mask); config gamma .999 / 15 000 / 500."""

    DEFAULT_BINS = 50
    # measured on 50^4 (profiles/r04/dim_order.txt): every order is within the noise of the env's own, which stays
    # (a class WITHOUT a MEMORY_ORDER of its own has its order measured at construction: solver._tune_memory_order)
    MEMORY_ORDER = "user"
    ACTIONS = np.array([-20.0, -10.0, 0.0, 10.0, 20.0], dtype=np.float32)
    CONFIG = dict(gamma=0.999, theta=1e-4, max_eval_iter=15_000, max_pi_iter=500, log_interval=500)

    @staticmethod
    def bins_space(bins: int = DEFAULT_BINS) -> dict:
        return {"x": _lin(-2.5, 2.5, bins), "x_dot": _lin(-5.0, 5.0, bins),
                "theta": _lin(-np.pi, np.pi, bins), "th_dot": _lin(-10.0, 10.0, bins)}

    def _dynamics_cuda_src(self) -> str:
        return _PI_WRAP + _CARTPOLE_CORE + r'''
        #define SWUP_XLIM  2.4f
        #define SWUP_EREF  (CART_MPOLE * CART_GRAV * CART_HALFL)   /* pole energy at upright rest */
        __device__ void step_dynamics(float x, float xd, float th, float thd, float F,
                                      float* x2, float* xd2, float* th2, float* thd2,
                                      float* rew, bool* done) {
            const float cs = cosf(th);
            const float sn = sinf(th);
            float xacc, thacc;
            cart_accel(thd, sn, cs, F, &xacc, &thacc);
            *x2 = x + CART_DT * xd;
            *xd2 = xd + CART_DT * xacc;
            *th2 = env_wrap_angle(th + CART_DT * thd);
            *thd2 = thd + CART_DT * thacc;
            /* pole energy 0.5 m (l w)^2 + m g l cos(th'), compared with the upright level */
            const float tip = CART_HALFL * (*thd2);
            const float E = 0.5f * CART_MPOLE * tip * tip
                          + CART_MPOLE * CART_GRAV * CART_HALFL * cosf(*th2);
            float miss = fabsf(E - SWUP_EREF) / (2.0f * SWUP_EREF);
            miss = fminf(miss, 1.0f);
            const float off = *x2 / SWUP_XLIM;
            *rew = cosf(*th2) - 0.5f * miss - 0.1f * off * off;
            *done = (*x2 < -SWUP_XLIM) || (*x2 > SWUP_XLIM);
        }
        '''

    def _terminal_fn(self, states: np.ndarray):
        x = states[:, 0]
        return (x < -2.4) | (x > 2.4), 0.0

    def _terminal_fn_axes(self, axes):
        x = axes[0]
        return (x < -2.4) | (x > 2.4), 0.0


class DoublePendulumSwingUpCuda(CudaPolicyIteration4D):
    """Base-actuated two-link pendulum swing-up (the BASELINE headline grid).  Follows
    runners/double_pendulum_swingup_cuda.py:53-66 (grid/actions), :94-195 (dynamics and
    reward shaping), :288-294 (config)."""

    DEFAULT_BINS = 15
    # device memory order (theta1, theta2, th1_dot, th2_dot): 80^4 evaluation sweep 0.399 -> 0.372 ms on MI355X
    # (tools/dim_order_sweep.py, profiles/r04/dim_order.txt); big grids only (solver._ORDER_MIN_STATES)
    MEMORY_ORDER = (0, 2, 1, 3)
    SHARDED_MEMORY_ORDER = (0, 2, 1, 3)      # theta1 stays slowest: the same order serves the sharded fused exchange
    ACTIONS = np.array([-3.0, -1.5, -0.5, -0.15, -0.05, 0.0, 0.05, 0.15, 0.5, 1.5, 3.0],
                       dtype=np.float32)
    CONFIG = dict(gamma=0.999, theta=1e-4, max_eval_iter=15_000, max_pi_iter=300, log_interval=500)

    @staticmethod
    def bins_space(bins: int = DEFAULT_BINS) -> dict:
        return {"theta1": _lin(-np.pi, np.pi, bins), "th1_dot": _lin(-15.0, 15.0, bins),
                "theta2": _lin(-np.pi, np.pi, bins), "th2_dot": _lin(-15.0, 15.0, bins)}

    def _dynamics_cuda_src(self) -> str:
        return _PI_WRAP + r'''
        #define LK_G    9.8f
        #define LK_MA   0.1f      /* link-1 tip mass */
        #define LK_MB   0.1f      /* link-2 tip mass */
        #define LK_LA   0.5f
        #define LK_LB   0.5f
        #define LK_DT   0.02f
        /* mechanical energy at upright rest: (ma + mb) g la + mb g lb */
        #define LK_EREF ((LK_MA + LK_MB) * LK_G * LK_LA + LK_MB * LK_G * LK_LB)

        __device__ void step_dynamics(float q1, float w1, float q2, float w2, float tau,
                                      float* q1n, float* w1n, float* q2n, float* w2n,
                                      float* rew, bool* done) {
            const float msum = LK_MA + LK_MB;
            const float dq = q1 - q2;
            const float cd = cosf(dq);
            const float sd = sinf(dq);
            /* M [w1', w2']^T = b with M = [[msum la^2, mb la lb cd], [., mb lb^2]] */
            const float m11 = msum * LK_LA * LK_LA;
            const float m12 = LK_MB * LK_LA * LK_LB * cd;
            const float m22 = LK_MB * LK_LB * LK_LB;
            /* only the base joint is driven */
            const float b1 = tau + msum * LK_G * LK_LA * sinf(q1) - LK_MB * LK_LA * LK_LB * sd * w2 * w2;
            const float b2 = LK_MB * LK_G * LK_LB * sinf(q2) + LK_MB * LK_LA * LK_LB * sd * w1 * w1;
            const float det = m11 * m22 - m12 * m12;
            const float a1 = (m22 * b1 - m12 * b2) / det;
            const float a2 = (m11 * b2 - m12 * b1) / det;

            *q1n = env_wrap_angle(q1 + LK_DT * w1);
            *w1n = w1 + LK_DT * a1;
            *q2n = env_wrap_angle(q2 + LK_DT * w2);
            *w2n = w2 + LK_DT * a2;

            /* ---- shaped reward, all on the successor state ---- */
            const float c1 = cosf(*q1n);
            const float c2 = cosf(*q2n);
            const float cdn = cosf(*q1n - *q2n);
            const float u1 = *w1n;
            const float u2 = *w2n;
            const float kin = 0.5f * msum * LK_LA * LK_LA * u1 * u1
                            + 0.5f * LK_MB * LK_LB * LK_LB * u2 * u2
                            + LK_MB * LK_LA * LK_LB * u1 * u2 * cdn;
            const float pot = msum * LK_G * LK_LA * c1 + LK_MB * LK_G * LK_LB * c2;
            /* too little energy costs 1.5x what too much does */
            const float gap = (kin + pot) - LK_EREF;
            const float e_cost = (gap < 0.0f) ? 1.5f * (-gap) / (2.0f * LK_EREF)
                                              : gap / (2.0f * LK_EREF);
            const float up1 = fmaxf(0.0f, c1);
            const float up2 = fmaxf(0.0f, c2);
            const float gate = up1 * up2;                 /* both links above horizontal */
            const float fold = c1 - c2;                   /* "I-shape" (folded) attractor */
            const float fold_cost = 0.5f * fold * fold;
            const float spin_cost = 0.1f * gate * (u1 * u1 + u2 * u2);
            const float gate_sq = gate * gate;
            const float near_top = 4.0f * gate_sq;
            /* smooth stillness bowl: 5 at gate = 1, zero speed; gone once |w|^2 >= 2.5 */
            const float speed2 = u1 * u1 + u2 * u2;
            const float calm = fmaxf(0.0f, 1.0f - speed2 / 2.5f);
            const float calm_sq = calm * calm;
            const float gate_4 = gate_sq * gate_sq;
            const float settle = 5.0f * gate_4 * calm_sq;

            *rew = 0.5f + 0.5f * (c1 + c2) + near_top + settle
                 - 1.0f * e_cost - fold_cost - spin_cost;
            *done = false;      /* angles wrap; speeds are clamped by the grid border */
        }
        '''


class OverheadCraneCuda(CudaPolicyIteration4D):
    """Overhead-crane anti-sway positioning.  Follows runners/overhead_crane_cuda.py:56-64
    (grid/actions), :106-155 (dynamics), :175-206 (goal/fail masks and goal-value seeding)."""

    DEFAULT_BINS = 30
    ACTIONS = np.array([-30.0, -20.0, -10.0, 0.0, 10.0, 20.0, 30.0], dtype=np.float32)
    CONFIG = dict(gamma=0.999, theta=1e-4, max_eval_iter=10_000, max_pi_iter=100, log_interval=500)
    _RAIL = 3.0
    _TH_EDGE = (np.pi / 2.0) * 1.1

    def __init__(self, bins_space, action_space, config=None, target_x: float = 0.0, **kw):
        self.target_x = float(target_x)
        super().__init__(bins_space, action_space, config, **kw)

    @classmethod
    def bins_space(cls, bins: int = DEFAULT_BINS) -> dict:
        return {"x": _lin(-cls._RAIL, cls._RAIL, bins), "x_dot": _lin(-4.0, 4.0, bins),
                "theta": _lin(-cls._TH_EDGE, cls._TH_EDGE, bins), "theta_dot": _lin(-4.0, 4.0, bins)}

    def _dynamics_cuda_src(self) -> str:
        body = r'''
        #define CR_G      9.81f
        #define CR_GOAL_X @TARGET@f    /* trolley set-point, fixed when the kernel is built */
        #define CR_MT     1.0f         /* trolley */
        #define CR_ML     5.0f         /* load */
        #define CR_ROPE   1.5f
        #define CR_DT     0.02f
        #define CR_RAIL   3.0f
        __device__ void step_dynamics(float x, float xd, float th, float thd, float F,
                                      float* x2, float* xd2, float* th2, float* thd2,
                                      float* rew, bool* done) {
            const float cs = cosf(th);
            const float sn = sinf(th);
            /* H = [[a, b], [b, d]] */
            const float a = CR_MT + CR_ML;
            const float b = CR_ML * CR_ROPE * cs;
            const float d = CR_ML * CR_ROPE * CR_ROPE;
            const float r1 = F + CR_ML * CR_ROPE * thd * thd * sn;
            const float r2 = -CR_ML * CR_G * CR_ROPE * sn;
            const float det = a * d - b * b;
            const float inv = 1.0f / det;
            const float xacc = (d * r1 - b * r2) * inv;
            const float thacc = (-b * r1 + a * r2) * inv;

            *x2 = x + CR_DT * xd;
            *xd2 = xd + CR_DT * xacc;
            *th2 = th + CR_DT * thd;
            *thd2 = thd + CR_DT * thacc;

            /* quadratic costs, each normalised by its scale (rail, 4 m/s, 30 deg, 4 rad/s) */
            const float ex = (*x2 - CR_GOAL_X) / (2.0f * CR_RAIL);
            const float ev = *xd2 / 4.0f;
            const float eq = *th2 / 0.52360f;
            const float ew = *thd2 / 4.0f;
            *rew = 1.0f - 0.15f * ex * ex - 0.15f * ev * ev - 0.45f * eq * eq - 0.25f * ew * ew;

            const bool crashed = (*x2 <= -CR_RAIL) || (*x2 >= CR_RAIL);
            const bool parked = (fabsf(*x2 - CR_GOAL_X) <= 0.20f) && (fabsf(*th2) <= 0.10f)
                             && (fabsf(*xd2) <= 0.20f);
            *done = crashed || parked;
        }
        '''
        return body.replace("@TARGET@", f"{self.target_x:.6f}")

    def _terminal_fn(self, states: np.ndarray):
        x, xd, th = states[:, 0], states[:, 1], states[:, 2]
        crashed = (x <= -self._RAIL) | (x >= self._RAIL)
        parked = (np.abs(x - self.target_x) <= 0.20) & (np.abs(th) <= 0.10) & (np.abs(xd) <= 0.15)
        self._goal_mask = parked
        return crashed | parked, 0.0

    def _allocate_tensors_and_compile(self) -> None:
        # Goal cells start at the value of collecting reward 1 forever (reference :193-206).
        super()._allocate_tensors_and_compile()
        goal = getattr(self, "_goal_mask", None)
        if goal is not None and np.any(goal):
            self._seed_values(goal, float(1.0 / (1.0 - self.config.gamma)))

    def save(self, filepath) -> None:
        super().save(filepath)
        path = Path(filepath).with_suffix(".npz")
        data = dict(np.load(path))
        data["target_x"] = np.float32(self.target_x)
        np.savez(path, **data)

    @classmethod
    def load(cls, filepath):
        inst = super().load(filepath)
        data = np.load(Path(filepath).with_suffix(".npz"))
        inst.target_x = float(data["target_x"]) if "target_x" in data else 0.0
        return inst


# ═════════════════════════════ 6-D ═══════════════════════════════════════════════

_DOUBLE_CART_CORE = r'''
#define DC_G     9.8f
#define DC_MC    1.0f      /* cart */
#define DC_MA    0.1f      /* pole-1 tip mass */
#define DC_MB    0.1f      /* pole-2 tip mass */
#define DC_LA    0.5f
#define DC_LB    0.5f
#define DC_DT    0.02f
/* accelerations of (x, th1, th2) from the 3x3 symmetric mass matrix, by cofactors */
__device__ void dc_accel(float th1, float w1, float th2, float w2, float F,
                         float* ax, float* a1, float* a2) {
    const float msum = DC_MA + DC_MB;
    const float c1 = cosf(th1), s1 = sinf(th1);
    const float c2 = cosf(th2), s2 = sinf(th2);
    const float dq = th1 - th2;
    const float cd = cosf(dq), sd = sinf(dq);
    /* H = [[h11,h12,h13],[.,h22,h23],[.,.,h33]] */
    const float h11 = DC_MC + msum;
    const float h12 = msum * DC_LA * c1;
    const float h13 = DC_MB * DC_LB * c2;
    const float h22 = msum * DC_LA * DC_LA;
    const float h23 = DC_MB * DC_LA * DC_LB * cd;
    const float h33 = DC_MB * DC_LB * DC_LB;
    const float f1 = F + msum * DC_LA * w1 * w1 * s1 + DC_MB * DC_LB * w2 * w2 * s2;
    const float f2 = msum * DC_G * DC_LA * s1 - DC_MB * DC_LA * DC_LB * w2 * w2 * sd;
    const float f3 = DC_MB * DC_G * DC_LB * s2 + DC_MB * DC_LA * DC_LB * w1 * w1 * sd;
    const float k11 = h22 * h33 - h23 * h23;
    const float k12 = h23 * h13 - h12 * h33;
    const float k13 = h12 * h23 - h22 * h13;
    const float k22 = h11 * h33 - h13 * h13;
    const float k23 = h12 * h13 - h11 * h23;
    const float k33 = h11 * h22 - h12 * h12;
    const float det = h11 * k11 + h12 * k12 + h13 * k13;
    const float inv = 1.0f / det;
    *ax = (k11 * f1 + k12 * f2 + k13 * f3) * inv;
    *a1 = (k12 * f1 + k22 * f2 + k23 * f3) * inv;
    *a2 = (k13 * f1 + k23 * f2 + k33 * f3) * inv;
}
'''


class DoubleCartPoleCuda(CudaPolicyIteration6D):
    """Double inverted pendulum on a cart, balance task.  Follows
    runners/double_cartpole_cuda.py:57-72 (grid/actions), :100-168 (dynamics), :171-181
    (terminal mask); config gamma .999 / 10 000 / 200."""

    DEFAULT_BINS = 15
    # device memory order (th2_dot, theta2, theta1, th1_dot, x, x_dot) — the cart's speed along the lanes (a wave's 64
    # successors then share their cell along every other dimension), its position next to it: 25^6 evaluation sweep
    # 3.68 -> 2.87 ms, improvement 7.19 -> 6.88 ms (tools/dim_order_sweep.py, profiles/r04/dim_order.txt; with x kept
    # slowest, (0, 2, 3, 5, 4, 1): 3.07 / 7.11).  Single-rank solvers only: sharded ones keep the env's order.
    MEMORY_ORDER = (5, 4, 2, 3, 0, 1)
    SHARDED_MEMORY_ORDER = (0, 2, 3, 5, 4, 1)   # x slowest (shards), x_dot along the lanes: solver.SHARDED_MEMORY_ORDER
    ACTIONS = np.array([-10.0, 0.0, 10.0], dtype=np.float32)
    CONFIG = dict(gamma=0.999, theta=1e-4, max_eval_iter=10_000, max_pi_iter=200, log_interval=500)
    _TH_FAIL = 20.0 * np.pi / 180.0
    _TH_EDGE = _TH_FAIL * 1.15

    @classmethod
    def bins_space(cls, bins: int = DEFAULT_BINS) -> dict:
        e = cls._TH_EDGE
        return {"x": _lin(-2.5, 2.5, bins), "x_dot": _lin(-5.0, 5.0, bins),
                "theta1": _lin(-e, e, bins), "th1_dot": _lin(-5.0, 5.0, bins),
                "theta2": _lin(-e, e, bins), "th2_dot": _lin(-5.0, 5.0, bins)}

    def _dynamics_cuda_src(self) -> str:
        return _DOUBLE_CART_CORE + r'''
        #define BAL_XLIM   2.4f
        #define BAL_THLIM  0.34906585f    /* 20 deg */
        #define BAL_XCOST  0.0f           /* weight of the (disabled) position penalty */
        __device__ void step_dynamics(float x, float xd, float th1, float w1, float th2, float w2,
                                      float F,
                                      float* x_n, float* xd_n, float* th1_n, float* w1_n,
                                      float* th2_n, float* w2_n, float* rew, bool* done) {
            float ax, a1, a2;
            dc_accel(th1, w1, th2, w2, F, &ax, &a1, &a2);
            *x_n = x + DC_DT * xd;
            *xd_n = xd + DC_DT * ax;
            *th1_n = th1 + DC_DT * w1;
            *w1_n = w1 + DC_DT * a1;
            *th2_n = th2 + DC_DT * w2;
            *w2_n = w2 + DC_DT * a2;
            const float off = *x_n / BAL_XLIM;
            *rew = 1.0f - BAL_XCOST * off * off;
            *done = (*x_n < -BAL_XLIM) || (*x_n > BAL_XLIM)
                 || (*th1_n < -BAL_THLIM) || (*th1_n > BAL_THLIM)
                 || (*th2_n < -BAL_THLIM) || (*th2_n > BAL_THLIM);
        }
        '''

    def _terminal_fn(self, states: np.ndarray):
        x, t1, t2 = states[:, 0], states[:, 2], states[:, 4]
        lim = self._TH_FAIL
        return ((x < -2.4) | (x > 2.4) | (t1 < -lim) | (t1 > lim) | (t2 < -lim) | (t2 > lim)), 0.0

    def _terminal_fn_axes(self, axes):
        x, t1, t2 = axes[0], axes[2], axes[4]
        lim = self._TH_FAIL
        return ((x < -2.4) | (x > 2.4) | (t1 < -lim) | (t1 > lim) | (t2 < -lim) | (t2 > lim)), 0.0


class DoubleCartPoleSwingUpCuda(CudaPolicyIteration6D):
    """Double inverted pendulum swing-up on a cart.  Follows
    runners/double_cartpole_swingup_cuda.py:62-74 (grid/actions), :114-238 (dynamics and
    reward shaping), :241-245 (terminal mask); config gamma .999 / 20 000 / 300."""

    DEFAULT_BINS = 20
    # device memory order (theta2, th2_dot, theta1, th1_dot, x, x_dot) — x_dot along the lanes, x next to it: 25^6
    # evaluation sweep 7.84 -> 5.16 ms, improvement 35.2 -> 24.3 ms (tools/dim_order_sweep.py, profiles/r04/dim_order.txt;
    # with x kept slowest, (0, 4, 5, 2, 3, 1): 5.38 / 28.9).  Single-rank solvers only: sharded ones keep the env's order.
    MEMORY_ORDER = (4, 5, 2, 3, 0, 1)
    SHARDED_MEMORY_ORDER = (0, 4, 5, 2, 3, 1)   # x slowest (shards), x_dot along the lanes
    ACTIONS = np.array([-60.0, -30.0, -10.0, -3.0, 0.0, 3.0, 10.0, 30.0, 60.0], dtype=np.float32)
    CONFIG = dict(gamma=0.999, theta=1e-4, max_eval_iter=20_000, max_pi_iter=300, log_interval=500)

    @staticmethod
    def bins_space(bins: int = DEFAULT_BINS) -> dict:
        return {"x": _lin(-2.5, 2.5, bins), "x_dot": _lin(-8.0, 8.0, bins),
                "theta1": _lin(-np.pi, np.pi, bins), "th1_dot": _lin(-15.0, 15.0, bins),
                "theta2": _lin(-np.pi, np.pi, bins), "th2_dot": _lin(-15.0, 15.0, bins)}

    def _dynamics_cuda_src(self) -> str:
        return _PI_WRAP + _DOUBLE_CART_CORE + r'''
        #define SW_XLIM  2.4f
        #define SW_EREF  ((DC_MA + DC_MB) * DC_G * DC_LA + DC_MB * DC_G * DC_LB)
        __device__ void step_dynamics(float x, float xd, float th1, float w1, float th2, float w2,
                                      float F,
                                      float* x_n, float* xd_n, float* th1_n, float* w1_n,
                                      float* th2_n, float* w2_n, float* rew, bool* done) {
            const float msum = DC_MA + DC_MB;
            float ax, a1, a2;
            dc_accel(th1, w1, th2, w2, F, &ax, &a1, &a2);
            *x_n = x + DC_DT * xd;
            *xd_n = xd + DC_DT * ax;
            *th1_n = env_wrap_angle(th1 + DC_DT * w1);
            *w1_n = w1 + DC_DT * a1;
            *th2_n = env_wrap_angle(th2 + DC_DT * w2);
            *w2_n = w2 + DC_DT * a2;

            /* ---- shaped reward on the successor state ---- */
            const float c1 = cosf(*th1_n);
            const float c2 = cosf(*th2_n);
            const float cdn = cosf(*th1_n - *th2_n);
            const float u1 = *w1_n;
            const float u2 = *w2_n;
            const float kin = 0.5f * msum * DC_LA * DC_LA * u1 * u1
                            + 0.5f * DC_MB * DC_LB * DC_LB * u2 * u2
                            + DC_MB * DC_LA * DC_LB * u1 * u2 * cdn;
            const float pot = msum * DC_G * DC_LA * c1 + DC_MB * DC_G * DC_LB * c2;
            const float gap = (kin + pot) - SW_EREF;
            const float e_cost = (gap < 0.0f) ? 2.5f * (-gap) / (2.0f * SW_EREF)
                                              : 1.5f * gap / (2.0f * SW_EREF);
            const float up1 = fmaxf(0.0f, c1);
            const float up2 = fmaxf(0.0f, c2);
            const float gate = up1 * up2;
            const float spin_cost = 0.1f * gate * (u1 * u1 + u2 * u2);
            const float plateau = (c1 > 0.7f && c2 > 0.7f) ? 6.0f : 0.0f;
            const float ev = *xd_n / 8.0f;
            const float ex = *x_n / SW_XLIM;
            *rew = 0.5f + 0.5f * (c1 + c2) + 0.5f * up1 + 1.0f * up2 + 6.0f * gate + plateau
                 - 1.0f * e_cost - 0.5f * ex * ex - 0.2f * ev * ev - spin_cost;
            if ((*x_n < -SW_XLIM) || (*x_n > SW_XLIM)) {
                *rew -= 100.0f;               /* leaving the rail */
            }
            *done = (*x_n < -SW_XLIM) || (*x_n > SW_XLIM);
        }
        '''

    def _terminal_fn(self, states: np.ndarray):
        x = states[:, 0]
        return (x < -2.4) | (x > 2.4), 0.0

    def _terminal_fn_axes(self, axes):
        x = axes[0]
        return (x < -2.4) | (x > 2.4), 0.0


ENVS = {
    "pendulum": PendulumCuda,
    "mountain_car": MountainCarCuda,
    "continuous_mountain_car": ContinuousMountainCarCuda,
    "cartpole": CartPoleCuda,
    "cartpole_swingup": CartPoleSwingUpCuda,
    "double_pendulum_swingup": DoublePendulumSwingUpCuda,
    "overhead_crane": OverheadCraneCuda,
    "double_cartpole": DoubleCartPoleCuda,
    "double_cartpole_swingup": DoubleCartPoleSwingUpCuda,
}


def dynamics_source(name: str, **kw) -> str:
    """The env's plugin string without constructing a solver (no GPU needed)."""
    cls = ENVS[name]
    inst = object.__new__(cls)
    for k, v in kw.items():
        setattr(inst, k, v)
    if cls is OverheadCraneCuda and "target_x" not in kw:
        inst.target_x = 0.0
    return cls._dynamics_cuda_src(inst)


def make(name: str, bins: int | None = None, config: CudaPIConfig | None = None, **kw):
    """Construct env `name` on its reference grid with `bins` points per dimension."""
    cls = ENVS[name]
    cfg = config or CudaPIConfig(**cls.CONFIG)
    return cls(cls.bins_space(bins or cls.DEFAULT_BINS), cls.ACTIONS, cfg, **kw)
