"""
GPU parity: the HIP path (through the C ABI of libpi_mi355.so) against the CPU oracle.

Bar (north_star): greedy-policy indices bit-exact; V within a stated fp32 tolerance.  Because
the kernels and the oracle share include/pi_math.h and neither contracts fp32 operations, the
tests below ask for more: V, residuals and change counts are compared BIT FOR BIT against the
oracle built in the product's arithmetic mode.  Against the reference-text goldens (glibc
libm) the tolerance is written out: |dV| <= 1e-5 * max(1, |V|) (measured: 2.3e-6) and policy equal
wherever the top-2 action-value gap exceeds 1e-5 * max(1, |V|) (measured: no mismatch at all).
"""
from __future__ import annotations

import os

import numpy as np
import pytest

import oracle
from dynamicprogramming_amd import _native, envs
from tests import helpers as H

pytestmark = pytest.mark.gpu


def _torch():
    import torch
    return torch


def _engine(name, shape, dev, actions=None):
    cls = envs.ENVS[name]
    bins = H.env_bins(name, shape)
    acts = np.asarray(cls.ACTIONS if actions is None else actions, np.float32)
    eng = _native.Engine(cls._D, [len(b) for b in bins], [b.min() for b in bins],
                         [b.max() for b in bins], bins, acts, device=dev.index or 0)
    eng.compile(envs.dynamics_source(name))
    return eng, bins, acts


def _dev(a, dev):
    torch = _torch()
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


@pytest.mark.parametrize("name", H.ENV_NAMES)
def test_dynamics_bit_exact(name, cuda_device):
    """step_dynamics on the GPU == the same string on the CPU (pi_math mode), bit for bit."""
    torch = _torch()
    g = H.golden(name)
    D = int(g["D"])
    eng, bins, _ = _engine(name, H.golden(name)["g0_shape"], cuda_device)
    rng = np.random.default_rng(7)
    st = np.concatenate([g["step_states"], H.sample_states(rng, bins, 4096)])
    act = np.concatenate([g["step_actions"],
                          rng.choice(envs.ENVS[name].ACTIONS, size=4096).astype(np.float32)])
    m = len(st)
    d_st, d_act = _dev(st, cuda_device), _dev(act, cuda_device)
    d_next = torch.empty((m, D), dtype=torch.float32, device=cuda_device)
    d_rew = torch.empty(m, dtype=torch.float32, device=cuda_device)
    d_done = torch.empty(m, dtype=torch.uint8, device=cuda_device)
    eng.probe_step(d_st.data_ptr(), d_act.data_ptr(), d_next.data_ptr(), d_rew.data_ptr(),
                   d_done.data_ptr(), m)
    torch.cuda.synchronize()
    o_next, o_rew, o_done = H.oracle_for(name).step(st, act)
    H.assert_bits_equal(d_next.cpu().numpy(), o_next, f"{name} next state")
    H.assert_bits_equal(d_rew.cpu().numpy(), o_rew, f"{name} reward")
    assert np.array_equal(d_done.cpu().numpy().astype(bool), o_done)
    # against the reference-text golden (glibc libm): ulp-level agreement only
    k = len(g["step_states"])
    np.testing.assert_allclose(d_next.cpu().numpy()[:k], g["step_next"], rtol=5e-6, atol=5e-6)     # measured 1.1e-6
    np.testing.assert_allclose(d_rew.cpu().numpy()[:k], g["step_reward"], rtol=5e-6, atol=5e-6)    # measured 6.2e-7
    eng.close()


@pytest.mark.parametrize("name", list(H.STEP_PYTHON_ENVS))
def test_dynamics_match_reference_step_python(name, cuda_device):
    """step_dynamics on the GPU against vectors the reference's OWN `_step_python` produced (reference-executed
    code, tests/golden/make_step_python_golden.py): next state within 1e-5 relative (angles on the circle), reward
    within 5e-5 relative, `terminated` exact wherever the successor is 1e-4 or more from every threshold."""
    torch = _torch()
    g = np.load(H.GOLDEN / "step_python.npz")
    st, act = g[f"{name}_states"], g[f"{name}_actions"]
    D, m = st.shape[1], len(st)
    eng, _, _ = _engine(name, H.golden(name)["g0_shape"], cuda_device)
    d_st, d_act = _dev(st, cuda_device), _dev(act, cuda_device)
    d_next = torch.empty((m, D), dtype=torch.float32, device=cuda_device)
    d_rew = torch.empty(m, dtype=torch.float32, device=cuda_device)
    d_done = torch.empty(m, dtype=torch.uint8, device=cuda_device)
    eng.probe_step(d_st.data_ptr(), d_act.data_ptr(), d_next.data_ptr(), d_rew.data_ptr(), d_done.data_ptr(), m)
    torch.cuda.synchronize()
    got = H.check_against_step_python(name, d_next.cpu().numpy(), d_rew.cpu().numpy(),
                                      d_done.cpu().numpy().astype(bool), "GPU")
    assert got["flags_compared"] >= m - 10
    eng.close()


@pytest.mark.parametrize("name", ["pendulum", "double_pendulum_swingup", "double_cartpole"])
@pytest.mark.parametrize("gi", [0, 1])
def test_interpolation_matches_reference_golden(name, gi, cuda_device):
    """get_barycentric_{2,4,6}d: indices AND weights equal the reference-text golden exactly
    (no transcendental is involved, so libm does not matter)."""
    torch = _torch()
    g = H.golden(name)
    D = int(g["D"])
    eng, _, _ = _engine(name, g[f"g{gi}_shape"], cuda_device)
    pts = g[f"g{gi}_pts"]
    m = len(pts)
    d_idx = torch.empty((m, 1 << D), dtype=torch.int32, device=cuda_device)
    d_w = torch.empty((m, 1 << D), dtype=torch.float32, device=cuda_device)
    eng.probe_interp(_dev(pts, cuda_device).data_ptr(), d_idx.data_ptr(), d_w.data_ptr(), m)
    torch.cuda.synchronize()
    assert np.array_equal(d_idx.cpu().numpy(), g[f"g{gi}_idx"])
    H.assert_bits_equal(d_w.cpu().numpy(), g[f"g{gi}_w"], f"{name} weights")
    eng.close()


def _sweep_case(name, shape, dev, seed=0, actions=None):
    eng, bins, acts = _engine(name, shape, dev, actions)
    lo, hi, gshape, strides = oracle.grid_metadata(bins)
    states = oracle.states_from_bins(bins)
    term, tval = H.terminal_mask(name, states)
    rng = np.random.default_rng(seed)
    V = (rng.standard_normal(len(states)) * 3.0).astype(np.float32)
    V[term] = np.float32(tval)
    pol = rng.integers(0, len(acts), size=len(states)).astype(np.int32)
    pol[term] = 0
    return eng, acts, (lo, hi, gshape, strides), states, term, V, pol


@pytest.mark.parametrize("name", H.ENV_NAMES)
@pytest.mark.parametrize("gi", [0, 1])
def test_sweeps_bit_exact_and_golden(name, gi, cuda_device):
    torch = _torch()
    g = H.golden(name)
    shape = g[f"g{gi}_shape"]
    eng, bins, acts = _engine(name, shape, cuda_device)
    lo, hi, gshape, strides = oracle.grid_metadata(bins)
    states = oracle.states_from_bins(bins)
    V, pol, term = g[f"g{gi}_V"], g[f"g{gi}_policy"], g[f"g{gi}_term"]
    gamma = float(g["gamma"])
    n = len(V)
    d_V, d_pol = _dev(V, cuda_device), _dev(pol, cuda_device)
    d_term = _dev(term.astype(np.uint8), cuda_device)
    d_Vn = torch.full((n,), float("nan"), dtype=torch.float32, device=cuda_device)
    d_delta = torch.full((1,), 123.0, dtype=torch.float32, device=cuda_device)
    d_changed = torch.full((1,), 77, dtype=torch.int32, device=cuda_device)
    eng.eval_sweep(d_V.data_ptr(), d_Vn.data_ptr(), d_pol.data_ptr(), d_term.data_ptr(), 0, n,
                   gamma, d_delta.data_ptr())
    eng.improve_sweep(d_V.data_ptr(), d_pol.data_ptr(), d_term.data_ptr(), 0, n, gamma,
                      d_changed.data_ptr())
    torch.cuda.synchronize()
    Vn, pol_n = d_Vn.cpu().numpy(), d_pol.cpu().numpy()
    chk = H.oracle_for(name)
    o_Vn, o_delta = chk.eval_sweep(states, acts, pol, V, term, lo, hi, gshape, strides, gamma)
    o_pol, o_changed = chk.improve_sweep(states, acts, pol, V, term, lo, hi, gshape, strides, gamma)
    # bit-exact against the oracle in the product's arithmetic
    H.assert_bits_equal(Vn, o_Vn, f"{name} V'")
    assert np.float32(d_delta.item()) == np.float32(o_delta)
    assert np.array_equal(pol_n, o_pol)
    assert int(d_changed.item()) == o_changed
    # reference-text golden (glibc libm): stated tolerance
    gV = g[f"g{gi}_V_next"]
    assert np.all(np.abs(Vn - gV) <= 1e-5 * np.maximum(1.0, np.abs(gV)))          # measured: <= 2.3e-6
    firm = g[f"g{gi}_q_gap"] > 1e-5 * max(1.0, float(np.abs(gV).max()))
    assert np.array_equal(pol_n[firm], g[f"g{gi}_policy_next"][firm])              # measured: no mismatch anywhere
    assert np.mean(pol_n == g[f"g{gi}_policy_next"]) >= 0.999
    eng.close()


@pytest.mark.parametrize("name,shape", [("pendulum", (200, 200)), ("cartpole_swingup", (17, 13, 19, 11)),
                                         ("double_pendulum_swingup", (20, 20, 20, 20)),
                                         ("double_cartpole", (7, 6, 8, 5, 7, 6))])
def test_ragged_ranges_and_pingpong(name, shape, cuda_device):
    """Sub-range sweeps (shard boundaries not multiples of 256, empty ranges) touch exactly
    their range, and pi_eval_sweeps ping-pongs like repeated single sweeps."""
    torch = _torch()
    eng, acts, (lo, hi, gshape, strides), states, term, V, pol = _sweep_case(name, shape, cuda_device, 3)
    n = len(V)
    gamma = float(np.float32(envs.ENVS[name].CONFIG["gamma"]))
    d_V, d_pol = _dev(V, cuda_device), _dev(pol, cuda_device)
    d_term = _dev(term.astype(np.uint8), cuda_device)
    chk = H.oracle_for(name)
    cuts = [0, 1, 255, 257, n // 3, n // 3, n - 1, n]
    d_Vn = torch.full((n,), -777.0, dtype=torch.float32, device=cuda_device)
    d_delta = torch.zeros(1, dtype=torch.float32, device=cuda_device)
    o_Vn = np.full(n, -777.0, dtype=np.float32)
    for a, b in zip(cuts[:-1], cuts[1:]):
        eng.eval_sweep(d_V.data_ptr(), d_Vn.data_ptr(), d_pol.data_ptr(), d_term.data_ptr(), a, b,
                       gamma, d_delta.data_ptr())
        torch.cuda.synchronize()
        _, o_delta = chk.eval_sweep(states, acts, pol, V, term, lo, hi, gshape, strides, gamma,
                                    a, b, out=o_Vn)
        assert np.float32(d_delta.item()) == np.float32(o_delta), (a, b)
    H.assert_bits_equal(d_Vn.cpu().numpy(), o_Vn, "piecewise V'")
    # 5 ping-pong sweeps == 5 oracle sweeps, in BOTH buffers: Vb starts as NaN, the first sweep has to put the
    # terminal states' values there, and the later ones (which no longer re-copy them) must leave them alone
    d_A, d_B = d_V.clone(), torch.full_like(d_V, float("nan"))
    eng.eval_sweeps(d_A.data_ptr(), d_B.data_ptr(), d_pol.data_ptr(), d_term.data_ptr(), 0, n,
                    gamma, 5, d_delta.data_ptr())
    torch.cuda.synchronize()
    cur, prev = V, V
    for _ in range(5):
        prev = cur
        cur, o_delta = chk.eval_sweep(states, acts, pol, cur, term, lo, hi, gshape, strides, gamma)
    H.assert_bits_equal(d_B.cpu().numpy(), cur, "5 ping-pong sweeps")
    H.assert_bits_equal(d_A.cpu().numpy(), prev, "the iterate before (other buffer)")
    assert np.float32(d_delta.item()) == np.float32(o_delta)
    # piecewise improvement
    d_changed = torch.zeros(1, dtype=torch.int32, device=cuda_device)
    total = 0
    for a, b in zip(cuts[:-1], cuts[1:]):
        eng.improve_sweep(d_V.data_ptr(), d_pol.data_ptr(), d_term.data_ptr(), a, b, gamma,
                          d_changed.data_ptr())
        total += int(d_changed.item())
    o_pol, o_changed = chk.improve_sweep(states, acts, pol, V, term, lo, hi, gshape, strides, gamma)
    assert np.array_equal(d_pol.cpu().numpy(), o_pol)
    assert total == o_changed
    # idempotence: improving again against the same V changes nothing
    eng.improve_sweep(d_V.data_ptr(), d_pol.data_ptr(), d_term.data_ptr(), 0, n, gamma,
                      d_changed.data_ptr())
    assert int(d_changed.item()) == 0
    eng.close()


def test_c1_full_run_matches_oracle_and_reference_counts(cuda_device):
    """BASELINE config C1 (Pendulum 50x50, 11 torques) through the public solver API: V, policy
    and sweep counts equal the oracle's run; the reference-text golden (glibc libm) agrees on
    the iteration structure and to the stated tolerance."""
    g = np.load(H.GOLDEN / "pendulum_c1_run.npz")
    cfg = envs.CudaPIConfig(**envs.PendulumCuda.CONFIG)
    solver = envs.PendulumCuda(envs.PendulumCuda.bins_space(50), g["actions"], cfg, device=cuda_device)
    solver.run()
    bins = [g["bins0"], g["bins1"]]
    lo, hi, gshape, strides = oracle.grid_metadata(bins)
    states = oracle.states_from_bins(bins)
    ref = H.oracle_for("pendulum").run(states, g["actions"], np.zeros(len(states), bool), lo, hi,
                                       gshape, strides, gamma=cfg.gamma, theta=cfg.theta,
                                       max_eval_iter=cfg.max_eval_iter, max_pi_iter=cfg.max_pi_iter)
    assert solver.stats["eval_sweeps"] == ref["eval_sweeps"]
    assert solver.stats["pi_iterations"] == ref["outer_iterations"]
    assert np.array_equal(solver.policy, ref["policy"])
    H.assert_bits_equal(solver.value_function, ref["value_function"], "C1 V")
    # reference-text golden
    assert np.mean(solver.policy == g["policy"]) >= 0.999                        # measured: identical
    assert np.max(np.abs(solver.value_function - g["value_function"])) <= 2e-6 * np.max(np.abs(g["value_function"]))  # measured 4.3e-7


def test_c2_full_run_matches_oracle(cuda_device):
    """BASELINE config C2: Pendulum 200x200, 21 torques, fp32, full run() on one MI355X."""
    cls = envs.PendulumCuda
    cfg = envs.CudaPIConfig(**cls.CONFIG)
    solver = envs.make("pendulum", 200, device=cuda_device)
    solver.run()
    # the run compared with the oracle below IS the one-launch path (pi_policy_iteration on the XCD-local kernel): one whole
    # run launched, no fallback to the round-by-round loop (pi_info 33 / 32, read when run() released the device)
    assert solver._backend.whole_runs == 1 and solver._backend.xcd_fallbacks == 0
    bins = H.env_bins("pendulum", (200, 200))
    lo, hi, gshape, strides = oracle.grid_metadata(bins)
    states = oracle.states_from_bins(bins)
    ref = H.oracle_for("pendulum").run(states, cls.ACTIONS, np.zeros(len(states), bool), lo, hi,
                                       gshape, strides, gamma=cfg.gamma, theta=cfg.theta,
                                       max_eval_iter=cfg.max_eval_iter, max_pi_iter=cfg.max_pi_iter)
    assert solver.stats["eval_sweeps"] == ref["eval_sweeps"]
    assert solver.stats["pi_iterations"] == ref["outer_iterations"]
    assert np.array_equal(solver.policy, ref["policy"])
    H.assert_bits_equal(solver.value_function, ref["value_function"], "C2 V")




def test_error_paths(cuda_device):
    eng, bins, acts = _engine("pendulum", (16, 16), cuda_device)
    torch = _torch()
    v = torch.zeros(256, dtype=torch.float32, device=cuda_device)
    p = torch.zeros(256, dtype=torch.int32, device=cuda_device)
    t = torch.zeros(256, dtype=torch.uint8, device=cuda_device)
    with pytest.raises(_native.NativeError, match="outside"):
        eng.eval_sweep(v.data_ptr(), v.clone().data_ptr(), p.data_ptr(), t.data_ptr(), 0, 257, 0.99)
    with pytest.raises(_native.NativeError, match="different buffers"):
        eng.eval_sweep(v.data_ptr(), v.data_ptr(), p.data_ptr(), t.data_ptr(), 0, 256, 0.99)
    eng.close()
    bad = _native.Engine(2, [16, 16], [0, 0], [1, 1], [np.linspace(0, 1, 16)] * 2, [0.0], device=0)
    with pytest.raises(_native.NativeError, match="compil"):
        bad.compile("__device__ void step_dynamics(float a) { this is not C }")
    with pytest.raises(_native.NativeError, match="pi_compile has not been called"):
        bad.eval_sweep(v.data_ptr(), v.clone().data_ptr(), p.data_ptr(), t.data_ptr(), 0, 256, 0.99)
    bad.close()


@pytest.mark.parametrize("name,shape", [("pendulum", (41, 13)), ("double_pendulum_swingup", (20, 9, 10, 9)),
                                         ("double_cartpole", (6, 4, 5, 4, 5, 4))])
def test_reach_planes_matches_cpu_restatement(name, shape, cuda_device):
    """pi_reach_planes (what the multi-GPU halo exchange is planned from) == the CPU restatement
    used in the gloo tests, for whole grids and for shard sub-ranges."""
    from dynamicprogramming_amd.solver import HipSweepBackend
    torch = _torch()
    cls = envs.ENVS[name]
    bins = H.env_bins(name, shape)
    lo, hi, gshape, strides = oracle.grid_metadata(bins)
    args = (cls._D, gshape, lo, hi, bins, cls.ACTIONS, envs.dynamics_source(name))
    gpu = HipSweepBackend(*args, device=cuda_device)
    cpu = H.OracleSweepBackend(*args)
    states = oracle.states_from_bins(bins)
    term, _ = H.terminal_mask(name, states)
    n = len(states)
    d_term = _dev(term.astype(np.uint8), cuda_device)
    c_term = torch.from_numpy(term.astype(np.uint8))
    for a, b in [(0, n), (0, n // 3), (n // 3, 2 * n // 3 + 5), (n - 7, n)]:
        got = gpu.reach_planes(d_term, a, b, int(gshape[0]))
        want = cpu.reach_planes(c_term, a, b, int(gshape[0]))
        assert np.array_equal(got, want), (name, a, b)   # planes of all 2^D corners
        # the finer units the exchange is planned on: planes (depth 1) and rows (i0, i1) (depth 2)
        for depth in range(1, gpu.engine.reach_depth_max() + 1):
            got_u = gpu.reach_units(d_term, a, b, depth)
            want_u = cpu.reach_units(c_term, a, b, depth)
            assert np.array_equal(got_u, want_u), (name, a, b, depth)
            if depth == 1:
                assert np.array_equal(got_u, want)
            else:                                        # rows refine planes: same planes, fewer values
                rows = got_u.reshape(int(gshape[0]), int(gshape[1]))
                assert np.array_equal(rows.any(axis=1), want)
    assert gpu.engine.reach_depth_max() == (2 if len(shape) >= 3 else 1)
    gpu.close()


@pytest.mark.parametrize("name,shape", [("pendulum", (33, 29)), ("cartpole", (9, 7, 11, 5)),
                                         ("double_cartpole_swingup", (5, 4, 6, 4, 5, 4))])
def test_value_sweep_bit_exact(name, shape, cuda_device):
    """pi_value_sweep (fused max-backup) == oracle: V', policy, residual, changed count."""
    torch = _torch()
    eng, acts, (lo, hi, gshape, strides), states, term, V, pol = _sweep_case(name, shape, cuda_device, 5)
    n = len(V)
    gamma = float(np.float32(envs.ENVS[name].CONFIG["gamma"]))
    d_V, d_pol = _dev(V, cuda_device), _dev(pol, cuda_device)
    d_term = _dev(term.astype(np.uint8), cuda_device)
    d_Vn = torch.full((n,), -9.0, dtype=torch.float32, device=cuda_device)
    d_delta = torch.zeros(1, dtype=torch.float32, device=cuda_device)
    d_changed = torch.zeros(1, dtype=torch.int32, device=cuda_device)
    a, b = 3, n - 5
    eng.value_sweep(d_V.data_ptr(), d_Vn.data_ptr(), d_pol.data_ptr(), d_term.data_ptr(), a, b, gamma,
                    d_delta.data_ptr(), d_changed.data_ptr())
    torch.cuda.synchronize()
    o_Vn = np.full(n, -9.0, dtype=np.float32)
    _, o_pol, o_delta, o_changed = H.oracle_for(name).value_sweep(
        states, acts, pol, V, term, lo, hi, gshape, strides, gamma, a, b, out=o_Vn)
    H.assert_bits_equal(d_Vn.cpu().numpy(), o_Vn, "value sweep V'")
    assert np.array_equal(d_pol.cpu().numpy(), o_pol)
    assert np.float32(d_delta.item()) == np.float32(o_delta) and int(d_changed.item()) == o_changed
    eng.close()




@pytest.mark.parametrize("name", ["cartpole", "double_pendulum_swingup", "double_cartpole"])
def test_small_4d_6d_runs_against_reference_text_goldens(name, cuda_device):
    """The GPU's full run() on the small 4-D / 6-D grids of tests/golden/small_runs.npz (generated from
    the reference's own kernel text, glibc libm): same number of outer iterations, |dV| <= 1e-5 max|V|,
    policy >= 99.9 % (measured: identical for cartpole and double cartpole, 3 of 4 096 entries
    differ for the double pendulum)."""
    g = np.load(H.GOLDEN / "small_runs.npz")
    shape = tuple(int(x) for x in g[f"{name}_shape"])
    cls = envs.ENVS[name]
    cfg = dict(cls.CONFIG, max_eval_iter=int(g[f"{name}_max_eval_iter"]), max_pi_iter=int(g[f"{name}_max_pi_iter"]))
    s = cls(H.env_bins_space(name, shape), cls.ACTIONS, envs.CudaPIConfig(**cfg), device=cuda_device)
    s.run()
    gV, gP = g[f"{name}_value_function"], g[f"{name}_policy"]
    assert s.stats["pi_iterations"] == int(g[f"{name}_outer_iterations"])
    assert np.mean(s.policy == gP) >= 0.999
    assert np.max(np.abs(s.value_function - gV)) <= 1e-5 * max(1.0, float(np.abs(gV).max()))


@pytest.mark.parametrize("name,bins", [("double_pendulum_swingup", 15), ("double_cartpole", 7)])
def test_full_run_4d_6d_matches_oracle(name, bins, cuda_device):
    """End-to-end run() with the env's own settings on a 4-D grid (15^4, the reference runner's
    default; ~66 000 sweeps of wrap- and trig-heavy dynamics) and a 6-D grid: V, policy and the
    sweep count of every outer iteration equal the oracle's run."""
    cls = envs.ENVS[name]
    cfg = envs.CudaPIConfig(**cls.CONFIG)
    solver = envs.make(name, bins, device=cuda_device)
    solver.run()
    tables = H.env_bins(name, (bins,) * cls._D)
    lo, hi, gshape, strides = oracle.grid_metadata(tables)
    states = oracle.states_from_bins(tables)
    term, tval = H.terminal_mask(name, states)
    ref = H.oracle_for(name).run(states, cls.ACTIONS, term, lo, hi, gshape, strides, gamma=cfg.gamma,
                                 theta=cfg.theta, max_eval_iter=cfg.max_eval_iter,
                                 max_pi_iter=cfg.max_pi_iter, terminal_value=tval)
    assert solver.stats["sweeps_per_iter"] == list(ref["sweeps_per_iter"])
    assert solver.stats.get("stable") == ref["stable"]
    assert np.array_equal(solver.policy, ref["policy"])
    H.assert_bits_equal(solver.value_function, ref["value_function"], f"{name} full run V")


# ── launch geometry, graphs, guards ─────────────────────────────────────────────────────
@pytest.mark.parametrize("name,shape", [("pendulum", (200, 200)), ("double_pendulum_swingup", (13, 9, 11, 17)),
                                         ("double_cartpole", (5, 4, 6, 3, 5, 7))])
def test_chunks_per_workgroup_change_speed_not_results(name, shape, cuda_device):
    """A workgroup sweeps `cpw` consecutive 256-state chunks with the next chunk's inputs
    prefetched; whatever cpw is (1 .. more chunks than the range has), V', policy, residual and
    changed-count are the oracle's, on ragged sub-ranges too.  Also checks the coordinate probe
    (pi_state_coords as the sweeps walk it) against states_space."""
    torch = _torch()
    eng, acts, (lo, hi, gshape, strides), states, term, V, pol = _sweep_case(name, shape, cuda_device, seed=21)
    n = len(states)
    chk = H.oracle_for(name)
    gamma = float(np.float32(0.97))
    d_V, d_pol, d_term = _dev(V, cuda_device), _dev(pol, cuda_device), _dev(term.astype(np.uint8), cuda_device)
    d_delta = torch.zeros(1, dtype=torch.float32, device=cuda_device)
    d_changed = torch.zeros(1, dtype=torch.int32, device=cuda_device)
    for a, b in ((0, n), (n // 7 + 3, n - n // 5 - 1), (5, 6), (n - 300, n)):
        a, b = max(a, 0), min(b, n)
        o_Vn, o_delta = chk.eval_sweep(states, acts, pol, V, term, lo, hi, gshape, strides, gamma, a, b)
        o_pol, o_changed = chk.improve_sweep(states, acts, pol, V, term, lo, hi, gshape, strides, gamma, a, b)
        for cpw in (1, 2, 3, 8, 64):
            eng.set_option(0, cpw)
            eng.set_option(1, cpw)
            d_Vn = d_V.clone()
            eng.eval_sweep(d_V.data_ptr(), d_Vn.data_ptr(), d_pol.data_ptr(), d_term.data_ptr(), a, b,
                           gamma, d_delta.data_ptr())
            H.assert_bits_equal(d_Vn.cpu().numpy()[a:b], o_Vn[a:b], f"V' cpw={cpw} [{a},{b})")
            assert np.array_equal(d_Vn.cpu().numpy()[:a], V[:a]) and np.array_equal(d_Vn.cpu().numpy()[b:], V[b:])
            H.assert_bits_equal(np.float32(d_delta.item()), np.float32(o_delta), "residual")
            d_p2 = d_pol.clone()
            eng.improve_sweep(d_V.data_ptr(), d_p2.data_ptr(), d_term.data_ptr(), a, b, gamma,
                              d_changed.data_ptr())
            assert np.array_equal(d_p2.cpu().numpy()[a:b], o_pol[a:b])
            assert np.array_equal(d_p2.cpu().numpy()[:a], pol[:a]) and np.array_equal(d_p2.cpu().numpy()[b:], pol[b:])
            assert int(d_changed.item()) == o_changed
            out = torch.full(((b - a) * len(shape),), float("nan"), dtype=torch.float32, device=cuda_device)
            eng.probe_coords(a, b, out.data_ptr(), cpw)
            H.assert_bits_equal(out.cpu().numpy().reshape(-1, len(shape)), states[a:b], "state coordinates")
    eng.close()


@pytest.mark.parametrize("name,shape", [("double_pendulum_swingup", (13, 19, 17, 23)), ("cartpole", (11, 21, 19, 17)),
                                         ("double_cartpole", (5, 4, 6, 5, 7, 8))])
def test_the_strip_schedule_changes_placement_not_results(name, shape, cuda_device, monkeypatch):
    """Round 6: the strip schedule (every XCD takes its eighth of every period of groups, pi_set_option 7) against the
    slab schedule and the oracle — V', residual, policy, changed count on the whole grid and on ragged ranges that start
    inside a period, state ranges and the live-state list alike, for periods that are and are not whole numbers of
    groups; pi_probe_coords walks the same schedule and must write every state of the range exactly once."""
    torch = _torch()
    eng, acts, (lo, hi, gshape, strides), states, term, V, pol = _sweep_case(name, shape, cuda_device, seed=31)
    n, D = len(states), len(shape)
    chk = H.oracle_for(name)
    gamma = float(np.float32(0.96))
    d_V, d_pol, d_term = _dev(V, cuda_device), _dev(pol, cuda_device), _dev(term.astype(np.uint8), cuda_device)
    d_delta = torch.zeros(1, dtype=torch.float32, device=cuda_device)
    d_changed = torch.zeros(1, dtype=torch.int32, device=cuda_device)
    plane = n // shape[0]
    used = 0
    for a, b in ((0, n), (n // 9 + 5, n - n // 7), (plane * 2 + 256, n)):
        o_Vn, o_delta = chk.eval_sweep(states, acts, pol, V, term, lo, hi, gshape, strides, gamma, a, b)
        o_pol, o_changed = chk.improve_sweep(states, acts, pol, V, term, lo, hi, gshape, strides, gamma, a, b)
        for period, cpw in ((plane, 1), (plane, 2), (256 * 16, 1), (256 * 37 + 100, 3), (0, 2)):
            eng.set_option(7, period)
            eng.set_option(0, cpw)
            eng.set_option(1, cpw)
            assert eng.info(35) == period
            used += eng.plan_schedule(256, a, b - a, chunks_per_workgroup=cpw)["period"] > 0
            d_Vn = d_V.clone()
            eng.eval_sweep(d_V.data_ptr(), d_Vn.data_ptr(), d_pol.data_ptr(), d_term.data_ptr(), a, b, gamma, d_delta.data_ptr())
            H.assert_bits_equal(d_Vn.cpu().numpy()[a:b], o_Vn[a:b], f"V' period={period} cpw={cpw} [{a},{b})")
            assert np.array_equal(d_Vn.cpu().numpy()[:a], V[:a]) and np.array_equal(d_Vn.cpu().numpy()[b:], V[b:])
            H.assert_bits_equal(np.float32(d_delta.item()), np.float32(o_delta), "residual")
            d_p2 = d_pol.clone()
            eng.improve_sweep(d_V.data_ptr(), d_p2.data_ptr(), d_term.data_ptr(), a, b, gamma, d_changed.data_ptr())
            assert np.array_equal(d_p2.cpu().numpy()[a:b], o_pol[a:b])
            assert np.array_equal(d_p2.cpu().numpy()[:a], pol[:a]) and np.array_equal(d_p2.cpu().numpy()[b:], pol[b:])
            assert int(d_changed.item()) == o_changed
            out = torch.full(((b - a) * D,), float("nan"), dtype=torch.float32, device=cuda_device)
            eng.probe_coords(a, b, out.data_ptr(), cpw)
            H.assert_bits_equal(out.cpu().numpy().reshape(-1, D), states[a:b], "state coordinates")
    assert used >= 5, "the strip schedule was hardly ever taken"
    # batches of sweeps (graphs on this size; the live-state list on grids with terminal states) and the value sweep
    eng.set_option(0, 1)
    eng.set_option(1, 1)
    if term.any():
        monkeypatch.setenv("PI_MI355_LIVE_MIN", "1")
        eng.set_option(6, 1)
        assert eng.prepare_mask(d_term.data_ptr()) == int((~term).sum())
    res = {}
    for period in (0, plane, 256 * 23):
        eng.set_option(7, period)
        d_a, d_b = d_V.clone(), d_V.clone()
        eng.eval_sweeps(d_a.data_ptr(), d_b.data_ptr(), d_pol.data_ptr(), d_term.data_ptr(), 0, n, gamma, 7, d_delta.data_ptr())
        d_p2, d_c = d_pol.clone(), d_V.clone()
        eng.improve_sweep(d_b.data_ptr(), d_p2.data_ptr(), d_term.data_ptr(), 0, n, gamma, d_changed.data_ptr())
        ch = int(d_changed.item())
        eng.value_sweep(d_b.data_ptr(), d_c.data_ptr(), d_p2.data_ptr(), d_term.data_ptr(), 0, n, gamma, d_delta.data_ptr(),
                        d_changed.data_ptr())
        torch.cuda.synchronize()
        res[period] = (d_a.cpu().numpy(), d_b.cpu().numpy(), d_p2.cpu().numpy(), ch, d_c.cpu().numpy(), float(d_delta.item()))
    for period in (plane, 256 * 23):
        for x, y in zip(res[0], res[period]):
            assert H.bits_equal(np.asarray(x), np.asarray(y)), f"period {period} differs from the slab schedule"
    cur = V.copy()
    for _ in range(7):
        prev = cur
        cur, _dl = chk.eval_sweep(states, acts, pol, cur, term, lo, hi, gshape, strides, gamma)
    H.assert_bits_equal(res[plane][1], cur, "7-sweep batch under strips vs the oracle")
    H.assert_bits_equal(res[plane][0], prev, "6th iterate under strips vs the oracle")
    eng.close()


@pytest.mark.parametrize("name,shape", [("pendulum", (50, 50)), ("mountain_car", (111, 110)),
                                        ("cartpole", (8, 7, 9, 8)), ("overhead_crane", (8, 8, 8, 8)),
                                        ("double_pendulum_swingup", (7, 6, 9, 5)),
                                        ("double_cartpole", (3, 3, 4, 3, 3, 3)),
                                        ("double_cartpole_swingup", (2, 3, 3, 4, 3, 4))])
def test_small_grids_run_whole_batches_in_lds(name, shape, cuda_device):
    """Grids that fit one workgroup's LDS and registers run a whole batch of evaluation sweeps in
    ONE launch (pi_eval_resident_kernel); both iterates the caller sees and the residual equal the
    sweep-by-sweep launches bit for bit, for odd, even and minimal batch lengths, with terminal
    states and done successors in the grid."""
    torch = _torch()
    eng, acts, (lo, hi, gshape, strides), states, term, V, pol = _sweep_case(name, shape, cuda_device, seed=11)
    n = len(states)
    assert eng.info(13) > 0 and eng.info(14) == 1, "this grid should qualify for the resident kernel"
    gamma = float(np.float32(0.985))
    d_pol, d_term = _dev(pol, cuda_device), _dev(term.astype(np.uint8), cuda_device)
    junk = np.random.default_rng(1).standard_normal(n).astype(np.float32)
    for n_sweeps in (2, 3, 25, 26):
        out = {}
        for resident in (1, 0):
            eng.set_option(3, resident)
            eng.set_option(2, 0)                   # the comparison path: plain launches, no graph
            A, B = _dev(V, cuda_device), _dev(junk, cuda_device)
            d_delta = torch.full((1,), -1.0, dtype=torch.float32, device=cuda_device)
            eng.eval_sweeps(A.data_ptr(), B.data_ptr(), d_pol.data_ptr(), d_term.data_ptr(), 0, n, gamma,
                            n_sweeps, d_delta.data_ptr())
            torch.cuda.synchronize()
            out[resident] = (A.cpu().numpy(), B.cpu().numpy(), np.float32(d_delta.item()))
        H.assert_bits_equal(out[1][0], out[0][0], f"Va after {n_sweeps} sweeps")
        H.assert_bits_equal(out[1][1], out[0][1], f"Vb after {n_sweeps} sweeps")
        H.assert_bits_equal(out[1][2], out[0][2], "residual")
    # against the oracle as well (one batch of 3)
    chk = H.oracle_for(name)
    cur = V.copy()
    for _ in range(3):
        cur, o_delta = chk.eval_sweep(states, acts, pol, cur, term, lo, hi, gshape, strides, gamma)
    eng.set_option(3, 1)
    A, B = _dev(V, cuda_device), torch.zeros(n, dtype=torch.float32, device=cuda_device)
    d_delta = torch.zeros(1, dtype=torch.float32, device=cuda_device)
    eng.eval_sweeps(A.data_ptr(), B.data_ptr(), d_pol.data_ptr(), d_term.data_ptr(), 0, n, gamma, 3,
                    d_delta.data_ptr())
    H.assert_bits_equal(B.cpu().numpy(), cur, "third iterate vs oracle")
    H.assert_bits_equal(np.float32(d_delta.item()), np.float32(o_delta), "residual vs oracle")
    # a partial range never takes the resident path (it would see the wrong values outside the range)
    a, b = n // 5, n - n // 7
    res = {}
    for resident in (1, 0):
        eng.set_option(3, resident)
        A, B = _dev(V, cuda_device), _dev(junk, cuda_device)
        eng.eval_sweeps(A.data_ptr(), B.data_ptr(), d_pol.data_ptr(), d_term.data_ptr(), a, b, gamma, 4, 0)
        res[resident] = (A.cpu().numpy(), B.cpu().numpy())
    H.assert_bits_equal(res[1][0], res[0][0], "partial range Va")
    H.assert_bits_equal(res[1][1], res[0][1], "partial range Vb")
    eng.close()


@pytest.mark.parametrize("name,bins,max_eval,kernel", [
    ("pendulum", 50, 5000, "lds"), ("mountain_car", 64, 300, "lds"), ("cartpole", 8, 5000, "lds"),
    ("double_pendulum_swingup", 8, 777, "lds"),
    # launch-bound grids beyond one CU's LDS: the dataflow kernel (BASELINE config C2 is the first; 113 x 113 has a ragged
    # last workgroup; cartpole 15^4 has terminal states; the double pendulum wraps two angles)
    ("pendulum", 200, 5000, "flow"), ("mountain_car", 113, 700, "flow"), ("cartpole", 15, 1200, "flow"),
    ("double_pendulum_swingup", 15, 777, "flow"),
    # ... and 2-D grids of up to 2^16 states on the CUs of one XCD (pi_xcd_kernel; 113 x 113: ragged workgroups, terminal
    # states; the ring of 16 versions wraps many times): one evaluation per launch, and the whole run() in one launch
    ("pendulum", 200, 5000, "xcd"), ("mountain_car", 113, 700, "xcd"), ("continuous_mountain_car", 150, 400, "xcd")])
def test_small_grid_policy_evaluation_in_one_launch(name, bins, max_eval, kernel, cuda_device, monkeypatch):
    """pi_policy_evaluation runs the reference's evaluation loop (sweeps, the residual looked at on
    sweeps 0, 25, 50, ... and the last, stop below theta) in ONE launch — with V in one CU's LDS on grids that fit it,
    as tagged granules flowing between workgroups (pi_eval_flow_kernel) on launch-bound grids beyond, on the CUs of
    one XCD (pi_xcd_kernel) for the 2-D ones among those: same number of sweeps, same residuals, same V — and the same
    full run(), which the XCD-local kernel does in ONE launch (pi_policy_iteration) — as the host-driven loop."""
    torch = _torch()
    cfg = dict(envs.ENVS[name].CONFIG, max_eval_iter=max_eval, max_pi_iter=6)
    solvers = {}
    monkeypatch.setenv("PI_MI355_XCD", "1" if kernel == "xcd" else "0")
    for resident in ("1", "0"):
        monkeypatch.setenv("PI_MI355_RESIDENT", resident)
        s = envs.make(name, bins, config=envs.CudaPIConfig(**cfg), device=cuda_device)
        assert s._backend.resident == (resident == "1")
        solvers[resident] = s
    a, b = solvers["1"], solvers["0"]
    assert (a._backend.engine.info(13) > 0) == (kernel == "lds")
    assert (a._backend.engine.info(19) > 0) == (kernel in ("flow", "xcd")) and (a._backend.engine.info(30) > 0) == (kernel == "xcd")
    assert b._backend.engine.info(13) == 0 and b._backend.engine.info(19) == 0 and b._backend.engine.info(30) == 0
    # one evaluation from the same start (zero V, zero policy; cartpole has terminal states)
    da, db = a.policy_evaluation(), b.policy_evaluation()
    assert a.stats["sweeps_per_iter"] == b.stats["sweeps_per_iter"]
    H.assert_bits_equal(np.float32(da), np.float32(db), "residual of the evaluation")
    H.assert_bits_equal(a.d_value_function.cpu().numpy(), b.d_value_function.cpu().numpy(), "V after evaluation")
    # the C ABI call by itself: residuals looked at == the host loop's, sweep by sweep
    n = a.n_states
    V = torch.zeros(n, dtype=torch.float32, device=cuda_device)
    V2, scratch = V.clone(), torch.zeros(n, dtype=torch.float32, device=cuda_device)
    sweeps, looked = a._backend.policy_evaluation(V, a.d_policy, a.d_terminal_mask, float(np.float32(0.95)), 1e-3, 130, 25)
    eng = b._backend.engine
    d_delta = torch.zeros(1, dtype=torch.float32, device=cuda_device)
    host_looked, i = [], 0
    src, dst = V2, scratch
    while i < 130:
        check = i if i % 25 == 0 else min((i // 25 + 1) * 25, 129)
        check = min(check, 129)
        k = check - i + 1
        eng.eval_sweeps(src.data_ptr(), dst.data_ptr(), b.d_policy.data_ptr(), b.d_terminal_mask.data_ptr(), 0, n,
                        float(np.float32(0.95)), k, d_delta.data_ptr())
        if k & 1:
            src, dst = dst, src
        i = check + 1
        host_looked.append(np.float32(d_delta.item()))
        if host_looked[-1] < 1e-3:
            break
    assert sweeps == i
    H.assert_bits_equal(np.asarray(looked, np.float32), np.asarray(host_looked, np.float32), "residuals looked at")
    H.assert_bits_equal(V.cpu().numpy(), src.cpu().numpy(), "V of the C ABI call")
    # and whole runs
    for s in (a, b):
        s.run()
    assert a.stats["sweeps_per_iter"] == b.stats["sweeps_per_iter"] and a.stats["pi_iterations"] == b.stats["pi_iterations"]
    assert a.stats["stable"] == b.stats["stable"] and a.stats["eval_sweeps"] == b.stats["eval_sweeps"]
    H.assert_bits_equal(a.value_function, b.value_function, "V after run()")
    assert np.array_equal(a.policy, b.policy)
    if kernel == "lds":
        assert a._backend.whole_runs == 1                  # the run() above was ONE launch (pi_run_resident_kernel)
    if kernel == "xcd":
        # the run() above was ONE launch; the evaluations before it ran in the XCD-local kernel too, none fell back
        assert (a._backend.whole_runs, a._backend.xcd_evaluations, a._backend.xcd_fallbacks) == (1, 2, 0)
        # ... and round by round through the same kernel (one launch per evaluation)
        monkeypatch.setenv("PI_MI355_RESIDENT", "1")
        monkeypatch.setenv("PI_MI355_WHOLE_RUN", "0")
        c = envs.make(name, bins, config=envs.CudaPIConfig(**cfg), device=cuda_device)
        c.policy_evaluation()
        c.run()
        assert c.stats["sweeps_per_iter"] == b.stats["sweeps_per_iter"] and c.stats["stable"] == b.stats["stable"]
        assert (c._backend.whole_runs, c._backend.xcd_evaluations, c._backend.xcd_fallbacks) == (0, 1 + c.stats["pi_iterations"], 0)
        H.assert_bits_equal(c.value_function, b.value_function, "V after run(), round by round")
        assert np.array_equal(c.policy, b.policy)


@pytest.mark.parametrize("name,bins,rounds,max_eval,theta,kernel", [
    ("pendulum", 200, 3, 60, 1e-4, "xcd"), ("mountain_car", 113, 40, 400, 1e-3, "xcd"), ("continuous_mountain_car", 150, 2, 1, 1e-4, "xcd"),
    # the bigger ones of the grids one CU holds run on the XCD too (32 CUs are faster than one beyond ~4 000 states) ...
    ("pendulum", 80, 4, 80, 1e-4, "xcd"), ("mountain_car", 100, 40, 400, 1e-3, "xcd"),
    # ... the smaller ones, and every grid of that size when the XCD-local kernel is switched off, in one CU
    # (pi_run_resident_kernel): 12 states per thread, terminal states, 4-D and 6-D
    ("pendulum", 110, 4, 80, 1e-4, "lds"), ("mountain_car", 64, 40, 400, 1e-3, "lds"), ("cartpole", 8, 5, 120, 1e-4, "lds"),
    ("double_cartpole", 3, 3, 30, 1e-4, "lds")])
def test_whole_run_in_one_launch_through_the_c_abi(name, bins, rounds, max_eval, theta, kernel, cuda_device, monkeypatch):
    """pi_policy_iteration from a random V and a random policy, with limits that cut the loop short (3 rounds of at most
    60 sweeps; one sweep per evaluation) and limits that let it converge: rounds done, the stable flag, the sweeps, last
    residual and changed entries of every round, V and the policy equal the same loop driven call by call
    (pi_policy_evaluation + pi_improve_sweep on a second handle whose one-launch kernels are off) — on launch-bound 2-D
    grids (pi_xcd_kernel) and on grids one CU's LDS holds (pi_run_resident_kernel)."""
    if kernel == "lds":
        monkeypatch.setenv("PI_MI355_XCD", "0")
    torch = _torch()
    s = envs.make(name, bins, device=cuda_device)
    assert s._backend.whole_run and (s._backend.engine.info(30) > 0) == (kernel == "xcd")
    assert (s._backend.engine.info(13) > 0) == (s.n_states <= {2: 12288, 4: 4096, 6: 1024}[len(s.grid_shape)])
    n = s.n_states
    gamma = float(np.float32(0.97))
    gen = torch.Generator(device="cpu").manual_seed(5)
    term = s._mask_arg()
    V0 = torch.randn(n, generator=gen, dtype=torch.float32).to(cuda_device)
    if term is not None:
        V0[term[:n].bool()] = 0.0
    pol0 = torch.randint(0, s.n_actions, (n,), generator=gen, dtype=torch.int32).to(cuda_device)
    V, pol = V0.clone(), pol0.clone()
    got = s._backend.policy_iteration(V, pol, term, gamma, theta, max_eval, 25, rounds)
    assert got is not None
    assert s._backend.engine.info(33) == 1 and s._backend.engine.info(32) == 0
    # the same loop, call by call, sweep batch by sweep batch
    monkeypatch.setenv("PI_MI355_RESIDENT", "0")
    ref = envs.make(name, bins, device=cuda_device)
    assert not ref._backend.resident
    Vr, polr = V0.clone(), pol0.clone()
    scratch = torch.zeros_like(Vr)
    d_delta = torch.zeros(1, dtype=torch.float32, device=cuda_device)
    d_changed = torch.zeros(1, dtype=torch.int32, device=cuda_device)
    want, stable = [], False
    eng = ref._backend.engine
    for _ in range(rounds):
        sweeps, last = 0, None
        for i in range(max_eval):                         # the reference's loop (:300-336), one launch per sweep
            look = i % 25 == 0 or i == max_eval - 1
            eng.eval_sweep(Vr.data_ptr(), scratch.data_ptr(), polr.data_ptr(), ref._backend._ptr(term), 0, n, gamma,
                           d_delta.data_ptr() if look else 0)
            Vr, scratch = scratch, Vr
            sweeps = i + 1
            if look:
                last = float(d_delta.item())
                if last < theta:
                    break
        ref._backend.improve_sweep(Vr, polr, term, 0, n, gamma, d_changed)
        want.append((sweeps, last, int(d_changed.item())))
        if want[-1][2] == 0:
            stable = True
            break
    assert got[0] == len(want) and got[1] == stable
    assert [(a, np.float32(b).tobytes(), c) for a, b, c in got[2]] == [(a, np.float32(b).tobytes(), c) for a, b, c in want]
    H.assert_bits_equal(V.cpu().numpy(), Vr.cpu().numpy(), "V after the run")
    assert torch.equal(pol, polr)
    for x in (s, ref):
        x._backend.close()


def test_whole_run_falls_back_when_it_cannot_be_placed(cuda_device, monkeypatch):
    """Every wait of the XCD-local kernel is bounded and nothing is written before a clean end: with a time limit no
    barrier can meet (100 ns) the one-launch run reports failure with V and the policy untouched (and, round 6, is COUNTED:
    pi_set_option 8), run() goes on round by round, the first evaluation fails the same way — the second failure: the form
    is switched off — and the dataflow kernel finishes the run with the results of the sweep-by-sweep loop."""
    monkeypatch.setenv("PI_MI355_XCD_TIMEOUT", "0.0000001")
    cfg = envs.CudaPIConfig(**dict(envs.ENVS["mountain_car"].CONFIG, max_pi_iter=4))
    s = envs.make("mountain_car", 113, config=cfg, device=cuda_device)
    assert s._backend.whole_run
    s.run()
    assert (s._backend.whole_runs, s._backend.xcd_evaluations, s._backend.xcd_fallbacks) == (1, 1, 2)
    # a grid one CU's LDS holds as well (pendulum 80 x 80 prefers the 32 CUs of an XCD): the failed launch is followed by
    # the LDS-resident whole-run kernel, which cannot fail — still no round-by-round loop
    cfg2 = envs.CudaPIConfig(**dict(envs.ENVS["pendulum"].CONFIG, max_pi_iter=4))
    s2 = envs.make("pendulum", 80, config=cfg2, device=cuda_device)
    assert s2._backend.whole_run and s2._backend.engine.info(30) > 0 and s2._backend.engine.info(13) > 0
    s2.run()
    assert (s2._backend.whole_runs, s2._backend.xcd_evaluations, s2._backend.xcd_fallbacks) == (2, 0, 1)
    monkeypatch.delenv("PI_MI355_XCD_TIMEOUT")
    monkeypatch.setenv("PI_MI355_RESIDENT", "0")
    for solver, (name, bins, c) in ((s, ("mountain_car", 113, cfg)), (s2, ("pendulum", 80, cfg2))):
        plain = envs.make(name, bins, config=c, device=cuda_device)
        plain.run()
        assert solver.stats["sweeps_per_iter"] == plain.stats["sweeps_per_iter"] and solver.stats["stable"] == plain.stats["stable"]
        H.assert_bits_equal(solver.value_function, plain.value_function, f"{name}: V after the fallback")
        assert np.array_equal(solver.policy, plain.policy)


def test_a_subclass_with_its_own_improvement_step_is_called_round_by_round(cuda_device):
    """run() takes the one-launch path only when the loop is the reference's own: a plugin that overrides
    policy_improvement (or policy_evaluation) sees its method called every round."""
    calls = []

    class Counting(envs.ENVS["pendulum"]):
        def policy_improvement(self):
            calls.append(len(calls))
            return super().policy_improvement()

    cfg = envs.CudaPIConfig(**dict(envs.ENVS["pendulum"].CONFIG, max_pi_iter=3))
    s = Counting(Counting.bins_space(200), Counting.ACTIONS, cfg, device=cuda_device)
    assert s._backend.whole_run
    s.run()
    assert calls == [0, 1, 2] and s._backend.whole_runs == 0 and s._backend.xcd_evaluations == 3


def test_small_batches_replay_as_graphs(cuda_device):
    """pi_eval_sweeps on a launch-bound range builds one hipGraph per argument set and replays it;
    results equal the eager launches' (graphs off) bit for bit, for both ping-pong orders."""
    torch = _torch()
    name, shape = "pendulum", (64, 48)
    eng, acts, (lo, hi, gshape, strides), states, term, V, pol = _sweep_case(name, shape, cuda_device, seed=3)
    n = len(states)
    gamma = float(np.float32(0.99))
    d_pol, d_term = _dev(pol, cuda_device), _dev(term.astype(np.uint8), cuda_device)
    results = {}
    eng.set_option(3, 0)                           # keep the LDS-resident kernel out of this test
    for graphs in (1, 0):
        eng.set_option(2, graphs)
        A, B = _dev(V, cuda_device), torch.zeros(n, dtype=torch.float32, device=cuda_device)
        d_delta = torch.zeros(1, dtype=torch.float32, device=cuda_device)
        deltas = []
        for rep in range(3):                       # 25 + 25 + 25 sweeps: the same graph replayed
            eng.eval_sweeps(A.data_ptr(), B.data_ptr(), d_pol.data_ptr(), d_term.data_ptr(), 0, n, gamma,
                            25, d_delta.data_ptr())
            A, B = B, A                            # odd batch: newest iterate is in B
            deltas.append(float(d_delta.item()))
        results[graphs] = (A.cpu().numpy(), B.cpu().numpy(), deltas)
        if graphs:
            assert eng.info(9) == 2                # (A,B) and (B,A)
        else:
            assert eng.info(10) == 0
    H.assert_bits_equal(results[1][0], results[0][0], "newest iterate, graph vs eager")
    H.assert_bits_equal(results[1][1], results[0][1], "previous iterate, graph vs eager")
    assert results[1][2] == results[0][2]
    chk = H.oracle_for(name)
    a, b = V.copy(), np.zeros_like(V)
    for _ in range(75):
        b, dl = chk.eval_sweep(states, acts, pol, a, term, lo, hi, gshape, strides, gamma, 0, n)
        a, b = b, a
    H.assert_bits_equal(results[1][0], a, "75 sweeps vs the oracle")
    eng.close()


def test_interpolation_division_guard_edges(cuda_device):
    """The reciprocal-multiply division of the interpolation is only taken for 2^-40 <= |s - lo|
    (or s - lo == 0 exactly: a successor clamped onto a lower bound, trivially exact on that path)
    and sum |s - lo| < 2^40; denormal, tiny, huge, infinite and NaN coordinates must take the IEEE
    path; every point lands where the reference's arithmetic puts it (indices exact, weights
    bit-exact) whichever path its lane takes."""
    torch = _torch()
    name, shape = "cartpole", (9, 7, 11, 5)
    eng, bins, acts = _engine(name, shape, cuda_device)
    assert [eng.info(20 + d) for d in range(4)] == [1, 1, 1, 1]
    lo, hi, gshape, strides = oracle.grid_metadata(bins)
    rng = np.random.default_rng(5)
    pts = H.sample_states(rng, bins, 4096)
    specials = np.array([0.0, -0.0, 1e-45, -1e-45, 1e-30, 3e-13, 1e12, 1.2e12, 3e38, -3e38,
                         np.inf, -np.inf, np.nan], dtype=np.float32)
    for k, v in enumerate(specials):              # each special in each dimension, alone and together
        for d in range(4):
            pts[16 * k + d, d] = lo[d] + v if np.isfinite(v) and abs(v) < 1e20 else v
        pts[16 * k + 4, :] = v
        pts[16 * k + 5, :] = lo            # s - lo == 0 exactly
        pts[16 * k + 6, :] = hi
        # on a lower bound in some dimensions (fast path admits a == 0) and `v` away from it in the others
        pts[16 * k + 7, :] = lo
        pts[16 * k + 7, 1::2] = (lo[1::2] + v) if np.isfinite(v) and abs(v) < 1e20 else v
        pts[16 * k + 8, :] = lo
        pts[16 * k + 8, 0] = 0.5 * (lo[0] + hi[0])   # one ordinary coordinate, three exactly on lo
    chk = H.oracle_for(name)
    o_idx, o_w = chk.interp(pts, lo, hi, gshape, strides)
    d_pts = _dev(pts, cuda_device)
    d_idx = torch.zeros((len(pts), 16), dtype=torch.int32, device=cuda_device)
    d_w = torch.zeros((len(pts), 16), dtype=torch.float32, device=cuda_device)
    eng.probe_interp(d_pts.data_ptr(), d_idx.data_ptr(), d_w.data_ptr(), len(pts))
    torch.cuda.synchronize()
    assert np.array_equal(d_idx.cpu().numpy(), o_idx)
    got, want = d_w.cpu().numpy(), o_w
    same = (got.view(np.uint32) == want.view(np.uint32)) | (np.isnan(got) & np.isnan(want))
    assert same.all(), np.argwhere(~same)[:5]
    eng.close()


# ── the sharded driver of the library over the in-process transport ─────────────────────
@pytest.mark.parametrize("mode", ["halo", "allgather"])
@pytest.mark.parametrize("world,name,shape", [(2, "pendulum", (41, 13)),
                                               (3, "double_pendulum_swingup", (14, 9, 11, 8)),
                                               (2, "double_cartpole", (6, 4, 5, 4, 5, 4)),
                                               (4, "cartpole_swingup", (18, 7, 9, 8)),
                                               (8, "double_pendulum_swingup", (40, 6, 8, 6))])   # C4 @ 8 in small
def test_sharded_driver_local_transport(world, name, shape, mode, cuda_device, monkeypatch):
    """pi_eval_sweeps_sharded / pi_improve_sweep_sharded / pi_exchange_plan (csrc/pi_comm.cpp) with
    `world` logical ranks on ONE GPU: one host thread, one stream, one set of V buffers per rank,
    exchanging through the in-process transport (device copies ordered by HIP events, the
    stream semantics of RCCL send/recv).  Exercises the overlap path with real kernels: planes
    peers wait for swept first, exchange on the second stream, interior meanwhile.  Every rank's
    run() must equal the single-rank run bit for bit."""
    import threading
    import uuid
    from dynamicprogramming_amd import transport as T
    torch = _torch()
    monkeypatch.setenv("PI_MI355_EXCHANGE", mode)
    cls = envs.ENVS[name]
    cfg_kw = {**cls.CONFIG, "max_pi_iter": 3, "max_eval_iter": 60}
    single = cls(H.env_bins_space(name, shape), cls.ACTIONS, envs.CudaPIConfig(**cfg_kw), device=cuda_device)
    single.run()
    group = f"test-{uuid.uuid4().hex}"
    out, errors = [None] * world, []

    def rank_main(r):
        try:
            stream = torch.cuda.Stream(device=cuda_device)
            with torch.cuda.stream(stream):
                s = cls(H.env_bins_space(name, shape), cls.ACTIONS, envs.CudaPIConfig(**cfg_kw),
                        device=cuda_device, transport=T.NativeTransport.local(r, world, group))
                info = dict(s._comm.info)
                s.run()
            out[r] = (s.value_function, s.policy, list(s.stats["sweeps_per_iter"]), info)
        except Exception as exc:  # noqa: BLE001
            errors.append((r, repr(exc)))

    threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert not errors, errors
    assert all(o is not None for o in out), "a rank did not finish"
    for r in range(world):
        V, pol, sweeps, info = out[r]
        H.assert_bits_equal(V, single.value_function, f"rank {r} V")
        assert np.array_equal(pol, single.policy)
        assert sweeps == single.stats["sweeps_per_iter"]
        assert info["mode"] == mode
        if mode == "halo":
            assert info["send_ranges"] >= 1 and 0 < info["recv_elems"] < (world - 1) * -(-single.n_states // world)




def test_runner_script_end_to_end(cuda_device, tmp_path):
    """`python runners/pendulum_cuda.py --bins 50 --retrain --save-path ...` (the reference's command
    line, README.md:323-347) trains on the GPU and writes an archive with the reference's nine keys
    and dtypes (src/cuda_policy_iteration.py:397-408); a second invocation without --retrain loads
    it; the archive read with the raw key access of runners/hybrid_double_cartpole.py:35-43 drives
    utils.get_optimal_action; V and policy equal the oracle's run."""
    import subprocess
    import sys
    from pathlib import Path
    from utils import get_optimal_action
    root = Path(__file__).resolve().parents[1]
    out = tmp_path / "sub" / "pendulum_policy"            # no suffix, missing parent: both fixed like the reference
    cmd = [sys.executable, str(root / "runners" / "pendulum_cuda.py"), "--bins", "50", "--retrain",
           "--save-path", str(out), "--episodes", "1", "--no-plot"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=tmp_path)
    assert res.returncode == 0, res.stdout + res.stderr
    d = np.load(out.with_suffix(".npz"))
    want = {"value_function": np.float32, "policy": np.int32, "bounds_low": np.float32,
            "bounds_high": np.float32, "grid_shape": np.int32, "strides": np.int32, "corner_bits": np.int32,
            "action_space": np.float32, "states_space": np.float32}
    assert set(d.files) == set(want) and all(d[k].dtype == t for k, t in want.items())
    assert d["states_space"].shape == (2500, 2) and d["corner_bits"].shape == (4, 2)
    a = float(get_optimal_action(np.array([3.0, 0.0], np.float32), d["policy"], d["action_space"],
                                 d["bounds_low"], d["bounds_high"], d["grid_shape"], d["strides"], d["corner_bits"]))
    assert -2.0 <= a <= 2.0
    cls = envs.ENVS["pendulum"]
    bins = [np.asarray(b, np.float32) for b in cls.bins_space(50).values()]
    lo, hi, gshape, strides = oracle.grid_metadata(bins)
    ref = H.oracle_for("pendulum").run(oracle.states_from_bins(bins), cls.ACTIONS, np.zeros(2500, bool), lo, hi,
                                       gshape, strides, **{k: v for k, v in cls.CONFIG.items() if k != "log_interval"})
    H.assert_bits_equal(d["value_function"], ref["value_function"], "runner V vs oracle")
    assert np.array_equal(d["policy"], ref["policy"])
    res2 = subprocess.run(cmd[:2] + ["--save-path", str(out)], capture_output=True, text=True, timeout=300,
                          cwd=tmp_path)
    assert res2.returncode == 0 and "Loading existing policy" in res2.stdout
    # extension flag: the same command line with --value-iteration writes the same archive schema, holding the
    # fixed point of the fused max-backup sweep (what solver.value_iteration() converges to)
    out_vi = tmp_path / "vi_policy.npz"
    res3 = subprocess.run(cmd[:2] + ["--bins", "50", "--retrain", "--value-iteration", "--save-path", str(out_vi)],
                          capture_output=True, text=True, timeout=600, cwd=tmp_path)
    assert res3.returncode == 0 and "value iteration:" in res3.stdout and "converged=True" in res3.stdout, res3.stdout + res3.stderr
    v = np.load(out_vi)
    assert set(v.files) == set(want) and all(v[k].dtype == t for k, t in want.items())
    direct = envs.make("pendulum", 50)
    while not direct.value_iteration() < direct.config.theta:
        pass
    direct._pull_tensors_from_gpu()
    H.assert_bits_equal(v["value_function"], direct.value_function, "runner --value-iteration vs solver.value_iteration()")
    assert np.array_equal(v["policy"], direct.policy)
    assert np.mean(v["policy"] == d["policy"]) > 0.99 and np.abs(v["value_function"] - d["value_function"]).max() < 0.05


def test_rccl_entry_points_world_1(cuda_device):
    """The RCCL side of the C ABI with a one-rank communicator (all a single-GPU box can host):
    pi_comm_unique_id / pi_comm_init, in-place all-gathers, scalar all-reduces, pi_exchange_plan and
    the sharded sweep drivers must reproduce the unsharded entry points bit for bit."""
    torch = _torch()
    name, shape = "double_pendulum_swingup", (12, 9, 11, 10)
    eng, acts, (lo, hi, gshape, strides), states, term, V, pol = _sweep_case(name, shape, cuda_device, seed=9)
    n = len(states)
    gamma = float(np.float32(0.999))
    uid = _native.comm_unique_id()
    assert len(uid) == 128
    eng.comm_init(0, 1, uid)
    assert (eng.comm_info(0), eng.comm_info(1), eng.comm_info(2)) == (0, 1, 1)
    d_pol, d_term = _dev(pol, cuda_device), _dev(term.astype(np.uint8), cuda_device)
    info = eng.exchange_plan(d_term.data_ptr(), n, 0, True)
    assert info["recv_elems"] == 0 and eng.comm_info(3) in (1, 2)
    A, B = _dev(V, cuda_device), torch.zeros(n, dtype=torch.float32, device=cuda_device)
    A2, B2 = A.clone(), B.clone()
    d1 = torch.zeros(1, dtype=torch.float32, device=cuda_device)
    d2 = torch.zeros(1, dtype=torch.float32, device=cuda_device)
    eng.eval_sweeps_sharded(A.data_ptr(), B.data_ptr(), d_pol.data_ptr(), d_term.data_ptr(), gamma, 7, d1.data_ptr())
    eng.eval_sweeps(A2.data_ptr(), B2.data_ptr(), d_pol.data_ptr(), d_term.data_ptr(), 0, n, gamma, 7, d2.data_ptr())
    assert torch.equal(A, A2) and torch.equal(B, B2) and float(d1.item()) == float(d2.item())
    c1 = torch.zeros(1, dtype=torch.int32, device=cuda_device)
    c2 = torch.zeros(1, dtype=torch.int32, device=cuda_device)
    p1, p2 = d_pol.clone(), d_pol.clone()
    eng.improve_sweep_sharded(B.data_ptr(), p1.data_ptr(), d_term.data_ptr(), gamma, c1.data_ptr())
    eng.improve_sweep(B2.data_ptr(), p2.data_ptr(), d_term.data_ptr(), 0, n, gamma, c2.data_ptr())
    assert torch.equal(p1, p2) and int(c1.item()) == int(c2.item())
    before = B.clone()
    eng.allgather_V(B.data_ptr(), n)
    eng.allgather_policy(p1.data_ptr(), n)
    eng.exchange_V(B.data_ptr())
    eng.allreduce_max_f32(d1.data_ptr())
    eng.allreduce_sum_u32(c1.data_ptr())
    torch.cuda.synchronize()
    assert torch.equal(B, before) and torch.equal(p1, p2)
    assert float(d1.item()) == float(d2.item()) and int(c1.item()) == int(c2.item())
    eng.comm_destroy()
    with pytest.raises(_native.NativeError, match="no communicator"):
        eng.allgather_V(B.data_ptr(), n)
    eng.close()


# ── memory order of the dimensions (pi_set_option 4 / solver.MEMORY_ORDER) ───────────────────────────────
ORDER_CASES = [("pendulum", (23, 17), (1, 0)),
               ("cartpole_swingup", (9, 7, 11, 5), (0, 2, 1, 3)),
               ("double_pendulum_swingup", (8, 9, 7, 10), (2, 0, 3, 1)),
               ("overhead_crane", (7, 6, 8, 5), (3, 2, 1, 0)),
               ("double_cartpole", (5, 4, 6, 4, 5, 4), (0, 1, 2, 3, 5, 4)),
               ("double_cartpole_swingup", (4, 5, 4, 3, 5, 4), (0, 2, 3, 1, 4, 5))]


@pytest.mark.parametrize("name,shape,order", ORDER_CASES)
def test_memory_order_sweeps_are_bit_exact(name, shape, order, cuda_device):
    """An engine whose grid lives in another memory order (which dimensions are slow; lanes run along order[-1]) gives
    the oracle's bits: evaluation sweeps (whole grid, ragged ranges of MEMORY-order indices, a batch), the improvement
    and the value sweep, the dynamics probe (user-order I/O) and the interpolation probe (weights in the reference's
    corner order; indices are memory-order flat indices).  Only where a state lives changes."""
    torch = _torch()
    cls = envs.ENVS[name]
    D = cls._D
    bins = H.env_bins(name, shape)
    acts = np.asarray(cls.ACTIONS, np.float32)
    eng = _native.Engine(D, [len(b) for b in bins], [b.min() for b in bins], [b.max() for b in bins], bins, acts,
                         device=cuda_device.index or 0, order=order)
    eng.compile(envs.dynamics_source(name))
    assert eng.order == tuple(order)
    lo, hi, gshape, strides = oracle.grid_metadata(bins)
    states = oracle.states_from_bins(bins)
    term, tval = H.terminal_mask(name, states)
    n = len(states)
    rng = np.random.default_rng(31)
    V = (rng.standard_normal(n) * 3.0).astype(np.float32)
    V[term] = np.float32(tval)
    pol = rng.integers(0, len(acts), size=n).astype(np.int32)
    pol[term] = 0
    gamma = float(np.float32(0.97))
    chk = H.oracle_for(name)
    tm = lambda a: np.ascontiguousarray(eng.to_memory(np.ascontiguousarray(a)))     # noqa: E731
    d_V, d_pol, d_term = _dev(tm(V), cuda_device), _dev(tm(pol), cuda_device), _dev(tm(term.astype(np.uint8)), cuda_device)
    tptr = d_term.data_ptr() if term.any() else 0
    d_delta = torch.zeros(1, dtype=torch.float32, device=cuda_device)
    # whole-grid evaluation sweep + residual
    d_Vn = torch.full((n,), float("nan"), dtype=torch.float32, device=cuda_device)
    eng.eval_sweep(d_V.data_ptr(), d_Vn.data_ptr(), d_pol.data_ptr(), tptr, 0, n, gamma, d_delta.data_ptr())
    torch.cuda.synchronize()
    o_V, o_delta = chk.eval_sweep(states, acts, pol, V, term, lo, hi, gshape, strides, gamma)
    H.assert_bits_equal(eng.to_user(d_Vn.cpu().numpy()), o_V, f"{name} {order} V'")
    assert np.float32(d_delta.item()) == np.float32(o_delta)
    # a ragged range of memory-order indices: exactly those states are written
    a, b = n // 5 + 3, n - n // 7 - 1
    d_Vr = torch.full((n,), float("nan"), dtype=torch.float32, device=cuda_device)
    eng.eval_sweep(d_V.data_ptr(), d_Vr.data_ptr(), d_pol.data_ptr(), tptr, a, b, gamma, 0)
    torch.cuda.synchronize()
    got = d_Vr.cpu().numpy()
    assert np.isnan(got[:a]).all() and np.isnan(got[b:]).all()
    H.assert_bits_equal(got[a:b], tm(o_V)[a:b], f"{name} {order} ragged range")
    # a 4-sweep batch (graph / eager / LDS-resident path, whatever the grid takes)
    Va, Vb = d_V.clone(), torch.zeros_like(d_V)
    eng.eval_sweeps(Va.data_ptr(), Vb.data_ptr(), d_pol.data_ptr(), tptr, 0, n, gamma, 4, d_delta.data_ptr())
    torch.cuda.synchronize()
    ref = V
    for _ in range(4):
        ref, o_delta = chk.eval_sweep(states, acts, pol, ref, term, lo, hi, gshape, strides, gamma)
    H.assert_bits_equal(eng.to_user(Va.cpu().numpy()), ref, f"{name} {order} batch of 4")
    assert np.float32(d_delta.item()) == np.float32(o_delta)
    # improvement and value sweep
    d_p2, d_ch = d_pol.clone(), torch.zeros(1, dtype=torch.int32, device=cuda_device)
    eng.improve_sweep(d_V.data_ptr(), d_p2.data_ptr(), tptr, 0, n, gamma, d_ch.data_ptr())
    torch.cuda.synchronize()
    o_pol, o_changed = chk.improve_sweep(states, acts, pol, V, term, lo, hi, gshape, strides, gamma)
    assert np.array_equal(eng.to_user(d_p2.cpu().numpy()), o_pol) and int(d_ch.item()) == o_changed
    d_p3, d_Vv = d_pol.clone(), torch.zeros_like(d_V)
    eng.value_sweep(d_V.data_ptr(), d_Vv.data_ptr(), d_p3.data_ptr(), tptr, 0, n, gamma, d_delta.data_ptr(), d_ch.data_ptr())
    torch.cuda.synchronize()
    v_V, v_pol, v_delta, v_changed = chk.value_sweep(states, acts, pol, V, term, lo, hi, gshape, strides, gamma)
    H.assert_bits_equal(eng.to_user(d_Vv.cpu().numpy()), v_V, f"{name} {order} value sweep V'")
    assert np.array_equal(eng.to_user(d_p3.cpu().numpy()), v_pol)
    assert np.float32(d_delta.item()) == np.float32(v_delta) and int(d_ch.item()) == v_changed
    # probes: dynamics in the user's argument order; interpolation weights in the reference's corner order
    pts = H.sample_states(rng, bins, 600)
    act = rng.choice(acts, size=len(pts)).astype(np.float32)
    m = len(pts)
    d_next = torch.empty((m, D), dtype=torch.float32, device=cuda_device)
    d_rew = torch.empty(m, dtype=torch.float32, device=cuda_device)
    d_done = torch.empty(m, dtype=torch.uint8, device=cuda_device)
    d_pts, d_act = _dev(pts, cuda_device), _dev(act, cuda_device)      # kept alive until the probes have run
    eng.probe_step(d_pts.data_ptr(), d_act.data_ptr(), d_next.data_ptr(), d_rew.data_ptr(), d_done.data_ptr(), m)
    d_idx = torch.empty((m, 1 << D), dtype=torch.int32, device=cuda_device)
    d_w = torch.empty((m, 1 << D), dtype=torch.float32, device=cuda_device)
    eng.probe_interp(d_pts.data_ptr(), d_idx.data_ptr(), d_w.data_ptr(), m)
    torch.cuda.synchronize()
    o_next, o_rew, o_done = chk.step(pts, act)
    H.assert_bits_equal(d_next.cpu().numpy(), o_next, "probe: next state")
    H.assert_bits_equal(d_rew.cpu().numpy(), o_rew, "probe: reward")
    assert np.array_equal(d_done.cpu().numpy().astype(bool), o_done)
    o_idx, o_w = chk.interp(pts, lo, hi, gshape, strides)
    H.assert_bits_equal(d_w.cpu().numpy(), o_w, "probe: weights")
    mem_of_user_flat = np.empty(n, dtype=np.int64)                     # user flat index -> memory flat index
    mem_of_user_flat[eng.to_memory(np.arange(n, dtype=np.int64))] = np.arange(n, dtype=np.int64)
    assert np.array_equal(d_idx.cpu().numpy(), mem_of_user_flat[o_idx])
    # the coordinate probe: columns in the user's order, rows in memory order
    d_xy = torch.empty((n, D), dtype=torch.float32, device=cuda_device)
    eng.probe_coords(0, n, d_xy.data_ptr(), 2)
    torch.cuda.synchronize()
    assert np.array_equal(d_xy.cpu().numpy(), states[eng.to_memory(np.arange(n, dtype=np.int64))])
    eng.close()


@pytest.mark.parametrize("name,shape,order,world", [("double_pendulum_swingup", (14, 9, 11, 8), (0, 2, 1, 3), 1),
                                                     ("double_pendulum_swingup", (14, 9, 11, 8), (0, 2, 1, 3), 3),
                                                     ("cartpole_swingup", (18, 7, 9, 8), (1, 0, 3, 2), 2),
                                                     ("double_cartpole", (6, 4, 5, 4, 5, 4), (0, 1, 2, 3, 5, 4), 2),
                                                     ("pendulum", (41, 13), (1, 0), 1)])
def test_memory_order_full_runs_equal_the_oracle(name, shape, order, world, cuda_device, monkeypatch):
    """run() of a solver whose device tensors are in another memory order (PI_MI355_ORDER), on one rank and sharded
    over logical ranks (slabs of the SLOWEST MEMORY dimension; halo exchange through the in-process transport):
    value_function, policy and the sweep count of every evaluation equal the oracle's run in the user's order."""
    import threading
    import uuid
    from dynamicprogramming_amd import transport as T
    torch = _torch()
    monkeypatch.setenv("PI_MI355_ORDER", ",".join(map(str, order)))
    monkeypatch.setenv("PI_MI355_EXCHANGE", "halo")
    cls = envs.ENVS[name]
    cfg_kw = {**cls.CONFIG, "max_pi_iter": 3, "max_eval_iter": 60}
    cfg = envs.CudaPIConfig(**cfg_kw)
    tables = H.env_bins(name, shape)
    lo, hi, gshape, strides = oracle.grid_metadata(tables)
    states = oracle.states_from_bins(tables)
    term, tval = H.terminal_mask(name, states)
    ref = H.oracle_for(name).run(states, cls.ACTIONS, term, lo, hi, gshape, strides, gamma=cfg.gamma, theta=cfg.theta,
                                 max_eval_iter=cfg.max_eval_iter, max_pi_iter=cfg.max_pi_iter, terminal_value=tval)
    group = f"order-{uuid.uuid4().hex}"
    out, errors = [None] * world, []

    def rank_main(r):
        try:
            with torch.cuda.stream(torch.cuda.Stream(device=cuda_device)):
                kw = {"transport": T.NativeTransport.local(r, world, group)} if world > 1 else {}
                s = cls(H.env_bins_space(name, shape), cls.ACTIONS, envs.CudaPIConfig(**cfg_kw), device=cuda_device, **kw)
                assert s._order == tuple(order) and s._backend.engine.order == tuple(order)
                s.run()
                out[r] = (s.value_function, s.policy, list(s.stats["sweeps_per_iter"]))
        except Exception as exc:  # noqa: BLE001
            errors.append((r, repr(exc)))

    threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert not errors, errors
    for r in range(world):
        V, pol, sweeps = out[r]
        H.assert_bits_equal(V, ref["value_function"], f"rank {r} V")
        assert np.array_equal(pol, ref["policy"]) and sweeps == list(ref["sweeps_per_iter"])


@pytest.mark.parametrize("name,bins,lane", [("double_cartpole_swingup", 11, 1), ("double_cartpole", 11, 1),
                                             ("double_pendulum_swingup", 24, None), ("pendulum", 40, None)])
def test_memory_order_auto_measures_candidates_derived_from_the_dynamics(name, bins, lane, cuda_device, monkeypatch, tmp_path):
    """PI_MI355_ORDER=auto (and, round 6, every big plugin without a MEMORY_ORDER of its own): a handful of candidate orders
    derived from the plugin's own dynamics — the lane dimension is the one along which a wave's successors stay together;
    for the double cartpole that is the cart's speed, what the exhaustive measurement finds, with the cart's position
    tried second-fastest — each timed on the device; the decision is cached; the run's results are the oracle's whatever
    order wins."""
    monkeypatch.setenv("PI_MI355_ORDER", "auto")
    monkeypatch.setattr(_native, "KERNEL_CACHE", tmp_path)
    cls = envs.ENVS[name]
    D = cls._D
    cfg_kw = {**cls.CONFIG, "max_pi_iter": 2, "max_eval_iter": 40}
    s = cls(cls.bins_space(bins), cls.ACTIONS, envs.CudaPIConfig(**cfg_kw), device=cuda_device)
    order = s._order if s._order is not None else tuple(range(D))
    rec = s._order_tuning
    cands = [tuple(r["order"]) for r in rec["candidates"]]
    assert sorted(order) == list(range(D)) and order in cands and cands[0] == tuple(range(D))
    assert all(sorted(c) == list(range(D)) for c in cands) and len(set(cands)) == len(cands) and len(cands) <= 14
    assert all(r["eval_ms"] > 0 for r in rec["candidates"])
    if lane is not None:
        with_lane = [c for c in cands if c[-1] == lane]
        assert len(with_lane) >= 4 and any(c[-2:] == (0, lane) for c in with_lane)        # (.., x, x_dot) is tried
        assert any(c[-2:] == (lane, 0) for c in cands)                                     # ... and so is (.., x_dot, x)
    # the decision is cached beside the code objects: a second solver reads it instead of measuring again
    files = list(tmp_path.glob("order_*.json"))
    assert len(files) == 1
    s2 = cls(cls.bins_space(bins), cls.ACTIONS, envs.CudaPIConfig(**cfg_kw), device=cuda_device)
    assert s2._order == s._order and s2._order_tuning["candidates"] == rec["candidates"]
    s2._backend.close()
    s.run()
    tables = H.env_bins(name, (bins,) * D)
    lo, hi, gshape, strides = oracle.grid_metadata(tables)
    states = oracle.states_from_bins(tables)
    term, tval = H.terminal_mask(name, states)
    cfg = envs.CudaPIConfig(**cfg_kw)
    ref = H.oracle_for(name).run(states, cls.ACTIONS, term, lo, hi, gshape, strides, gamma=cfg.gamma, theta=cfg.theta,
                                 max_eval_iter=cfg.max_eval_iter, max_pi_iter=cfg.max_pi_iter, terminal_value=tval)
    H.assert_bits_equal(s.value_function, ref["value_function"], f"{name} auto order {order}")
    assert np.array_equal(s.policy, ref["policy"]) and s.stats["sweeps_per_iter"] == list(ref["sweeps_per_iter"])


@pytest.mark.parametrize("name,bins,xcd", [("pendulum", 200, "1"), ("pendulum", 200, "0"), ("cartpole", 15, "1"),
                                           ("pendulum", 50, "1")])
@pytest.mark.parametrize("interval,max_sweeps", [(1, 37), (7, 60), (25, 26), (25, 1), (50, 300)])
def test_one_launch_evaluation_with_any_look_interval(name, bins, xcd, interval, max_sweeps, cuda_device, monkeypatch):
    """pi_policy_evaluation (dataflow kernel on the 200 x 200 and 15^4 grids, LDS-resident on 50 x 50) looks at the
    residual on sweeps 0, k, 2k, ... and the last one for ANY interval k — every sweep (k = 1: the control block holds one
    slot per look and grows with it), an interval that does not divide the limit, a single sweep — and stops at the first
    look below theta: sweeps done, every residual looked at and V equal the same schedule driven sweep by sweep."""
    torch = _torch()
    monkeypatch.setenv("PI_MI355_XCD", xcd)
    s = envs.make(name, bins, device=cuda_device)
    eng = s._backend.engine
    assert s._backend.resident and (eng.info(30) > 0) == ((name, bins, xcd) == ("pendulum", 200, "1"))
    n = s.n_states
    gamma = float(np.float32(0.9))
    gen = torch.Generator(device="cpu").manual_seed(17)
    V0 = torch.randn(n, generator=gen, dtype=torch.float32).to(cuda_device)
    term = s._mask_arg()
    if term is not None:
        V0[term[:n].bool()] = 0.0
    pol = torch.randint(0, s.n_actions, (n,), generator=gen, dtype=torch.int32).to(cuda_device)
    theta = 0.05
    for rounds in range(2):                              # twice: the second evaluation reuses the control block
        V = V0.clone()
        sweeps, looked = s._backend.policy_evaluation(V, pol, term, gamma, theta, max_sweeps, interval)
        A, B = V0.clone(), V0.clone()
        d = torch.zeros(1, dtype=torch.float32, device=cuda_device)
        want_looked, done = [], 0
        for i in range(max_sweeps):
            look = i % interval == 0 or i == max_sweeps - 1
            eng.eval_sweep(A.data_ptr(), B.data_ptr(), pol.data_ptr(), s._backend._ptr(term), 0, n, gamma,
                           d.data_ptr() if look else 0)
            A, B = B, A
            done = i + 1
            if look:
                want_looked.append(np.float32(d.item()))
                if want_looked[-1] < theta:
                    break
        assert sweeps == done
        H.assert_bits_equal(np.asarray(looked, np.float32), np.asarray(want_looked, np.float32), "residuals looked at")
        H.assert_bits_equal(V.cpu().numpy(), A.cpu().numpy(), "V after the evaluation")
    s._backend.close()


def test_one_launch_evaluation_gives_up_loudly_instead_of_hanging(cuda_device, monkeypatch):
    """Every device-side wait of the dataflow kernel is bounded: with a time limit no hand-off can meet (100 ns) a wave
    gives up, raises the status word, every other wave leaves at its next poll, the launch ENDS with *d_sweeps = -1 and —
    round 6 — V untouched: the finish kernel is its only writer.  The solver then runs the evaluation sweep
    by sweep (same sweeps, residual and V as a solver that never had the kernel), stops trying the kernel after two such
    launches, and an evaluation with a sane limit on a fresh solver is unaffected."""
    torch = _torch()
    monkeypatch.setenv("PI_MI355_XCD", "0")              # the XCD-local kernel has its own fallback: the dataflow kernel is the one that gives up
    s = envs.make("pendulum", 200, device=cuda_device)
    assert s._backend.engine.info(19) > 0 and s._backend.engine.info(30) == 0
    n = s.n_states
    gen = torch.Generator(device="cpu").manual_seed(23)
    V0 = torch.randn(n, generator=gen, dtype=torch.float32).to(cuda_device)
    s.d_value_function[:n].copy_(V0)
    s.d_new_value_function.copy_(s.d_value_function)
    monkeypatch.setenv("PI_MI355_FLOW_TIMEOUT", "0.0000001")
    # through the C ABI: the launch ends, reports -1 and leaves V alone
    assert s._backend.policy_evaluation(s.d_value_function, s.d_policy, s._mask_arg(), 0.9, 1e-4, 100, 25) is None
    torch.cuda.synchronize()
    assert torch.equal(s.d_value_function[:n].view(torch.int32), V0.view(torch.int32)), "a failed launch touched V"
    # through the solver: the same evaluation, sweep by sweep
    delta = s.policy_evaluation()
    assert s._backend.one_launch_failures == 2 and not s._backend.resident       # not tried a third time
    monkeypatch.setenv("PI_MI355_RESIDENT", "0")
    plain = envs.make("pendulum", 200, device=cuda_device)
    assert not plain._backend.resident
    plain.d_value_function[:n].copy_(V0)
    plain.d_new_value_function.copy_(plain.d_value_function)
    want = plain.policy_evaluation()
    assert np.float32(delta) == np.float32(want) and s.stats["sweeps_per_iter"][-1] == plain.stats["sweeps_per_iter"][-1]
    assert torch.equal(s.d_value_function.view(torch.int32), plain.d_value_function.view(torch.int32))
    plain._backend.close()
    s._backend.close()
    monkeypatch.delenv("PI_MI355_RESIDENT")
    s = envs.make("pendulum", 200, device=cuda_device)
    monkeypatch.delenv("PI_MI355_FLOW_TIMEOUT")
    s.d_value_function.zero_()
    s.d_new_value_function.zero_()
    delta = s.policy_evaluation()
    ref = envs.make("pendulum", 200, device=cuda_device)
    monkeypatch.setenv("PI_MI355_RESIDENT", "0")
    plain = envs.make("pendulum", 200, device=cuda_device)
    assert not plain._backend.resident
    want = plain.policy_evaluation()
    assert np.float32(delta) == np.float32(want) and s.stats["sweeps_per_iter"][-1] == plain.stats["sweeps_per_iter"][-1]
    assert torch.equal(s.d_value_function.view(torch.int32), plain.d_value_function.view(torch.int32))
    for x in (s, ref, plain):
        x._backend.close()
