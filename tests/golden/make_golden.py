"""
tests/golden/make_golden.py — generate the committed golden vectors.

Runs ONLY in the build container (needs /root/reference; the GPU box never runs it):

    python -m tests.golden.make_golden

For every env it builds the reference's own kernel text as a CPU checker
(oracle/build_ref.py, text read with ``ast``; nothing from the reference is imported) and

  1. checks that oracle/pi_oracle.cpp + this repo's env plugin string, built in libm mode,
     agree with the reference's text BIT FOR BIT on every vector below (assertion — the
     generator refuses to write goldens otherwise), then
  2. stores the reference-produced outputs as small ``.npz`` fixtures:
       <env>.npz          dynamics (seeded state/action pairs, incl. wrap/termination edges),
                          interpolation (incl. out-of-range, exact-node and last-cell points),
                          one evaluation + one improvement sweep from a seeded V / policy on
                          a cubic and a non-cubic tiny grid, and the top-2 action-value gap;
       pendulum_c1_run.npz  full run() on BASELINE config C1 (Pendulum 50x50, 11 actions);
       reference_results.npz  value_function + policy of the two result archives the
                          reference commits (runners/results/*.npz) — the only
                          reference-PRODUCED end-to-end numbers.
"""
from __future__ import annotations

import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))

import oracle  # noqa: E402
from oracle import build_ref  # noqa: E402
from dynamicprogramming_amd import envs  # noqa: E402
from tests.helpers import env_bins, sample_states, terminal_mask  # noqa: E402

OUT = Path(__file__).resolve().parent

TINY_GRIDS = {
    2: [(16, 16), (13, 9)],
    4: [(7, 7, 7, 7), (9, 7, 11, 5)],
    6: [(4, 4, 4, 4, 4, 4), (5, 4, 6, 4, 5, 4)],
}


def assert_same(a, b, what):
    a, b = np.asarray(a), np.asarray(b)
    if a.dtype.kind == "f":
        same = np.array_equal(a.view(np.uint32), b.view(np.uint32)) or np.array_equal(a, b, equal_nan=True)
    else:
        same = np.array_equal(a, b)
    if not same:
        bad = np.flatnonzero((a != b).ravel())
        raise AssertionError(f"{what}: port != reference at {len(bad)} of {a.size} entries, "
                             f"first {bad[:5]}: {a.ravel()[bad[:5]]} vs {b.ravel()[bad[:5]]}")


def make_env(name: str) -> dict:
    cls = envs.ENVS[name]
    D = cls._D
    ref = build_ref.load(name)
    port = oracle.build(D, envs.dynamics_source(name), libm=True)
    rng = np.random.default_rng(sum(map(ord, name)))
    actions = np.asarray(cls.ACTIONS, dtype=np.float32)
    gamma = np.float32(cls.CONFIG["gamma"])
    out: dict = {"actions": actions, "gamma": gamma, "D": np.int32(D)}

    # -- dynamics ------------------------------------------------------------------
    bins_ref = env_bins(name, TINY_GRIDS[D][0])
    st = sample_states(rng, bins_ref, 512)
    act = rng.choice(actions, size=len(st)).astype(np.float32)
    act[:32] = rng.uniform(actions.min() * 1.5, actions.max() * 1.5, size=32).astype(np.float32)
    r_next, r_rew, r_term = ref.step(st, act)
    p_next, p_rew, p_term = port.step(st, act)
    assert_same(p_next, r_next, f"{name} step next")
    assert_same(p_rew, r_rew, f"{name} step reward")
    assert_same(p_term, r_term, f"{name} step terminated")
    out.update(step_states=st, step_actions=act, step_next=r_next, step_reward=r_rew,
               step_term=r_term)

    # -- interpolation + sweeps on tiny grids ------------------------------------------
    for gi, shape in enumerate(TINY_GRIDS[D]):
        bins = env_bins(name, shape)
        lo, hi, gshape, strides = oracle.grid_metadata(bins)
        assert tuple(gshape) == tuple(shape)
        states = oracle.states_from_bins(bins)
        n = len(states)
        pts = sample_states(rng, bins, {2: 1024, 4: 512, 6: 160}[D])
        r_idx, r_w = ref.interp(pts, lo, hi, gshape, strides)
        p_idx, p_w = port.interp(pts, lo, hi, gshape, strides)
        assert_same(p_idx, r_idx, f"{name} interp idx {shape}")
        assert_same(p_w, r_w, f"{name} interp w {shape}")

        term, tval = terminal_mask(name, states)
        V = rng.standard_normal(n).astype(np.float32) * np.float32(3.0)
        V[term] = np.float32(tval)
        pol = rng.integers(0, len(actions), size=n).astype(np.int32)
        pol[term] = 0
        r_V, r_delta = ref.eval_sweep(states, actions, pol, V, term, lo, hi, gshape, strides, gamma)
        p_V, p_delta = port.eval_sweep(states, actions, pol, V, term, lo, hi, gshape, strides, gamma)
        assert_same(p_V, r_V, f"{name} eval V' {shape}")
        assert np.float32(p_delta) == np.float32(r_delta), (name, shape, p_delta, r_delta)
        r_pol, r_changed = ref.improve_sweep(states, actions, pol, V, term, lo, hi, gshape, strides,
                                             gamma)
        p_pol, p_changed, qb, qs = port.improve_sweep(states, actions, pol, V, term, lo, hi, gshape,
                                                      strides, gamma, want_q=True)
        assert_same(p_pol, r_pol, f"{name} improve policy {shape}")
        assert p_changed == r_changed
        tag = f"g{gi}"
        out.update({
            f"{tag}_shape": np.asarray(shape, dtype=np.int32),
            f"{tag}_pts": pts, f"{tag}_idx": r_idx, f"{tag}_w": r_w,
            f"{tag}_V": V, f"{tag}_policy": pol, f"{tag}_term": term,
            f"{tag}_V_next": r_V, f"{tag}_delta": np.float32(r_delta),
            f"{tag}_policy_next": r_pol, f"{tag}_changed": np.int64(r_changed),
            f"{tag}_q_gap": (qb - qs).astype(np.float32),
        })
    return out


def make_c1_run() -> dict:
    """BASELINE config C1: Pendulum 50x50, 11 torques, full run() on the reference's text."""
    name = "pendulum"
    cls = envs.ENVS[name]
    bins = env_bins(name, (50, 50))
    actions = np.linspace(-2.0, 2.0, 11, dtype=np.float32)
    lo, hi, gshape, strides = oracle.grid_metadata(bins)
    states = oracle.states_from_bins(bins)
    term = np.zeros(len(states), dtype=bool)
    cfg = cls.CONFIG
    ref = build_ref.load(name)
    port = oracle.build(2, envs.dynamics_source(name), libm=True)
    kw = dict(gamma=cfg["gamma"], theta=cfg["theta"], max_eval_iter=cfg["max_eval_iter"],
              max_pi_iter=cfg["max_pi_iter"])
    r = ref.run(states, actions, term, lo, hi, gshape, strides, **kw)
    p = port.run(states, actions, term, lo, hi, gshape, strides, **kw)
    assert_same(p["value_function"], r["value_function"], "C1 V")
    assert_same(p["policy"], r["policy"], "C1 policy")
    assert p["eval_sweeps"] == r["eval_sweeps"] and p["outer_iterations"] == r["outer_iterations"]
    print(f"C1 run: {r['outer_iterations']} outer iterations, {r['eval_sweeps']} eval sweeps, "
          f"stable={r['stable']}")
    return dict(bins0=bins[0], bins1=bins[1], actions=actions, value_function=r["value_function"],
                policy=r["policy"], outer_iterations=np.int64(r["outer_iterations"]),
                eval_sweeps=np.int64(r["eval_sweeps"]), stable=np.bool_(r["stable"]),
                sweeps_per_iter=r["sweeps_per_iter"], **{k: np.float64(v) if isinstance(v, float)
                                                         else np.int64(v) for k, v in kw.items()})


SMALL_RUNS = [("cartpole", (8, 8, 8, 8)), ("double_pendulum_swingup", (8, 8, 8, 8)),
              ("double_cartpole", (4, 4, 4, 4, 4, 4))]


def make_small_runs() -> dict:
    """Full run()s of 4-D and 6-D envs on small grids through the reference's own kernel text (env
    settings, at most 10 outer iterations and 5 000 sweeps per evaluation): iteration structure,
    V and policy that the restatement must reproduce bit for bit."""
    out = {}
    for name, shape in SMALL_RUNS:
        cls = envs.ENVS[name]
        bins = env_bins(name, shape)
        lo, hi, gshape, strides = oracle.grid_metadata(bins)
        states = oracle.states_from_bins(bins)
        term, tval = terminal_mask(name, states)
        cfg = cls.CONFIG
        kw = dict(gamma=cfg["gamma"], theta=cfg["theta"], max_eval_iter=min(cfg["max_eval_iter"], 5000),
                  max_pi_iter=min(cfg["max_pi_iter"], 10), terminal_value=tval)
        ref = build_ref.load(name)
        port = oracle.build(cls._D, envs.dynamics_source(name), libm=True)
        r = ref.run(states, cls.ACTIONS, term, lo, hi, gshape, strides, **kw)
        p = port.run(states, cls.ACTIONS, term, lo, hi, gshape, strides, **kw)
        assert_same(p["value_function"], r["value_function"], f"{name} run V")
        assert_same(p["policy"], r["policy"], f"{name} run policy")
        assert np.array_equal(p["sweeps_per_iter"], r["sweeps_per_iter"])
        print(f"{name} {shape}: {r['outer_iterations']} outer iterations, {r['eval_sweeps']} eval sweeps, "
              f"stable={r['stable']}, terminal states {int(term.sum())}")
        out.update({f"{name}_shape": np.asarray(shape, np.int32), f"{name}_value_function": r["value_function"],
                    f"{name}_policy": r["policy"], f"{name}_sweeps_per_iter": r["sweeps_per_iter"],
                    f"{name}_outer_iterations": np.int64(r["outer_iterations"]),
                    f"{name}_stable": np.bool_(r["stable"]),
                    f"{name}_max_eval_iter": np.int64(kw["max_eval_iter"]),
                    f"{name}_max_pi_iter": np.int64(kw["max_pi_iter"])})
    return out


def trim_reference_results() -> dict:
    out = {}
    for env in ("mountain_car", "continuous_mountain_car"):
        data = np.load(build_ref.REFERENCE / "runners" / "results" / f"{env}_cuda_policy.npz")
        out[f"{env}_value_function"] = data["value_function"].astype(np.float32)
        out[f"{env}_policy"] = data["policy"].astype(np.int32)
        out[f"{env}_grid_shape"] = data["grid_shape"].astype(np.int32)
        out[f"{env}_bounds_low"] = data["bounds_low"].astype(np.float32)
        out[f"{env}_bounds_high"] = data["bounds_high"].astype(np.float32)
        out[f"{env}_action_space"] = data["action_space"].astype(np.float32)
    return out


def main() -> None:
    if not build_ref.available():
        raise SystemExit("needs /root/reference (build container only)")
    for name in envs.ENVS:
        data = make_env(name)
        np.savez_compressed(OUT / f"{name}.npz", **data)
        print(f"{name:28s} ok  ({(OUT / f'{name}.npz').stat().st_size / 1024:.0f} KiB)")
    np.savez_compressed(OUT / "pendulum_c1_run.npz", **make_c1_run())
    np.savez_compressed(OUT / "small_runs.npz", **make_small_runs())
    np.savez_compressed(OUT / "reference_results.npz", **trim_reference_results())
    total = sum(p.stat().st_size for p in OUT.glob("*.npz"))
    print(f"total {total / 1024:.0f} KiB")


if __name__ == "__main__":
    main()
