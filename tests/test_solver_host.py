"""
Host-logic tests (no GPU): the solver's run loop, batching of sweeps between residual checks,
persistence and plugin surface, driven through the public API with the CPU checker installed
as the sweep backend (``tests.helpers.with_checker_backend`` — a private, test-only hook; the
product has one backend, the HIP one, and raises without a GPU).
"""
from __future__ import annotations

import numpy as np
import pytest

import oracle
from dynamicprogramming_amd import envs
from dynamicprogramming_amd.solver import CudaPIConfig, CudaPolicyIteration2D
from tests import helpers as H


def _solver(name, shape, config=None, **kw):
    cls = envs.ENVS[name]
    cfg = config or CudaPIConfig(**cls.CONFIG)
    return H.with_checker_backend(cls)(H.env_bins_space(name, shape), cls.ACTIONS, cfg, **kw)


def _oracle_run(name, shape, cfg):
    cls = envs.ENVS[name]
    bins = H.env_bins(name, shape)
    lo, hi, gshape, strides = oracle.grid_metadata(bins)
    states = oracle.states_from_bins(bins)
    term, tval = H.terminal_mask(name, states)
    return H.oracle_for(name).run(states, cls.ACTIONS, term, lo, hi, gshape, strides, gamma=cfg.gamma,
                                  theta=cfg.theta, max_eval_iter=cfg.max_eval_iter,
                                  max_pi_iter=cfg.max_pi_iter, terminal_value=tval)


@pytest.mark.parametrize("name,shape", [("mountain_car", (40, 30)), ("pendulum", (31, 29)),
                                         ("cartpole", (7, 6, 9, 6))])
def test_run_loop_equals_oracle_run(name, shape):
    """run() = the reference's loop (:300-370): same sweep counts per outer iteration, same V,
    same policy as oracle_run (which restates that loop independently in C++)."""
    cfg = CudaPIConfig(**{**envs.ENVS[name].CONFIG, "max_pi_iter": 12, "max_eval_iter": 600})
    s = _solver(name, shape, cfg)
    s.run()
    ref = _oracle_run(name, shape, cfg)
    assert s.stats["sweeps_per_iter"] == list(ref["sweeps_per_iter"])
    assert s.stats["pi_iterations"] == ref["outer_iterations"]
    assert np.array_equal(s.policy, ref["policy"])
    H.assert_bits_equal(s.value_function, ref["value_function"], "V")
    assert s.policy.dtype == np.int32 and s.value_function.dtype == np.float32
    assert not hasattr(s, "d_value_function")          # device arrays dropped (:379-385)


def test_run_in_one_launch_keeps_the_books_of_the_round_by_round_loop():
    """run() on a grid whose backend offers the whole run in one launch (pi_policy_iteration): the solver hands over
    V, the policy and the limits, and fills stats / results from the rounds the call reports exactly as the loop
    would have; a call that reports failure (None: V and policy untouched) is followed by the round-by-round loop; a
    subclass with its own policy_evaluation is never handed to it.  The one-launch call is played by a twin solver
    driven round by round (host logic only: the real kernels are compared on the GPU)."""
    name, shape = "mountain_car", (40, 30)
    cfg = CudaPIConfig(**{**envs.ENVS[name].CONFIG, "max_pi_iter": 12, "max_eval_iter": 600})
    ref = _oracle_run(name, shape, cfg)

    def install(solver, fail=False):
        seen = []

        def policy_iteration(V, policy, term, gamma, theta, max_eval, interval, max_pi):
            seen.append((float(gamma), float(theta), max_eval, interval, max_pi))
            if fail:
                return None
            twin = _solver(name, shape, cfg)
            log, stable = [], False
            for _ in range(max_pi):
                delta = twin.policy_evaluation()
                stable = twin.policy_improvement()
                log.append((twin.stats["sweeps_per_iter"][-1], delta, twin.stats["last_changed"]))
                if stable:
                    break
            V.copy_(twin.d_value_function)
            policy.copy_(twin.d_policy)
            return len(log), stable, log
        solver._backend.whole_run = True
        solver._backend.policy_iteration = policy_iteration
        return seen

    s = _solver(name, shape, cfg)
    seen = install(s)
    s.run()
    assert seen == [(float(np.float32(cfg.gamma)), float(cfg.theta), 600, 25, 12)]
    assert s._backend.calls == {"eval": 0, "improve": 0}            # nothing went round by round
    assert s.stats["sweeps_per_iter"] == list(ref["sweeps_per_iter"]) and s.stats["pi_iterations"] == ref["outer_iterations"]
    assert s.stats["eval_sweeps"] == sum(ref["sweeps_per_iter"]) and s.stats["improve_sweeps"] == ref["outer_iterations"]
    assert s.stats["stable"] is True and s.stats["last_changed"] == 0
    assert np.array_equal(s.policy, ref["policy"])
    H.assert_bits_equal(s.value_function, ref["value_function"], "V")
    assert not hasattr(s, "d_value_function")
    # the call could not be placed: the loop runs round by round, same results
    f = _solver(name, shape, cfg)
    seen = install(f, fail=True)
    f.run()
    assert len(seen) == 1 and f._backend.calls["improve"] == ref["outer_iterations"]
    assert f.stats["sweeps_per_iter"] == list(ref["sweeps_per_iter"]) and np.array_equal(f.policy, ref["policy"])
    H.assert_bits_equal(f.value_function, ref["value_function"], "V after the fallback")
    # a plugin with its own evaluation step is called round by round
    base = H.with_checker_backend(envs.ENVS[name])
    rounds = []

    class Own(base):
        def policy_evaluation(self):
            rounds.append(len(rounds))
            return super().policy_evaluation()

    o = Own(H.env_bins_space(name, shape), envs.ENVS[name].ACTIONS, cfg)
    seen = install(o)
    o.run()
    assert seen == [] and len(rounds) == ref["outer_iterations"] and np.array_equal(o.policy, ref["policy"])


def test_sweeps_between_checks_follow_the_25_rule():
    """Residual looked at on sweeps 0, 25, 50, ... and on the last allowed sweep (:325)."""
    cfg = CudaPIConfig(gamma=0.99, theta=0.0, max_eval_iter=60, max_pi_iter=1)
    s = _solver("pendulum", (12, 12), cfg)
    calls = []
    orig = s._evaluation_sweeps
    s._evaluation_sweeps = lambda n, g: (calls.append(n), orig(n, g))[1]
    s.policy_evaluation()
    assert calls == [1, 25, 25, 9]                      # i = 0 | 1..25 | 26..50 | 51..59
    assert s.stats["eval_sweeps"] == 60
    cfg2 = CudaPIConfig(gamma=0.5, theta=1e-3, max_eval_iter=1000, max_pi_iter=1)
    s2 = _solver("pendulum", (12, 12), cfg2)
    s2.policy_evaluation()
    assert s2.stats["eval_sweeps"] in (1, 26, 51)


def test_plugin_surface_and_metadata():
    s = _solver("cartpole_swingup", (5, 4, 6, 3))
    assert s.n_states == 5 * 4 * 6 * 3 and s.n_actions == 5
    assert s.grid_shape.tolist() == [5, 4, 6, 3] and s.grid_shape.dtype == np.int32
    assert s.strides.tolist() == [72, 18, 3, 1] and s.strides.dtype == np.int32
    assert s.corner_bits.shape == (16, 4) and s.corner_bits[1].tolist() == [0, 0, 0, 1]
    assert s.bounds_low.dtype == np.float32 and np.isclose(s.bounds_high[2], np.float32(np.pi))
    st = s.states_space
    assert st.shape == (s.n_states, 4) and st.dtype == np.float32
    assert np.array_equal(st[1], [st[0, 0], st[0, 1], st[0, 2], s._bins[3][1]])   # last dim fastest
    mask = (st[:, 0] < -2.4) | (st[:, 0] > 2.4)
    assert np.array_equal(s.d_terminal_mask[: s.n_states].numpy().astype(bool), mask)
    with pytest.raises(AssertionError, match="exactly 4"):
        H.with_checker_backend(envs.CartPoleSwingUpCuda)({"a": [0, 1], "b": [0, 1]}, [0.0])
    with pytest.raises(ValueError, match="repeated"):
        H.with_checker_backend(envs.PendulumCuda)({"a": [0.0, 0.0, 1.0], "b": [0.0, 1.0]}, [0.0])
    with pytest.raises(TypeError):                      # _dynamics_cuda_src is abstract (:113)
        H.with_checker_backend(CudaPolicyIteration2D)({"a": [0, 1], "b": [0, 1]}, [0.0])


def test_terminal_value_and_goal_seeding():
    s = _solver("overhead_crane", (9, 7, 9, 7))
    n = s.n_states
    goal = s._goal_mask
    assert goal.any()
    v = s.d_value_function[:n].numpy()
    assert np.allclose(v[goal], 1.0 / (1.0 - s.config.gamma))
    assert np.array_equal(s.d_new_value_function[:n].numpy(), v)
    assert np.all(v[~goal] == 0.0)
    s.run()
    assert np.allclose(s.value_function[goal], 1.0 / (1.0 - s.config.gamma))   # terminal: kept


def test_save_load_schema(tmp_path):
    cfg = CudaPIConfig(gamma=0.9, theta=1e-3, max_eval_iter=100, max_pi_iter=3)
    s = _solver("mountain_car", (15, 11), cfg)
    s.run()
    s.save(tmp_path / "sub" / "policy.anything")
    path = tmp_path / "sub" / "policy.npz"
    assert path.exists()
    data = np.load(path)
    assert sorted(data.files) == sorted(["value_function", "policy", "bounds_low", "bounds_high",
                                         "grid_shape", "strides", "corner_bits", "action_space",
                                         "states_space"])
    assert data["policy"].dtype == np.int32 and data["states_space"].shape == (165, 2)
    loaded = envs.MountainCarCuda.load(path)
    assert np.array_equal(loaded.policy, s.policy) and loaded.n_states == 165 and loaded.n_actions == 3
    assert isinstance(loaded.config, CudaPIConfig)
    crane = _solver("overhead_crane", (5, 4, 5, 4), cfg, target_x=0.5)
    crane.run()
    crane.save(tmp_path / "crane")
    assert envs.OverheadCraneCuda.load(tmp_path / "crane").target_x == pytest.approx(0.5)


def test_reference_import_path():
    import src.cuda_policy_iteration as m
    assert m.CudaPolicyIteration4D is envs.CudaPolicyIteration4D and hasattr(m, "GPU_AVAILABLE")
    ref_fields = dict(gamma=0.99, theta=1e-4, max_eval_iter=10_000, max_pi_iter=50, log_interval=100)
    cfg = m.CudaPIConfig()
    assert {k: getattr(cfg, k) for k in ref_fields} == ref_fields          # reference :36-43
    assert list(cfg.__dict__)[:5] == list(ref_fields)                      # same positional order


def test_value_iteration_and_checkpoint(tmp_path):
    """Fused value-iteration sweeps reach the same fixed point as policy iteration (same greedy
    policy away from ties, V within the residual bound), and a mid-run checkpoint resumes."""
    name, shape = "mountain_car", (30, 20)
    cfg = CudaPIConfig(gamma=0.95, theta=1e-5, max_eval_iter=2000, max_pi_iter=30)
    pi = _solver(name, shape, cfg)
    pi.run()
    vi = _solver(name, shape, cfg)
    delta = vi.value_iteration()
    assert delta < cfg.theta and vi.stats["value_sweeps"] % 25 == 1
    v = vi.d_value_function[: vi.n_states].numpy()
    assert np.max(np.abs(v - pi.value_function)) < 1e-3
    assert np.mean(vi.d_policy[: vi.n_states].numpy() == pi.policy) > 0.98
    # checkpoint / resume: 1 PI iteration, snapshot, finish in a new solver
    a = _solver(name, shape, cfg)
    a.policy_evaluation()
    a.policy_improvement()
    a.save_checkpoint(tmp_path / "ck")
    b = _solver(name, shape, cfg)
    b.load_checkpoint(tmp_path / "ck")
    assert b.stats["eval_sweeps"] == a.stats["eval_sweeps"]
    a.run()
    b.run()
    assert np.array_equal(a.policy, b.policy)
    H.assert_bits_equal(a.value_function, b.value_function, "resumed run")
    with pytest.raises(ValueError, match="different grid"):
        _solver(name, (20, 20), cfg).load_checkpoint(tmp_path / "ck")


# ── runner command line (reference README.md:323-347) ───────────────────────
REFERENCE_FLAGS = ["--render", "--record", "--random", "--episodes", "--steps", "--bins", "--seed",
                   "--no-plot", "--retrain", "--save-path"]


@pytest.mark.parametrize("name", H.ENV_NAMES)
def test_runner_cli_accepts_the_reference_flag_set(name):
    """Every flag of the reference's shared CLI parses with the reference's types and defaults;
    the crane adds --start-x / --target-x (overhead_crane_cuda.py:588-591)."""
    from runners import _cli
    p = _cli.build_parser(name, f"results/{name}_cuda_policy.npz")
    flags = {s for a in p._actions for s in a.option_strings}
    assert set(REFERENCE_FLAGS) <= flags
    d = p.parse_args([])
    assert (d.render, d.random, d.record, d.episodes, d.steps, d.seed, d.no_plot, d.retrain) == (
        False, None, None, 5, 1000, 42, False, False)
    assert d.bins == envs.ENVS[name].DEFAULT_BINS and str(d.save_path) == f"results/{name}_cuda_policy.npz"
    a = p.parse_args(["--render", "--record", "out.gif", "--random", "--episodes", "3", "--steps", "10",
                      "--bins", "12", "--seed", "7", "--no-plot", "--retrain", "--save-path", "x/y.npz"])
    assert a.random == 5 and a.bins == 12 and a.retrain and str(a.record) == "out.gif"
    assert p.parse_args(["--random", "9"]).random == 9
    # the one extension flag: off by default, so a reference command line trains with policy iteration
    assert d.value_iteration is False and p.parse_args(["--value-iteration"]).value_iteration is True
    if name == "overhead_crane":
        c = p.parse_args(["--target-x", "0.0", "--start-x", "1.5"])
        assert (c.target_x, c.start_x) == (0.0, 1.5) and p.parse_args([]).target_x == -2.5
    else:
        assert "--target-x" not in flags


def test_runner_modules_keep_the_reference_module_surface():
    import importlib
    for name, cls in envs.ENVS.items():
        mod = importlib.import_module(f"runners.{name}_cuda")
        assert getattr(mod, cls.__name__) is cls
        assert mod.BINS_PER_DIM == cls.DEFAULT_BINS and list(mod.BINS_SPACE) == list(cls.bins_space(2))
        assert np.array_equal(mod.ACTION_SPACE, cls.ACTIONS) and callable(mod.train)
    # the crane's train() solves the -2.5 m task by default, as the reference's train(save_path, target_x=-2.5)
    import inspect
    crane = importlib.import_module("runners.overhead_crane_cuda")
    assert inspect.signature(crane.train).parameters["target_x"].default == -2.5


def test_runner_loads_an_existing_archive_without_a_gpu(tmp_path, capsys):
    """No --retrain and the archive exists -> load() (works without GPU or library), exactly the
    reference's control flow; the archive here has the reference's exact key set and dtypes
    (runners/results/mountain_car_cuda_policy.npz), rebuilt around the reference-produced V / policy
    of tests/golden/reference_results.npz, and is then read back the way
    runners/hybrid_double_cartpole.py:35-43 reads it (raw keys) and fed to utils.get_optimal_action."""
    from itertools import product
    from runners import _cli
    from utils import get_optimal_action
    g = H.golden("reference_results")
    shape = tuple(int(x) for x in g["mountain_car_grid_shape"])
    bins = H.env_bins("mountain_car", shape)
    lo, hi, gshape, strides = oracle.grid_metadata(bins)
    path = tmp_path / "mc_policy.npz"
    np.savez(path, value_function=g["mountain_car_value_function"], policy=g["mountain_car_policy"],
             bounds_low=lo, bounds_high=hi, grid_shape=gshape, strides=strides,
             corner_bits=np.array(list(product([0, 1], repeat=2)), dtype=np.int32),
             action_space=g["mountain_car_action_space"], states_space=oracle.states_from_bins(bins))
    pi = _cli.main("mountain_car", "unused.npz", ["--save-path", str(path), "--episodes", "2", "--no-plot"])
    assert "accepted and ignored" in capsys.readouterr().out
    assert type(pi).__name__ == "MountainCarCuda" and pi.n_states == 40000 and pi.n_actions == 3
    for key in ("value_function", "policy", "bounds_low", "bounds_high", "grid_shape", "strides",
                "corner_bits", "action_space", "states_space"):
        assert np.array_equal(getattr(pi, key), np.load(path)[key]), key
    d = np.load(path)                                   # the hybrid runner's raw access
    a = float(get_optimal_action(np.array([-0.5, 0.0], np.float32), d["policy"], d["action_space"],
                                 d["bounds_low"], d["bounds_high"], d["grid_shape"], d["strides"],
                                 d["corner_bits"]))
    assert d["action_space"].min() <= a <= d["action_space"].max()
    assert _cli.main("mountain_car", "unused.npz", ["--random"]) is None      # no training, like the reference


# ── memory order of the dimensions (solver.MEMORY_ORDER / PI_MI355_ORDER) ───────────────────────────────
@pytest.mark.parametrize("name,shape,order", [("cartpole_swingup", (7, 5, 9, 6), (0, 2, 1, 3)),
                                               ("double_pendulum_swingup", (6, 5, 7, 4), (3, 1, 0, 2)),
                                               ("pendulum", (17, 13), (1, 0)),
                                               ("double_cartpole", (3, 4, 3, 4, 3, 5), (0, 1, 2, 3, 5, 4))])
def test_memory_order_moves_states_not_results(name, shape, order, tmp_path, monkeypatch):
    """With a memory order the device tensors hold the grid transposed; everything the host sees — value_function,
    policy, checkpoints, archives — stays in the user's order and every number stays the same: run() with and without
    the order agree bit for bit (V, policy, sweep counts), the mask on the "device" is the transposed mask, a mid-run
    checkpoint written with the order is read back without it (and vice versa)."""
    cfg = CudaPIConfig(**{**envs.ENVS[name].CONFIG, "max_pi_iter": 3, "max_eval_iter": 80})
    plain = _solver(name, shape, cfg)
    cls = type(plain)
    ordered_cls = type(cls.__name__ + "Ordered", (cls,), {"MEMORY_ORDER": order, "_ORDER_MIN_STATES": 0})
    s = ordered_cls(H.env_bins_space(name, shape), envs.ENVS[name].ACTIONS, cfg)
    assert s._order == order and plain._order is None
    n = s.n_states
    mask_user, _ = H.terminal_mask(name, s.states_space)
    got = s.d_terminal_mask.numpy()[:n].astype(bool)
    want = np.ascontiguousarray(mask_user.reshape(shape).transpose(order)).reshape(-1)
    assert np.array_equal(got, want)
    assert np.array_equal(s._to_user(got), mask_user) and np.array_equal(s._to_memory(mask_user), got)
    # two evaluations + an improvement, then checkpoints cross over
    for sol in (s, plain):
        sol.policy_evaluation()
        sol.policy_improvement()
    s.save_checkpoint(tmp_path / "ordered")
    plain.save_checkpoint(tmp_path / "plain")
    a, b = np.load(tmp_path / "ordered.npz"), np.load(tmp_path / "plain.npz")
    H.assert_bits_equal(a["value_function"], b["value_function"], "checkpoint V (user order)")
    assert np.array_equal(a["policy"], b["policy"])
    s2 = ordered_cls(H.env_bins_space(name, shape), envs.ENVS[name].ACTIONS, cfg)
    s2.load_checkpoint(tmp_path / "plain")
    p2 = _solver(name, shape, cfg)
    p2.load_checkpoint(tmp_path / "ordered")
    for sol in (s2, p2):
        sol.run()
    plain_full = _solver(name, shape, cfg)
    plain_full.policy_evaluation()
    plain_full.policy_improvement()
    plain_full.run()
    for sol in (s2, p2):
        H.assert_bits_equal(sol.value_function, plain_full.value_function, "V after resume")
        assert np.array_equal(sol.policy, plain_full.policy)
    # PI_MI355_ORDER overrides the class: "user" switches it off, a list forces one
    monkeypatch.setenv("PI_MI355_ORDER", "user")
    assert ordered_cls(H.env_bins_space(name, shape), envs.ENVS[name].ACTIONS, cfg)._order is None
    monkeypatch.setenv("PI_MI355_ORDER", ",".join(map(str, order)))
    assert _solver(name, shape, cfg)._order == order
    monkeypatch.setenv("PI_MI355_ORDER", "0,0,1,2")
    with pytest.raises(ValueError, match="permutation"):
        _solver(name, shape, cfg)


def test_memory_order_applies_to_big_grids_only():
    """The class default is taken from a threshold on: small grids (all of this suite's) stay in the user's order."""
    cls = envs.ENVS["double_pendulum_swingup"]
    assert cls.MEMORY_ORDER is not None and sorted(cls.MEMORY_ORDER) == [0, 1, 2, 3]
    assert _solver("double_pendulum_swingup", (6, 5, 7, 4))._order is None
    for name, c in envs.ENVS.items():
        if c.MEMORY_ORDER is not None and c.MEMORY_ORDER != "user":      # "user": measured, the env's own order stays
            assert sorted(c.MEMORY_ORDER) == list(range(c._D)), name


def test_sharded_solvers_keep_the_envs_order_unless_forced(monkeypatch):
    """The fast single-GPU orders widen a shard's halo (a band of planes instead of a triangle of rows), so a rank of a
    sharded run stays in the env's order; PI_MI355_ORDER still forces one."""
    cls = envs.ENVS["double_pendulum_swingup"]
    s = object.__new__(cls)
    s.n_states, s._process_group = 1 << 25, None

    class TwoRanks:
        world, rank = 2, 0
    s._transport_arg = None
    assert s._choose_memory_order() == tuple(cls.MEMORY_ORDER)          # single rank, big grid
    s._transport_arg = TwoRanks()
    assert s._will_shard() and s._choose_memory_order() is None
    s._transport_arg = False
    assert not s._will_shard() and s._choose_memory_order() == tuple(cls.MEMORY_ORDER)
    s._transport_arg = TwoRanks()
    monkeypatch.setenv("PI_MI355_ORDER", "0,2,1,3")
    assert s._choose_memory_order() == (0, 2, 1, 3)


def test_untuned_plugins_get_their_order_measured_and_tuned_classes_do_not(monkeypatch):
    """Round 6: which solvers have their memory order MEASURED at construction (solver._tune_memory_order) — big
    single-rank grids of a class without a MEMORY_ORDER of its own that runs the stock allocation; never a sharded solver, a
    class with a tuple or "user", a small grid, or a subclass that replaces _allocate_tensors_and_compile."""
    from dynamicprogramming_amd.solver import CudaPolicyIteration4D
    calls = []

    class Plugin(CudaPolicyIteration4D):                        # what a user of the reference writes
        def _dynamics_cuda_src(self):
            return ""

        def _tune_memory_order(self):
            calls.append(type(self).__name__)
            return (0, 2, 1, 3)

    class OwnAllocation(Plugin):
        def _allocate_tensors_and_compile(self):
            pass

    class Tuned(Plugin):
        MEMORY_ORDER = (1, 0, 2, 3)

    class Measured(Plugin):
        MEMORY_ORDER = "user"

    def probe(cls, n_states, transport=None):
        s = object.__new__(cls)
        s.n_states, s._process_group, s._transport_arg = n_states, None, transport
        return s._choose_memory_order()

    class TwoRanks:
        world, rank = 2, 0

    assert probe(Plugin, 1 << 23) == (0, 2, 1, 3) and calls == ["Plugin"]
    assert probe(Plugin, 1 << 20) is None                       # small grid: the env's order, nothing measured
    assert probe(Plugin, 1 << 23, TwoRanks()) is None           # sharded: the env's order (halo rows stay contiguous)
    assert probe(OwnAllocation, 1 << 23) is None                # builds its own device arrays: left alone
    assert probe(Tuned, 1 << 23) == (1, 0, 2, 3) and probe(Measured, 1 << 23) is None
    assert calls == ["Plugin"]
    monkeypatch.setenv("PI_MI355_ORDER", "user")
    assert probe(Plugin, 1 << 23) is None and calls == ["Plugin"]
    monkeypatch.setenv("PI_MI355_ORDER", "auto")                # forces the measurement on any class and size
    assert probe(Tuned, 1 << 10) == (0, 2, 1, 3) and calls == ["Plugin", "Tuned"]


def test_candidate_orders_come_from_the_spread_of_a_waves_successors():
    """The candidates `_tune_memory_order` times, from spreads as `_lane_spreads` measures them on the device (here: made
    up after the double cartpole — x' = x + dt x_dot, so a wave along x_dot spreads over x, a wave along x does not
    spread at all; angles and angular speeds spread over each other): the env's order first, the two tightest lane
    dimensions each last, each with its partner second-fastest and every other dimension tried as the slowest."""
    s = object.__new__(envs.ENVS["double_cartpole"])
    big = {0: 0.0, 1: 0.3, 2: 2.0, 3: 2.0, 4: 2.0, 5: 2.0}     # the poles do not move the cart's position within one step
    spread = {0: {k: 0.0 for k in range(6)}, 1: {0: 0.9, 1: 0.0, 2: 0.0, 3: 0.05, 4: 0.0, 5: 0.05},
              2: dict(big), 3: dict(big), 4: dict(big), 5: dict(big)}
    c = s._candidate_orders(spread)
    assert c[0] == (0, 1, 2, 3, 4, 5) and len(c) == len(set(c)) == 11 and all(sorted(o) == list(range(6)) for o in c)
    assert (1, 2, 3, 4, 5, 0) in c and (0, 2, 3, 4, 5, 1) in c                   # lanes along x, along x_dot, rest in order
    for slow in (2, 3, 4, 5):                                                     # partner second-fastest, every other dim slowest
        rest = [k for k in (2, 3, 4, 5) if k != slow]
        assert tuple([slow] + rest + [0, 1]) in c and tuple([slow] + rest + [1, 0]) in c
    two = object.__new__(envs.ENVS["pendulum"])
    assert two._candidate_orders({0: {0: 0.0, 1: 3.0}, 1: {0: 0.5, 1: 0.0}}) == [(0, 1), (1, 0)]


# ── property tests (hypothesis): the index algebra of the memory order and of the exchange planner ───────
from hypothesis import given, settings, strategies as st  # noqa: E402


@st.composite
def _grids(draw):
    D = draw(st.sampled_from([2, 4, 6]))
    shape = [draw(st.integers(2, 5 if D == 6 else 7)) for _ in range(D)]
    order = draw(st.permutations(list(range(D))))
    return D, shape, tuple(order)


@settings(max_examples=25, deadline=None)
@given(_grids())
def test_memory_order_index_algebra(grid):
    """to_memory / to_user are inverse permutations of the flat index, and memory index m holds the state whose
    coordinates along the engine's memory dimensions are the row-major digits of m over the permuted shape."""
    from dynamicprogramming_amd import _native
    D, shape, order = grid
    bins = [np.linspace(-1.0 - d, 1.0 + d, g, dtype=np.float32) for d, g in enumerate(shape)]
    eng = _native.Engine(D, shape, [b.min() for b in bins], [b.max() for b in bins], bins, [0.0, 1.0], device=-1, order=order)
    try:
        n = int(np.prod(shape))
        user = np.arange(n, dtype=np.int64)
        mem = np.asarray(eng.to_memory(user))
        assert sorted(mem.tolist()) == list(range(n))
        assert np.array_equal(np.asarray(eng.to_user(mem)), user)
        digits_user = np.stack(np.unravel_index(mem, shape), axis=1)                 # user coordinates of memory slot m
        mshape = [shape[d] for d in eng.order]
        digits_mem = np.stack(np.unravel_index(np.arange(n), mshape), axis=1)        # memory coordinates of slot m
        for k, d in enumerate(eng.order):
            assert np.array_equal(digits_mem[:, k], digits_user[:, d])
        assert eng.info(18) == sum(d << (3 * k) for k, d in enumerate(eng.order))
    finally:
        eng.close()


@settings(max_examples=40, deadline=None)
@given(st.integers(1, 5), st.integers(3, 24), st.integers(1, 9), st.data())
def test_exchange_planner_sends_exactly_what_is_reachable(world, g0, stride0, data):
    """pi_plan_segments (host-only): every (dst, unit) a rank can reach and does not own arrives exactly once, from its
    owner; nothing else travels; no rank sends to itself."""
    from dynamicprogramming_amd import _native
    n = g0 * stride0 - data.draw(st.integers(0, stride0 - 1))                       # a ragged last unit
    per = -(-n // world)
    reach = np.array(data.draw(st.lists(st.lists(st.booleans(), min_size=g0, max_size=g0), min_size=world, max_size=world)))
    segs = _native.plan_segments(world, g0, stride0, n, per, reach)
    got = np.zeros((world, n), dtype=np.int32)
    for src, dst, a, b in segs:
        assert src != dst and 0 <= a < b <= n
        assert src * per <= a and b <= min((src + 1) * per, n)                      # the sender owns what it sends
        got[dst, a:b] += 1
    for dst in range(world):
        need = np.zeros(n, dtype=bool)
        for p in np.flatnonzero(reach[dst]):
            need[p * stride0:min((p + 1) * stride0, n)] = True
        need[dst * per:min((dst + 1) * per, n)] = False                             # its own shard does not travel
        assert np.array_equal(got[dst] > 0, need) and got[dst].max(initial=0) <= 1


AXES_ENVS = [("mountain_car", (40, 30)), ("continuous_mountain_car", (33, 21)), ("cartpole", (9, 5, 11, 4)),
             ("cartpole_swingup", (9, 4, 6, 3)), ("double_cartpole", (9, 3, 7, 2, 7, 3)),
             ("double_cartpole_swingup", (9, 2, 3, 2, 3, 2))]


@pytest.mark.parametrize("name,shape", AXES_ENVS)
@pytest.mark.parametrize("order", [None, "reversed"])
def test_terminal_mask_from_bin_tables_equals_the_reference_hook(name, shape, order, monkeypatch):
    """`_terminal_fn_axes` (the mask from the bin tables, built on the device: no (n, D) array) gives exactly the mask
    of the reference's `_terminal_fn(states_space)` hook (:127-138) — also when the device arrays live in another
    memory order — and the solver does not touch `states_space` for it."""
    cls = envs.ENVS[name]
    D = cls._D
    if order == "reversed":
        monkeypatch.setenv("PI_MI355_ORDER", ",".join(str(d) for d in reversed(range(D))))
    s = _solver(name, shape)
    assert s._states_space is None                                      # never materialised
    want, value = cls._terminal_fn(s, s.states_space)
    assert want.any() and not want.all()
    got = s._to_user(s.d_terminal_mask[: s.n_states].numpy())
    assert np.array_equal(got.astype(bool), np.asarray(want, bool))
    assert s._mask_arg() is s.d_terminal_mask
    tv = s._to_user(s.d_value_function[: s.n_states].numpy())
    assert np.all(tv[want] == np.float32(value)) and np.all(tv[~want] == 0.0)


def test_a_subclass_that_overrides_the_reference_hook_gets_its_own_mask():
    """A plugin written for the reference overrides `_terminal_fn` only: the bin-table hook of the class it derives
    from must not shadow it; and a class without any terminal hook has no mask."""
    class Narrow(envs.CartPoleSwingUpCuda):
        def _terminal_fn(self, states):
            return np.abs(states[:, 0]) > 1.0, -3.0
    bins = H.env_bins_space("cartpole_swingup", (9, 4, 6, 3))
    s = H.with_checker_backend(Narrow)(bins, Narrow.ACTIONS, CudaPIConfig(**Narrow.CONFIG))
    want = np.abs(s.states_space[:, 0]) > 1.0
    assert np.array_equal(s.d_terminal_mask[: s.n_states].numpy().astype(bool), want)
    assert np.all(s.d_value_function[: s.n_states].numpy()[want] == np.float32(-3.0))
    p = _solver("pendulum", (12, 12))
    assert p._mask_arg() is None and not p.d_terminal_mask.any()
