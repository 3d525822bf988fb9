"""
tests/golden/make_step_python_golden.py — 4-D / 6-D env dynamics pinned to reference-EXECUTED code.

The reference ships, beside every custom env's CUDA `step_dynamics` string, a float64 numpy mirror
`_step_python(state, action)` "matching the CUDA dynamics" that its rollouts run
(/root/reference/runners/cartpole_swingup_cuda.py:138-169, double_pendulum_swingup_cuda.py:211-268,
double_cartpole_cuda.py:186-235, double_cartpole_swingup_cuda.py:250-325, overhead_crane_cuda.py:211-245).
This script RUNS those functions — the reference's own Python, in this build container — on seeded
(state, action) pairs that span the grid, the angle wraps and the termination thresholds, and stores
inputs and outputs as data (tests/golden/step_python.npz).  The reference cannot travel; only the
vectors are committed.

How the function is run: the runner module itself cannot be imported here (its first import is the
solver module, which needs loguru and cupy), so the runner file is parsed with `ast` and exactly two
kinds of top-level statements are executed in a namespace that holds numpy only: simple constant
assignments (`_M1, _M2 = ...`, `_TH_THRESH = ...`) and the `def _step_python`.  Nothing of the
reference is written anywhere.

    python -m tests.golden.make_step_python_golden

Per env the archive holds  <env>_states (m, D) f32, <env>_actions (m,) f32,
  <env>_next (m, D) f32      the function's next state (it returns float32),
  <env>_reward (m,) f64, <env>_term (m,) bool,
  <env>_margin (m,) f64      distance of the successor from the nearest comparison threshold the function
                             branches on (termination, the swing-up's bonus gate): tests skip the flag /
                             reward comparison where a float32 rounding could legitimately flip the branch.
Inputs are float32-representable and handed over as float64, so the mirror computes in float64 under
any numpy version (numpy >= 2 would otherwise keep float32 scalars in float32).
"""
from __future__ import annotations

import ast
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from dynamicprogramming_amd import envs  # noqa: E402
from tests.helpers import env_bins, sample_states  # noqa: E402

REF = Path("/root/reference/runners")
OUT = Path(__file__).resolve().parent / "step_python.npz"
M = 2000

# env -> (runner file, wrapped angle dimensions, termination-edge spec used for sampling)
RUNNERS = {
    "cartpole_swingup": ("cartpole_swingup_cuda.py", (2,)),
    "double_pendulum_swingup": ("double_pendulum_swingup_cuda.py", (0, 2)),
    "overhead_crane": ("overhead_crane_cuda.py", ()),
    "double_cartpole": ("double_cartpole_cuda.py", ()),
    "double_cartpole_swingup": ("double_cartpole_swingup_cuda.py", (2, 4)),
}


def load_step_python(path: Path):
    """(_step_python, namespace) from the runner's text: constants + the one function, numpy only."""
    tree = ast.parse(path.read_text())
    ns = {"np": np, "__name__": "reference_step_python"}
    found = False
    for node in tree.body:
        keep = False
        if isinstance(node, ast.Assign):
            keep = all(isinstance(t, (ast.Name, ast.Tuple)) for t in node.targets)
        elif isinstance(node, ast.FunctionDef) and node.name == "_step_python":
            keep = found = True
        if not keep:
            continue
        mod = ast.Module(body=[node], type_ignores=[])
        try:
            exec(compile(mod, str(path), "exec"), ns)
        except Exception:                       # a constant built from something outside numpy: not needed
            if isinstance(node, ast.FunctionDef):
                raise
    if not found:
        raise LookupError(f"_step_python not found in {path}")
    return ns["_step_python"], ns


def thresholds(env: str, ns: dict):
    """Comparisons the reference function branches on, as (function of the float32 successor) -> distances."""
    if env == "cartpole_swingup":
        return lambda s: [abs(abs(s[0]) - 2.4)]
    if env == "double_pendulum_swingup":
        return lambda s: [np.inf]
    if env == "double_cartpole":
        th = float(ns["_TH_THRESH"])
        return lambda s: [abs(abs(s[0]) - 2.4), abs(abs(s[2]) - th), abs(abs(s[4]) - th)]
    if env == "double_cartpole_swingup":
        return lambda s: [abs(abs(s[0]) - 2.4), abs(np.cos(s[2]) - 0.7), abs(np.cos(s[4]) - 0.7)]
    if env == "overhead_crane":
        xm = float(ns["_X_MAX"])
        return lambda s: [abs(abs(s[0]) - xm), abs(abs(s[0]) - 0.20), abs(abs(s[2]) - 0.10), abs(abs(s[1]) - 0.20)]
    raise KeyError(env)


def sample(env: str, ns: dict, rng) -> tuple[np.ndarray, np.ndarray]:
    cls = envs.ENVS[env]
    D = cls._D
    bins = env_bins(env, (cls.DEFAULT_BINS,) * D)
    st = sample_states(rng, bins, M).astype(np.float64)
    lo = np.array([b.min() for b in bins], dtype=np.float64)
    hi = np.array([b.max() for b in bins], dtype=np.float64)
    actions = np.asarray(cls.ACTIONS, dtype=np.float32)
    act = rng.choice(actions, size=M).astype(np.float32)
    act[:64] = rng.uniform(actions.min() * 1.25, actions.max() * 1.25, size=64).astype(np.float32)
    k = M // 10
    _, wraps = RUNNERS[env]
    # angle wraps: theta close to +-pi with a velocity that carries it across (and one that does not)
    for j, d in enumerate(wraps):
        sl = slice(M - (j + 1) * k, M - j * k)
        sign = rng.choice([-1.0, 1.0], size=k)
        st[sl, d] = sign * (np.pi - rng.uniform(0.0, 0.05, size=k))
        st[sl, d + 1] = sign * rng.uniform(-0.3, 1.0, size=k) * hi[d + 1]
    # termination edges: the cart close to the rail's end, moving either way
    sl = slice(M - 4 * k, M - 3 * k)
    x_end = {"overhead_crane": float(ns.get("_X_MAX", 3.0))}.get(env, 2.4)
    if env != "double_pendulum_swingup":
        sign = rng.choice([-1.0, 1.0], size=k)
        st[sl, 0] = sign * (x_end - rng.uniform(-0.02, 0.1, size=k))
        st[sl, 1] = sign * rng.uniform(-0.2, 1.0, size=k) * hi[1]
    if env == "double_cartpole":                                  # pole angles close to the fall threshold
        th = float(ns["_TH_THRESH"])
        sl = slice(M - 5 * k, M - 4 * k)
        for d in (2, 4):
            sign = rng.choice([-1.0, 1.0], size=k)
            st[sl, d] = sign * (th - rng.uniform(-0.01, 0.05, size=k))
            st[sl, d + 1] = sign * rng.uniform(-0.2, 1.0, size=k) * hi[d + 1]
    if env == "overhead_crane":                                   # around the goal box
        sl = slice(M - 5 * k, M - 4 * k)
        st[sl, 0] = rng.uniform(-0.3, 0.3, size=k)
        st[sl, 1] = rng.uniform(-0.3, 0.3, size=k)
        st[sl, 2] = rng.uniform(-0.15, 0.15, size=k)
        st[sl, 3] = rng.uniform(-0.5, 0.5, size=k)
    return st.astype(np.float32), act


def main() -> None:
    if not REF.exists():
        raise SystemExit("/root/reference is not present: fixtures can only be generated in the build container")
    out = {"numpy_version": np.array(np.__version__)}
    for env, (fname, _) in RUNNERS.items():
        fn, ns = load_step_python(REF / fname)
        rng = np.random.default_rng(4000 + sum(map(ord, env)))
        st, act = sample(env, ns, rng)
        dist = thresholds(env, ns)
        D = st.shape[1]
        nxt = np.empty((M, D), dtype=np.float32)
        rew = np.empty(M, dtype=np.float64)
        term = np.empty(M, dtype=bool)
        margin = np.empty(M, dtype=np.float64)
        for i in range(M):
            kw = {"target_x": 0.0} if env == "overhead_crane" else {}
            n_i, r_i, t_i = fn(st[i].astype(np.float64), float(act[i]), **kw)
            assert n_i.dtype == np.float32 and n_i.shape == (D,)
            nxt[i], rew[i], term[i] = n_i, float(r_i), bool(t_i)
            margin[i] = min(dist(n_i.astype(np.float64)))
        out.update({f"{env}_states": st, f"{env}_actions": act, f"{env}_next": nxt, f"{env}_reward": rew,
                    f"{env}_term": term, f"{env}_margin": margin})
        print(f"{env:26s} {M} pairs: {int(term.sum())} terminated, {int((margin < 1e-4).sum())} within 1e-4 of a threshold, "
              f"reward in [{rew.min():.3f}, {rew.max():.3f}]")
    np.savez_compressed(OUT, **out)
    print(f"wrote {OUT} ({OUT.stat().st_size} bytes)")


if __name__ == "__main__":
    main()
