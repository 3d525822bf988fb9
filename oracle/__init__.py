"""
oracle — CPU checker for the Bellman-backup path.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import this package; product code (``dynamicprogramming_amd/``,
``src/``, ``runners/``, ``utils/``) never does.

``build(D, dynamics_src)`` compiles ``pi_oracle.cpp`` (the C++ restatement of
/root/reference/src/cuda_policy_iteration.py:183-370 and the 4D/6D twins) together
with the env's ``step_dynamics`` text into ``oracle/_build/liboracle_<hash>.so`` and
returns a thin ctypes wrapper.  The shared objects are git-ignored build products;
they travel to the GPU box with the snapshot and are rebuilt there on a miss
(g++ is in the image).
"""
from __future__ import annotations

import ctypes
import hashlib
import os
import subprocess
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
_BUILD = _HERE / "_build"
_INCLUDE = _HERE.parent / "include"
_SRC = _HERE / "pi_oracle.cpp"

CXX = os.environ.get("PI_ORACLE_CXX", "g++")
CXXFLAGS = ["-O2", "-mfma", "-msse4.1", "-ffp-contract=off", "-fno-fast-math",
            "-fopenmp", "-shared", "-fPIC", "-std=c++17"]

_f32p = ctypes.POINTER(ctypes.c_float)
_i32p = ctypes.POINTER(ctypes.c_int32)
_u8p = ctypes.POINTER(ctypes.c_uint8)
_i64p = ctypes.POINTER(ctypes.c_int64)


def _p(a, t):
    return a.ctypes.data_as(t) if a is not None else None


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def grid_metadata(bins_list):
    """bounds/shape/strides exactly as _precompute_grid_metadata (:95-109, :497-514,
    :912-937) derives them from the float32 meshgrid."""
    bins32 = [np.asarray(b).astype(np.float32) for b in bins_list]
    lo = np.array([b.min() for b in bins32], dtype=np.float32)
    hi = np.array([b.max() for b in bins32], dtype=np.float32)
    shape = np.array([len(np.unique(b)) for b in bins32], dtype=np.int32)
    strides = np.ones(len(bins32), dtype=np.int32)
    for d in range(len(bins32) - 2, -1, -1):
        strides[d] = strides[d + 1] * shape[d + 1]
    return lo, hi, shape, strides


def states_from_bins(bins_list):
    """column_stack(meshgrid(ij)) as float32 (n, D) — :84-87, :482-489, :896-904."""
    grids = np.meshgrid(*bins_list, indexing="ij")
    return np.column_stack([g.ravel() for g in grids]).astype(np.float32)


def default_threads() -> int:
    """Host threads the checker uses: the affinity mask, capped at 16 (a 1-GPU box's share)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    env = os.environ.get("PI_ORACLE_THREADS") or os.environ.get("OMP_NUM_THREADS")
    if env:
        n = min(n, int(env))
    return max(1, min(n, 16))


class OracleLib:
    def __init__(self, path: Path, D: int):
        self.path = Path(path)
        self.D = D
        self.C = 1 << D
        lib = ctypes.CDLL(str(path))
        self._lib = lib
        lib.oracle_dim.restype = ctypes.c_int
        lib.oracle_uses_libm.restype = ctypes.c_int
        assert lib.oracle_dim() == D
        self.libm = bool(lib.oracle_uses_libm())
        self.threads = 1
        if hasattr(lib, "oracle_set_threads"):
            lib.oracle_set_threads.argtypes = [ctypes.c_int]
            lib.oracle_set_threads.restype = None
            self.set_threads(default_threads())
        lib.oracle_interp.argtypes = [ctypes.c_int64, _f32p, _f32p, _f32p, _i32p, _i32p, _i32p, _f32p]
        lib.oracle_interp.restype = None
        lib.oracle_step.argtypes = [ctypes.c_int64, _f32p, _f32p, _f32p, _f32p, _u8p]
        lib.oracle_step.restype = None
        lib.oracle_eval_sweep.argtypes = [_f32p, _f32p, _i32p, _f32p, _f32p, _u8p, _f32p, _f32p,
                                          _i32p, _i32p, ctypes.c_int64, ctypes.c_int64, ctypes.c_float]
        lib.oracle_eval_sweep.restype = ctypes.c_float
        lib.oracle_improve_sweep.argtypes = [_f32p, _f32p, ctypes.c_int32, _i32p, _f32p, _u8p, _f32p,
                                             _f32p, _i32p, _i32p, ctypes.c_int64, ctypes.c_int64,
                                             ctypes.c_float, _f32p, _f32p]
        lib.oracle_improve_sweep.restype = ctypes.c_int64
        if hasattr(lib, "oracle_value_sweep"):
            lib.oracle_value_sweep.argtypes = [_f32p, _f32p, ctypes.c_int32, _i32p, _f32p, _f32p, _u8p,
                                               _f32p, _f32p, _i32p, _i32p, ctypes.c_int64,
                                               ctypes.c_int64, ctypes.c_float, _i64p]
            lib.oracle_value_sweep.restype = ctypes.c_float
        if hasattr(lib, "oracle_eval_points"):
            lib.oracle_eval_points.argtypes = [ctypes.c_int64, _f32p, _f32p, _f32p, _f32p, _f32p, _i32p, _i32p,
                                               ctypes.c_float, _f32p]
            lib.oracle_eval_points.restype = None
            lib.oracle_improve_points.argtypes = [ctypes.c_int64, _f32p, _f32p, ctypes.c_int32, _f32p, _f32p, _f32p,
                                                  _i32p, _i32p, ctypes.c_float, _i32p, _f32p]
            lib.oracle_improve_points.restype = None
        lib.oracle_run.argtypes = [_f32p, _f32p, ctypes.c_int32, _i32p, _f32p, _f32p, _u8p, _f32p,
                                   _f32p, _i32p, _i32p, ctypes.c_int64, ctypes.c_float, ctypes.c_double,
                                   ctypes.c_int32, ctypes.c_int32, _f32p, _i64p, _i32p]
        lib.oracle_run.restype = None

    def set_threads(self, n: int) -> None:
        """OpenMP threads for the sweeps (results do not depend on it: Jacobi sweeps)."""
        if hasattr(self._lib, "oracle_set_threads"):
            self.threads = int(n)
            self._lib.oracle_set_threads(int(n))

    # -- K1/K4/K7 ---------------------------------------------------------------
    def interp(self, pts, lo, hi, shape, strides):
        pts = _f32(pts).reshape(-1, self.D)
        m = len(pts)
        idxs = np.empty((m, self.C), dtype=np.int32)
        wgts = np.empty((m, self.C), dtype=np.float32)
        lo, hi, shape, strides = _f32(lo), _f32(hi), _i32(shape), _i32(strides)
        self._lib.oracle_interp(m, _p(pts, _f32p), _p(lo, _f32p), _p(hi, _f32p), _p(shape, _i32p),
                                _p(strides, _i32p), _p(idxs, _i32p), _p(wgts, _f32p))
        return idxs, wgts

    # -- env plugin -------------------------------------------------------------
    def step(self, states, act):
        states = _f32(states).reshape(-1, self.D)
        act = _f32(np.broadcast_to(act, (len(states),)))
        nxt = np.empty_like(states)
        rew = np.empty(len(states), dtype=np.float32)
        term = np.empty(len(states), dtype=np.uint8)
        self._lib.oracle_step(len(states), _p(states, _f32p), _p(act, _f32p), _p(nxt, _f32p),
                              _p(rew, _f32p), _p(term, _u8p))
        return nxt, rew, term.astype(bool)

    # -- K2/K5/K8 + K10 ---------------------------------------------------------
    def eval_sweep(self, states, actions, policy, V, is_term, lo, hi, shape, strides, gamma,
                   s0=0, s1=None, out=None):
        states = _f32(states)
        n = len(states)
        s1 = n if s1 is None else s1
        actions, policy, V = _f32(actions), _i32(policy), _f32(V)
        term = np.ascontiguousarray(is_term, dtype=np.uint8)
        newV = np.array(V, copy=True) if out is None else out
        lo, hi, shape, strides = _f32(lo), _f32(hi), _i32(shape), _i32(strides)
        delta = self._lib.oracle_eval_sweep(_p(states, _f32p), _p(actions, _f32p), _p(policy, _i32p),
                                            _p(V, _f32p), _p(newV, _f32p), _p(term, _u8p),
                                            _p(lo, _f32p), _p(hi, _f32p), _p(shape, _i32p),
                                            _p(strides, _i32p), s0, s1, np.float32(gamma))
        return newV, float(delta)

    # -- K3/K6/K9 + K11 ---------------------------------------------------------
    def improve_sweep(self, states, actions, policy, V, is_term, lo, hi, shape, strides, gamma,
                      s0=0, s1=None, want_q=False):
        states = _f32(states)
        n = len(states)
        s1 = n if s1 is None else s1
        actions, V = _f32(actions), _f32(V)
        policy = np.array(policy, dtype=np.int32, copy=True)
        term = np.ascontiguousarray(is_term, dtype=np.uint8)
        lo, hi, shape, strides = _f32(lo), _f32(hi), _i32(shape), _i32(strides)
        qb = np.zeros(n, dtype=np.float32) if want_q else None
        qs = np.zeros(n, dtype=np.float32) if want_q else None
        changed = self._lib.oracle_improve_sweep(_p(states, _f32p), _p(actions, _f32p), len(actions),
                                                 _p(policy, _i32p), _p(V, _f32p), _p(term, _u8p),
                                                 _p(lo, _f32p), _p(hi, _f32p), _p(shape, _i32p),
                                                 _p(strides, _i32p), s0, s1, np.float32(gamma),
                                                 _p(qb, _f32p), _p(qs, _f32p))
        if want_q:
            return policy, int(changed), qb, qs
        return policy, int(changed)

    # -- value-iteration sweep ------------------------------------------------------
    def value_sweep(self, states, actions, policy, V, is_term, lo, hi, shape, strides, gamma,
                    s0=0, s1=None, out=None):
        states = _f32(states)
        n = len(states)
        s1 = n if s1 is None else s1
        actions, V = _f32(actions), _f32(V)
        policy = np.array(policy, dtype=np.int32, copy=True)
        term = np.ascontiguousarray(is_term, dtype=np.uint8)
        newV = np.array(V, copy=True) if out is None else out
        lo, hi, shape, strides = _f32(lo), _f32(hi), _i32(shape), _i32(strides)
        changed = np.zeros(1, dtype=np.int64)
        delta = self._lib.oracle_value_sweep(_p(states, _f32p), _p(actions, _f32p), len(actions),
                                             _p(policy, _i32p), _p(V, _f32p), _p(newV, _f32p),
                                             _p(term, _u8p), _p(lo, _f32p), _p(hi, _f32p),
                                             _p(shape, _i32p), _p(strides, _i32p), s0, s1,
                                             np.float32(gamma), _p(changed, _i64p))
        return newV, policy, float(delta), int(changed[0])

    # -- the same backups for a list of states (coordinates) ------------------------
    def eval_points(self, coords, action_values, V, lo, hi, shape, strides, gamma):
        """V'(s) = r + gamma E[V](s') for the non-terminal states at `coords` (m, D) under the action VALUES
        `action_values` (m,); V is the whole table in the reference's flat order."""
        coords = _f32(coords).reshape(-1, self.D)
        m = len(coords)
        act, V = _f32(np.broadcast_to(action_values, (m,))), _f32(V)
        lo, hi, shape, strides = _f32(lo), _f32(hi), _i32(shape), _i32(strides)
        out = np.empty(m, dtype=np.float32)
        self._lib.oracle_eval_points(m, _p(coords, _f32p), _p(act, _f32p), _p(V, _f32p), _p(lo, _f32p), _p(hi, _f32p),
                                     _p(shape, _i32p), _p(strides, _i32p), np.float32(gamma), _p(out, _f32p))
        return out

    def improve_points(self, coords, actions, V, lo, hi, shape, strides, gamma):
        """(greedy action index, its value) of the states at `coords` (m, D): strict '>' from -1e30."""
        coords = _f32(coords).reshape(-1, self.D)
        m = len(coords)
        actions, V = _f32(actions), _f32(V)
        lo, hi, shape, strides = _f32(lo), _f32(hi), _i32(shape), _i32(strides)
        best = np.empty(m, dtype=np.int32)
        best_q = np.empty(m, dtype=np.float32)
        self._lib.oracle_improve_points(m, _p(coords, _f32p), _p(actions, _f32p), len(actions), _p(V, _f32p),
                                        _p(lo, _f32p), _p(hi, _f32p), _p(shape, _i32p), _p(strides, _i32p),
                                        np.float32(gamma), _p(best, _i32p), _p(best_q, _f32p))
        return best, best_q

    # -- run() ------------------------------------------------------------------
    def run(self, states, actions, is_term, lo, hi, shape, strides, gamma, theta, max_eval_iter,
            max_pi_iter, terminal_value=0.0, V0=None, policy0=None):
        states = _f32(states)
        n = len(states)
        actions = _f32(actions)
        term = np.ascontiguousarray(is_term, dtype=np.uint8)
        V = np.zeros(n, dtype=np.float32) if V0 is None else np.array(V0, dtype=np.float32, copy=True)
        if V0 is None and term.any():
            V[term.astype(bool)] = np.float32(terminal_value)
        Vtmp = V.copy()
        policy = np.zeros(n, dtype=np.int32) if policy0 is None else np.array(policy0, dtype=np.int32, copy=True)
        lo, hi, shape, strides = _f32(lo), _f32(hi), _i32(shape), _i32(strides)
        V_out = np.empty(n, dtype=np.float32)
        stats = np.zeros(3, dtype=np.int64)
        per_iter = np.zeros(max_pi_iter, dtype=np.int32)
        self._lib.oracle_run(_p(states, _f32p), _p(actions, _f32p), len(actions), _p(policy, _i32p),
                             _p(V, _f32p), _p(Vtmp, _f32p), _p(term, _u8p), _p(lo, _f32p), _p(hi, _f32p),
                             _p(shape, _i32p), _p(strides, _i32p), n, np.float32(gamma),
                             float(theta), max_eval_iter, max_pi_iter, _p(V_out, _f32p),
                             _p(stats, _i64p), _p(per_iter, _i32p))
        return {
            "value_function": V_out, "policy": policy, "outer_iterations": int(stats[0]),
            "eval_sweeps": int(stats[1]), "stable": bool(stats[2]),
            "sweeps_per_iter": per_iter[: int(stats[0])].copy(),
        }


_cache: dict = {}


def build(D: int, dynamics_src: str, libm: bool = False) -> OracleLib:
    """Compile (or reuse) the oracle for one (D, dynamics text, arithmetic mode)."""
    assert D in (2, 4, 6)
    src_text = _SRC.read_text()
    math_text = (_INCLUDE / "pi_math.h").read_text()
    key = hashlib.sha256("\0".join([str(D), dynamics_src, str(libm), src_text, math_text,
                                    " ".join(CXXFLAGS)]).encode()).hexdigest()[:20]
    if key in _cache:
        return _cache[key]
    _BUILD.mkdir(exist_ok=True)
    so = _BUILD / f"liboracle_{key}.so"
    if not so.exists():
        dyn = _BUILD / f"dyn_{key}.inc"
        dyn.write_text(dynamics_src)
        tmp = _BUILD / f".tmp_{os.getpid()}_{key}.so"
        cmd = [CXX, *CXXFLAGS, f"-DPI_D={D}", f'-DPI_DYN_FILE="{dyn}"', f"-I{_INCLUDE}"]
        if libm:
            cmd.append("-DPI_ORACLE_LIBM")
        cmd += [str(_SRC), "-o", str(tmp)]
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode != 0:
            raise RuntimeError(f"oracle build failed:\n{' '.join(cmd)}\n{res.stderr}")
        os.replace(tmp, so)
    lib = OracleLib(so, D)
    _cache[key] = lib
    return lib
