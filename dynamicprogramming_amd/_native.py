"""
ctypes binding of libpi_mi355.so (C ABI: include/pi_mi355.h).

The library is the only compute path of this package.  There is no CPU or eager
fallback: if the shared object is missing or fails to load, importing this module
still succeeds (so ``load()``-only workflows work on machines without ROCm) but
``lib()`` raises, and so does every solver constructor.
"""
from __future__ import annotations

import ctypes
import os
from pathlib import Path

import numpy as np

_PKG = Path(__file__).resolve().parent
LIB_PATH = _PKG / "libpi_mi355.so"
KERNEL_CACHE = Path(os.environ.get("PI_MI355_KERNEL_CACHE", str(_PKG / "_kcache")))

_f32p = ctypes.POINTER(ctypes.c_float)
_i32p = ctypes.POINTER(ctypes.c_int32)
_vp = ctypes.c_void_p

# name -> (restype, argtypes); must list every symbol include/pi_mi355.h declares.
SIGNATURES = {
    "pi_abi_version": (ctypes.c_int, []),
    "pi_last_error": (ctypes.c_char_p, []),
    "pi_create": (_vp, [ctypes.c_int, ctypes.c_int, _i32p, _f32p, _f32p,
                        ctypes.POINTER(_f32p), _f32p, ctypes.c_int]),
    "pi_destroy": (None, [_vp]),
    "pi_compile": (ctypes.c_int, [_vp, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p,
                                  ctypes.c_size_t]),
    "pi_kernel_source": (ctypes.c_size_t, [_vp, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_size_t]),
    "pi_eval_sweep": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, ctypes.c_int64, ctypes.c_int64,
                                     ctypes.c_float, _vp, _vp]),
    "pi_eval_sweeps": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, ctypes.c_int64, ctypes.c_int64,
                                      ctypes.c_float, ctypes.c_int, _vp, _vp]),
    "pi_policy_evaluation": (ctypes.c_int, [_vp, _vp, _vp, _vp, ctypes.c_float, ctypes.c_double, ctypes.c_int,
                                            ctypes.c_int, _vp, _vp, _vp, _vp]),
    "pi_policy_iteration": (ctypes.c_int, [_vp, _vp, _vp, _vp, ctypes.c_float, ctypes.c_double, ctypes.c_int,
                                           ctypes.c_int, ctypes.c_int, _vp, _vp, _vp]),
    "pi_improve_sweep": (ctypes.c_int, [_vp, _vp, _vp, _vp, ctypes.c_int64, ctypes.c_int64,
                                        ctypes.c_float, _vp, _vp]),
    "pi_value_sweep": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, ctypes.c_int64, ctypes.c_int64,
                                      ctypes.c_float, _vp, _vp, _vp]),
    "pi_reach_planes": (ctypes.c_int, [_vp, _vp, ctypes.c_int64, ctypes.c_int64, ctypes.c_int, _vp, _vp]),
    "pi_reach_depth_max": (ctypes.c_int, [_vp]),
    "pi_reach_units": (ctypes.c_int, [_vp, _vp, ctypes.c_int64, ctypes.c_int64, ctypes.c_int, _vp, _vp]),
    "pi_comm_unique_id": (ctypes.c_int, [_vp]),
    "pi_comm_init": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int, _vp]),
    "pi_comm_init_local": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int, ctypes.c_char_p]),
    "pi_p2p_describe": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int, _vp, _vp, ctypes.c_int, _vp]),
    "pi_comm_init_p2p": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int, _vp, ctypes.c_char_p]),
    "pi_p2p_compile_check": (ctypes.c_int, [ctypes.c_char_p]),
    "pi_comm_destroy": (ctypes.c_int, [_vp]),
    "pi_comm_info": (ctypes.c_int, [_vp, ctypes.c_int]),
    "pi_allgather_V": (ctypes.c_int, [_vp, _vp, ctypes.c_int64, _vp]),
    "pi_allgather_policy": (ctypes.c_int, [_vp, _vp, ctypes.c_int64, _vp]),
    "pi_allreduce_max_f32": (ctypes.c_int, [_vp, _vp, _vp]),
    "pi_allreduce_sum_u32": (ctypes.c_int, [_vp, _vp, _vp]),
    "pi_plan_segments": (ctypes.c_int64, [ctypes.c_int, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64,
                                          ctypes.c_int64, _vp, _vp, ctypes.c_int64]),
    "pi_exchange_plan": (ctypes.c_int, [_vp, _vp, ctypes.c_int64, ctypes.c_int, ctypes.c_int, _vp, _vp]),
    "pi_plan_ranges": (ctypes.c_int64, [_vp, _vp, ctypes.c_int64]),
    "pi_exchange_V": (ctypes.c_int, [_vp, _vp, _vp]),
    "pi_eval_sweep_part": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, ctypes.c_int, ctypes.c_float, _vp]),
    "pi_eval_sweeps_sharded": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, ctypes.c_float, ctypes.c_int,
                                              _vp, _vp]),
    "pi_improve_sweep_sharded": (ctypes.c_int, [_vp, _vp, _vp, _vp, ctypes.c_float, _vp, _vp]),
    "pi_probe_step": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, ctypes.c_int64, _vp]),
    "pi_probe_interp": (ctypes.c_int, [_vp, _vp, _vp, _vp, ctypes.c_int64, _vp]),
    "pi_probe_coords": (ctypes.c_int, [_vp, ctypes.c_int64, ctypes.c_int64, _vp, ctypes.c_int, _vp]),
    "pi_infer_create": (_vp, [ctypes.c_int, ctypes.c_int, _f32p, _f32p, _i32p, _i32p, _i32p, ctypes.c_int64,
                              ctypes.c_char_p]),
    "pi_infer_destroy": (None, [_vp]),
    "pi_infer_set_policy": (ctypes.c_int, [_vp, _i32p, ctypes.c_int64, _f32p, ctypes.c_int]),
    "pi_infer_query": (ctypes.c_int, [_vp, _vp, ctypes.c_int64, _vp, _vp, _vp, _vp]),
    "pi_plan_schedule": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_int,
                                        ctypes.POINTER(ctypes.c_uint64)]),
    "pi_set_option": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int64]),
    "pi_info": (ctypes.c_int64, [_vp, ctypes.c_int]),
    "pi_debug_report": (ctypes.c_int, [_vp, ctypes.POINTER(ctypes.c_uint32)]),
    "pi_prepare_mask": (ctypes.c_int, [_vp, _vp, _vp]),
    "pi_prepare_mask_range": (ctypes.c_int, [_vp, _vp, ctypes.c_int64, ctypes.c_int64, _vp]),
    "pi_live_list": (ctypes.c_int64, [_vp, _vp, ctypes.c_int64, _vp]),
    "pi_eval_begin": (ctypes.c_int, [_vp, _vp, _vp, _vp]),
    "pi_eval_end": (ctypes.c_int, [_vp]),
}

ABI_VERSION = 10
_lib = None
_load_error: Exception | None = None


class NativeError(RuntimeError):
    pass


def kernel_source_hash() -> str:
    """Fingerprint of the device code (kernel template + math header): profiles record it so that
    counters measured on one version of the kernels are never quoted for another (bench.py)."""
    import hashlib
    h = hashlib.sha256()
    for path in (_PKG / "csrc" / "pi_sweep_kernels.hip", _PKG.parent / "include" / "pi_math.h"):
        h.update(path.read_bytes())
    return h.hexdigest()[:16]


def lib() -> ctypes.CDLL:
    """The loaded library; raises NativeError (never falls back) when unavailable."""
    global _lib, _load_error
    if _lib is not None:
        return _lib
    if _load_error is not None:
        raise NativeError(str(_load_error)) from _load_error
    try:
        # torch ships its own libamdhip64 (same SONAME as /opt/rocm's).  Two HIP runtimes in
        # one process do not work ("no ROCm-capable device"), so make sure torch's copy is the
        # one already mapped before ours is resolved: device pointers and streams handed
        # across the C ABI then belong to the same runtime.
        import torch  # noqa: F401
        if not LIB_PATH.exists():
            raise FileNotFoundError(
                f"{LIB_PATH} not found — build it with `python -c 'import __graft_entry__ as g; "
                f"g.build()'` or `make -C dynamicprogramming_amd/csrc`")
        handle = ctypes.CDLL(str(LIB_PATH))
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype = res
            fn.argtypes = args
        if handle.pi_abi_version() != ABI_VERSION:
            raise RuntimeError(f"libpi_mi355 ABI {handle.pi_abi_version()} != binding {ABI_VERSION}")
        _lib = handle
        return _lib
    except Exception as exc:  # noqa: BLE001 - recorded and re-raised on every call
        _load_error = exc
        raise NativeError(f"libpi_mi355.so unavailable: {exc}") from exc


def available() -> bool:
    try:
        lib()
        return True
    except NativeError:
        return False


def last_error() -> str:
    return lib().pi_last_error().decode(errors="replace")


def _check(rc: int, what: str) -> None:
    if rc != 0:
        raise NativeError(f"{what} failed: {last_error()}")


def p2p_compile_check(cache_dir: Path | str | None = KERNEL_CACHE) -> None:
    """The peer-to-peer transport's device code builds for gfx950 (needs no GPU; fills the kernel cache)."""
    cdir = None
    if cache_dir is not None:
        Path(cache_dir).mkdir(parents=True, exist_ok=True)
        cdir = str(cache_dir).encode()
    _check(lib().pi_p2p_compile_check(cdir), "pi_p2p_compile_check")


def comm_unique_id() -> bytes:
    """A fresh 128-byte RCCL id (create on one rank, hand to every rank's `Engine.comm_init`)."""
    buf = ctypes.create_string_buffer(128)
    _check(lib().pi_comm_unique_id(buf), "pi_comm_unique_id")
    return buf.raw


def plan_segments(world, g0, stride0, n_states, per, reach) -> np.ndarray:
    """Host-only planner of the halo exchange (pi_plan_segments): reach is a (world, g0) bool
    array over g0 units of stride0 consecutive states each (planes of dimension 0, or rows
    (i0, i1)); returns an (m, 4) int64 array of {src, dst, a, b}."""
    r = np.ascontiguousarray(reach, dtype=np.uint8)
    assert r.shape == (world, g0)
    fn = lib().pi_plan_segments
    m = fn(world, g0, stride0, n_states, per, r.ctypes.data_as(_vp), None, 0)
    if m < 0:
        raise NativeError(f"pi_plan_segments failed: {last_error()}")
    out = np.zeros((max(int(m), 1), 4), dtype=np.int64)
    fn(world, g0, stride0, n_states, per, r.ctypes.data_as(_vp), out.ctypes.data_as(_vp), int(m))
    return out[: int(m)]


class Engine:
    """One pi_handle: a grid + action set + (after compile) its specialised kernels."""

    def __init__(self, D, grid_shape, lo, hi, bins, actions, device: int = -1, order=None):
        """`order` (optional): memory order of the dimensions — order[k] is the user dimension stored as memory
        dimension k, 0 = slowest (pi_set_option 4).  Every flat state index and every device array of this engine
        is then in that order (`to_memory` / `to_user` convert whole-grid arrays); results do not depend on it."""
        L = lib()
        self.D = int(D)
        self._shape = np.ascontiguousarray(grid_shape, dtype=np.int32)
        self._lo = np.ascontiguousarray(lo, dtype=np.float32)
        self._hi = np.ascontiguousarray(hi, dtype=np.float32)
        self._bins = [np.ascontiguousarray(b, dtype=np.float32) for b in bins]
        self._actions = np.ascontiguousarray(actions, dtype=np.float32)
        assert len(self._bins) == self.D and all(len(b) == g for b, g in zip(self._bins, self._shape))
        arr = (_f32p * self.D)(*[b.ctypes.data_as(_f32p) for b in self._bins])
        self._h = L.pi_create(int(device), self.D, self._shape.ctypes.data_as(_i32p),
                              self._lo.ctypes.data_as(_f32p), self._hi.ctypes.data_as(_f32p), arr,
                              self._actions.ctypes.data_as(_f32p), len(self._actions))
        if not self._h:
            raise NativeError(f"pi_create failed: {last_error()}")
        self.device = int(device)
        self.n_states = int(L.pi_info(self._h, 0))
        self.compile_log = ""
        self.order = tuple(range(self.D))
        if order is not None and tuple(int(d) for d in order) != self.order:
            order = tuple(int(d) for d in order)
            if sorted(order) != list(range(self.D)):
                raise NativeError(f"memory order {order} is not a permutation of the {self.D} dimensions")
            self.set_option(4, sum(d << (3 * k) for k, d in enumerate(order)))
            self.order = order

    # -- memory order of the dimensions ----------------------------------------
    def to_memory(self, a):
        """A whole-grid array (numpy or torch, n_states entries, the user's row-major order) in this engine's
        memory order.  The identity when no order was given."""
        if self.order == tuple(range(self.D)):
            return a
        shape = [int(g) for g in self._shape]
        b = a.reshape(shape)
        b = b.permute(*self.order) if hasattr(b, "permute") else b.transpose(self.order)
        return b.reshape(-1) if not hasattr(b, "contiguous") else b.contiguous().reshape(-1)

    def to_user(self, a):
        """The inverse of `to_memory`."""
        if self.order == tuple(range(self.D)):
            return a
        shape = [int(self._shape[d]) for d in self.order]
        inv = [self.order.index(d) for d in range(self.D)]
        b = a.reshape(shape)
        b = b.permute(*inv) if hasattr(b, "permute") else b.transpose(inv)
        return b.reshape(-1) if not hasattr(b, "contiguous") else b.contiguous().reshape(-1)

    # -- compilation ---------------------------------------------------------
    def kernel_source(self, dynamics_src: str) -> str:
        L = lib()
        src = dynamics_src.encode()
        n = L.pi_kernel_source(self._h, src, None, 0)
        buf = ctypes.create_string_buffer(n + 1)
        L.pi_kernel_source(self._h, src, buf, n + 1)
        return buf.value.decode()

    def compile(self, dynamics_src: str, cache_dir: Path | str | None = KERNEL_CACHE) -> None:
        L = lib()
        log = ctypes.create_string_buffer(1 << 16)
        cdir = None
        if cache_dir is not None:
            Path(cache_dir).mkdir(parents=True, exist_ok=True)
            cdir = str(cache_dir).encode()
        rc = L.pi_compile(self._h, dynamics_src.encode(), cdir, log, len(log))
        self.compile_log = log.value.decode(errors="replace")
        if rc != 0:
            raise NativeError(f"kernel compilation failed:\n{last_error()}")

    def info(self, what: int) -> int:
        return int(lib().pi_info(self._h, what))

    def prepare_mask(self, term, stream=0, s_begin=None, s_end=None) -> int:
        """Let later sweeps of a batch visit only the non-terminal states of the mask at `term` (0 drops the list);
        [s_begin, s_end) restricts the list to a rank's shard.  Returns the number of live states listed, 0 when
        the library keeps none."""
        if s_begin is None:
            _check(lib().pi_prepare_mask(self._h, term or None, stream or None), "pi_prepare_mask")
        else:
            _check(lib().pi_prepare_mask_range(self._h, term or None, int(s_begin), int(s_end), stream or None),
                   "pi_prepare_mask_range")
        return self.info(16)

    def live_list(self, d_out=0, capacity=0, stream=0) -> int:
        """Copy the live-state list into the device buffer at `d_out` (`capacity` int32 entries); returns its length
        (0: the library keeps no list).  Without a buffer: the length only."""
        m = int(lib().pi_live_list(self._h, d_out or None, int(capacity), stream or None))
        if m < 0:
            raise NativeError(f"pi_live_list failed: {last_error()}")
        return m

    def eval_begin(self, policy, term, stream=0) -> int:
        """Start of one policy evaluation under the policy at `policy`: returns the length of the shorter list the
        following whole-grid batches may use (0: none).  Pair with eval_end()."""
        _check(lib().pi_eval_begin(self._h, policy, term or None, stream or None), "pi_eval_begin")
        return self.info(17)

    def eval_end(self) -> None:
        _check(lib().pi_eval_end(self._h), "pi_eval_end")

    def debug_report(self) -> dict:
        """Checked build only (PI_MI355_DEBUG=1 when the engine was created): index violations of the sweeps
        launched since the last report — {"violations", "kind" (1 policy entry, 2 cell), "where", "value"}."""
        out = (ctypes.c_uint32 * 4)()
        _check(lib().pi_debug_report(self._h, out), "pi_debug_report")
        return {"violations": int(out[0]), "kind": int(out[1]), "where": int(out[2]), "value": int(out[3])}

    # -- launches (raw device pointers; all asynchronous on `stream`) ---------
    def eval_sweep(self, V, Vnew, policy, term, s_begin, s_end, gamma, d_delta=0, stream=0):
        _check(lib().pi_eval_sweep(self._h, V, Vnew, policy, term, s_begin, s_end, gamma,
                                   d_delta or None, stream or None), "pi_eval_sweep")

    def eval_sweeps(self, Va, Vb, policy, term, s_begin, s_end, gamma, n_sweeps, d_delta=0, stream=0):
        _check(lib().pi_eval_sweeps(self._h, Va, Vb, policy, term, s_begin, s_end, gamma, n_sweeps,
                                    d_delta or None, stream or None), "pi_eval_sweeps")

    def policy_evaluation(self, V, policy, term, gamma, theta, max_sweeps, check_interval, d_sweeps, d_delta,
                          d_residual_log, stream=0):
        _check(lib().pi_policy_evaluation(self._h, V, policy, term, gamma, float(theta), int(max_sweeps),
                                          int(check_interval), d_sweeps, d_delta or None, d_residual_log,
                                          stream or None), "pi_policy_evaluation")

    def policy_iteration(self, V, policy, term, gamma, theta, max_eval_sweeps, check_interval, max_pi_iter, d_result,
                         d_iter_log, stream=0):
        _check(lib().pi_policy_iteration(self._h, V, policy, term, gamma, float(theta), int(max_eval_sweeps),
                                         int(check_interval), int(max_pi_iter), d_result, d_iter_log, stream or None),
               "pi_policy_iteration")

    def improve_sweep(self, V, policy, term, s_begin, s_end, gamma, d_changed=0, stream=0):
        _check(lib().pi_improve_sweep(self._h, V, policy, term, s_begin, s_end, gamma,
                                      d_changed or None, stream or None), "pi_improve_sweep")

    def value_sweep(self, V, Vnew, policy, term, s_begin, s_end, gamma, d_delta=0, d_changed=0, stream=0):
        _check(lib().pi_value_sweep(self._h, V, Vnew, policy, term, s_begin, s_end, gamma,
                                    d_delta or None, d_changed or None, stream or None), "pi_value_sweep")

    def reach_planes(self, term, s_begin, s_end, d_bitmap, stream=0, dim=0):
        _check(lib().pi_reach_planes(self._h, term, s_begin, s_end, int(dim), d_bitmap, stream or None),
               "pi_reach_planes")

    def reach_depth_max(self) -> int:
        return int(lib().pi_reach_depth_max(self._h))

    def reach_units(self, term, s_begin, s_end, depth, d_bitmap, stream=0):
        _check(lib().pi_reach_units(self._h, term, s_begin, s_end, int(depth), d_bitmap, stream or None),
               "pi_reach_units")

    def probe_coords(self, s_begin, s_end, out, chunks_per_workgroup=1, stream=0):
        _check(lib().pi_probe_coords(self._h, s_begin, s_end, out, int(chunks_per_workgroup),
                                     stream or None), "pi_probe_coords")

    def plan_schedule(self, block, first, count, total=None, chunks_per_workgroup=1):
        """{grid_x, grid_y, period, phase, cpw} of the launch the sweeps would make (pi_plan_schedule)."""
        out = (ctypes.c_uint64 * 6)()
        _check(lib().pi_plan_schedule(self._h, int(block), int(first), int(count),
                                      int(self.info(0) if total is None else total), int(chunks_per_workgroup), out),
               "pi_plan_schedule")
        return dict(zip(("grid_x", "grid_y", "period", "phase", "cpw"), (int(v) for v in out)))

    def set_option(self, what: int, value: int) -> None:
        _check(lib().pi_set_option(self._h, int(what), int(value)), "pi_set_option")

    # -- multi-GPU (RCCL or the in-process test transport) ---------------------
    def comm_init(self, rank: int, world: int, unique_id: bytes) -> None:
        assert len(unique_id) == 128
        buf = ctypes.create_string_buffer(unique_id, 128)
        _check(lib().pi_comm_init(self._h, int(rank), int(world), buf), "pi_comm_init")

    def comm_init_local(self, rank: int, world: int, group: str) -> None:
        _check(lib().pi_comm_init_local(self._h, int(rank), int(world), group.encode()),
               "pi_comm_init_local")

    P2P_DESC_BYTES = 512

    def p2p_describe(self, rank: int, world: int, buffers) -> bytes:
        """Register `buffers` — (device pointer, bytes) pairs: the arrays sends and receives will name — for the
        peer-to-peer transport and return this rank's 512-byte descriptor (pi_p2p_describe)."""
        n = len(buffers)
        ptrs = (ctypes.c_void_p * max(n, 1))(*[int(p) for p, _ in buffers])
        sizes = (ctypes.c_int64 * max(n, 1))(*[int(b) for _, b in buffers])
        out = ctypes.create_string_buffer(self.P2P_DESC_BYTES)
        _check(lib().pi_p2p_describe(self._h, int(rank), int(world), ptrs, sizes, n, out), "pi_p2p_describe")
        return out.raw

    def comm_init_p2p(self, rank: int, world: int, descriptors, cache_dir: Path | str | None = KERNEL_CACHE) -> None:
        """`descriptors`: the p2p_describe results of ALL ranks, ordered by rank (pi_comm_init_p2p)."""
        blob = b"".join(descriptors)
        if len(blob) != self.P2P_DESC_BYTES * int(world):
            raise NativeError(f"comm_init_p2p needs {world} descriptors of {self.P2P_DESC_BYTES} bytes")
        cdir = None
        if cache_dir is not None:
            Path(cache_dir).mkdir(parents=True, exist_ok=True)
            cdir = str(cache_dir).encode()
        _check(lib().pi_comm_init_p2p(self._h, int(rank), int(world), ctypes.create_string_buffer(blob, len(blob)), cdir),
               "pi_comm_init_p2p")

    def comm_destroy(self) -> None:
        _check(lib().pi_comm_destroy(self._h), "pi_comm_destroy")

    def comm_info(self, what: int) -> int:
        return int(lib().pi_comm_info(self._h, int(what)))

    def allgather_V(self, V_full, shard_elems, stream=0):
        _check(lib().pi_allgather_V(self._h, V_full, shard_elems, stream or None), "pi_allgather_V")

    def allgather_policy(self, policy_full, shard_elems, stream=0):
        _check(lib().pi_allgather_policy(self._h, policy_full, shard_elems, stream or None),
               "pi_allgather_policy")

    def allreduce_max_f32(self, d_value, stream=0):
        _check(lib().pi_allreduce_max_f32(self._h, d_value, stream or None), "pi_allreduce_max_f32")

    def allreduce_sum_u32(self, d_value, stream=0):
        _check(lib().pi_allreduce_sum_u32(self._h, d_value, stream or None), "pi_allreduce_sum_u32")

    def exchange_plan(self, term, per, mode=0, overlap=True, stream=0):
        info = (ctypes.c_int64 * 5)()
        _check(lib().pi_exchange_plan(self._h, term, int(per), int(mode), int(bool(overlap)), info,
                                      stream or None), "pi_exchange_plan")
        return {"mode": "halo" if info[0] == 2 else "allgather", "recv_elems": int(info[1]),
                "send_elems": int(info[2]), "send_ranges": int(info[3]), "interior_ranges": int(info[4]),
                "row_exact": int(lib().pi_comm_info(self._h, 5)) == 1,
                "reach_units": {1: "planes of dimension 0", 2: "rows (i0, i1)"}.get(
                    int(lib().pi_comm_info(self._h, 4)), "none")}

    def plan_ranges(self):
        """[(kind, begin, end)] of the current exchange plan: kind 0 = swept first, 1 = interior."""
        fn = lib().pi_plan_ranges
        m = fn(self._h, None, 0)
        if m < 0:
            raise NativeError(f"pi_plan_ranges failed: {last_error()}")
        buf = (ctypes.c_int64 * (3 * max(int(m), 1)))()
        fn(self._h, buf, int(m))
        return [(int(buf[3 * i]), int(buf[3 * i + 1]), int(buf[3 * i + 2])) for i in range(int(m))]

    def eval_sweep_part(self, V, Vnew, policy, term, part, gamma, stream=0):
        """Part 0 (swept first) or 1 (interior) of one sharded evaluation sweep, without the exchange."""
        _check(lib().pi_eval_sweep_part(self._h, V, Vnew, policy, term or None, int(part), gamma, stream or None),
               "pi_eval_sweep_part")

    def exchange_V(self, V_full, stream=0):
        _check(lib().pi_exchange_V(self._h, V_full, stream or None), "pi_exchange_V")

    def eval_sweeps_sharded(self, Va, Vb, policy, term, gamma, n_sweeps, d_delta=0, stream=0):
        _check(lib().pi_eval_sweeps_sharded(self._h, Va, Vb, policy, term, gamma, n_sweeps,
                                            d_delta or None, stream or None), "pi_eval_sweeps_sharded")

    def improve_sweep_sharded(self, V, policy, term, gamma, d_changed=0, stream=0):
        _check(lib().pi_improve_sweep_sharded(self._h, V, policy, term, gamma, d_changed or None,
                                              stream or None), "pi_improve_sweep_sharded")

    def probe_step(self, states, acts, nxt, reward, done, m, stream=0):
        _check(lib().pi_probe_step(self._h, states, acts, nxt, reward, done, m, stream or None),
               "pi_probe_step")

    def probe_interp(self, pts, idxs, wgts, m, stream=0):
        _check(lib().pi_probe_interp(self._h, pts, idxs, wgts, m, stream or None), "pi_probe_interp")

    def close(self) -> None:
        if getattr(self, "_h", None):
            lib().pi_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001 - interpreter shutdown
            pass


class InferenceEngine:
    """One pi_infer handle: batched device interpolation of a trained policy (the reference's
    utils/barycentric.py as one kernel; include/pi_mi355.h "Inference")."""

    def __init__(self, bounds_low, bounds_high, grid_shape, strides, corner_bits, device: int = -1,
                 cache_dir: Path | str | None = KERNEL_CACHE):
        L = lib()
        self._lo = np.ascontiguousarray(bounds_low, dtype=np.float32)
        self._hi = np.ascontiguousarray(bounds_high, dtype=np.float32)
        self._shape = np.ascontiguousarray(grid_shape, dtype=np.int32)
        self._strides = np.ascontiguousarray(strides, dtype=np.int32)
        self._bits = np.ascontiguousarray(corner_bits, dtype=np.int32)
        self.D = len(self._shape)
        assert self._bits.ndim == 2 and self._bits.shape[1] == self.D
        cdir = None
        if cache_dir is not None:
            Path(cache_dir).mkdir(parents=True, exist_ok=True)
            cdir = str(cache_dir).encode()
        self._h = L.pi_infer_create(int(device), self.D, self._lo.ctypes.data_as(_f32p), self._hi.ctypes.data_as(_f32p),
                                    self._shape.ctypes.data_as(_i32p), self._strides.ctypes.data_as(_i32p),
                                    self._bits.ctypes.data_as(_i32p), len(self._bits), cdir)
        if not self._h:
            raise NativeError(f"pi_infer_create failed: {last_error()}")
        self.device = int(device)
        self.n_corners = len(self._bits)

    def set_policy(self, policy, action_space) -> None:
        pol = np.ascontiguousarray(policy, dtype=np.int32)
        act = np.ascontiguousarray(action_space, dtype=np.float32)
        _check(lib().pi_infer_set_policy(self._h, pol.ctypes.data_as(_i32p), len(pol), act.ctypes.data_as(_f32p),
                                         len(act)), "pi_infer_set_policy")

    def query(self, d_points, m, d_actions=0, d_weights=0, d_indices=0, stream=0) -> None:
        _check(lib().pi_infer_query(self._h, d_points, int(m), d_actions or None, d_weights or None,
                                    d_indices or None, stream or None), "pi_infer_query")

    def close(self) -> None:
        if getattr(self, "_h", None):
            lib().pi_infer_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001 - interpreter shutdown
            pass
