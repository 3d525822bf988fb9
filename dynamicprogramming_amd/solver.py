"""
Policy-iteration solver for MI355X — host side.

Public surface = the reference's (src/cuda_policy_iteration.py): ``CudaPIConfig`` (:36-43)
and ``CudaPolicyIteration{2D,4D,6D}`` with ``__init__(bins_space, action_space, config)``,
the subclass hooks ``_dynamics_cuda_src()`` / ``_terminal_fn(states)`` /
``_allocate_tensors_and_compile()``, and ``run()``, ``policy_evaluation()``,
``policy_improvement()``, ``save()``, ``load()``; same attributes after ``run()``/``load()``
and the same ``.npz`` schema (:397-408).  A runner written against the reference imports
``src.cuda_policy_iteration`` (a shim onto this module) and works unchanged.

Underneath nothing is shared with the reference: the grid is described by D bin tables,
V / policy / mask live in torch-ROCm tensors (device memory only), and every sweep is one
call through the ctypes C ABI of libpi_mi355.so (include/pi_mi355.h), whose hipRTC-built
gfx950 kernels fuse the backup, the residual reduction and the policy-change count.
With more than one rank (one process per GPU) the flat state range is split into contiguous
shards; each rank sweeps its shard and the new values travel between ranks inside the library
(RCCL over xGMI: halo exchange of the reachable planes, or an all-gather) — see transport.py.

There is no CPU or eager fallback: constructing a solver without the native library and
a GPU raises ``RuntimeError`` exactly where the reference raises for a missing CuPy (:71-75).
"""
from __future__ import annotations

import abc
import logging
import os
import time
from dataclasses import dataclass
from itertools import product
from pathlib import Path

import numpy as np

from . import _native

try:  # the reference logs through loguru; use it when present, stdlib logging otherwise
    from loguru import logger  # type: ignore
except ImportError:  # pragma: no cover - loguru is absent on the build/GPU images
    logger = logging.getLogger("dynamicprogramming_amd")
    if not hasattr(logger, "success"):
        logger.success = logger.info  # type: ignore[attr-defined]


def _gpu_available() -> bool:
    if not _native.available():
        return False
    try:
        import torch
        return bool(torch.cuda.is_available())
    except Exception:  # noqa: BLE001
        return False


GPU_AVAILABLE = _gpu_available()

SYNC_INTERVAL = 25   # the reference looks at the residual on sweeps 0, 25, 50, ... (:303, :325)


@dataclass
class CudaPIConfig:
    """Solver settings; the five reference fields (:36-43) with the reference defaults."""
    gamma: float = 0.99           # discount factor
    theta: float = 1e-4           # residual threshold that ends a policy evaluation
    max_eval_iter: int = 10_000   # sweeps per policy evaluation, at most
    max_pi_iter: int = 50         # outer evaluate/improve iterations, at most
    log_interval: int = 100       # log the residual every N sweeps (at check points)


class HipSweepBackend:
    """Sweeps through libpi_mi355.so on one GPU.  Tensors are torch-ROCm device tensors;
    launches go on torch's current stream so torch-side ops and events order with them."""

    def __init__(self, D, grid_shape, lo, hi, bins, actions, dynamics_src, device=None, order=None):
        import torch
        if not _native.available():
            raise RuntimeError("libpi_mi355.so is not available: " + _native_reason())
        if not torch.cuda.is_available():
            raise RuntimeError("no ROCm GPU visible to torch")
        self.torch = torch
        self.device = torch.device("cuda", torch.cuda.current_device() if device is None
                                   else torch.device(device).index or 0)
        self.engine = _native.Engine(D, grid_shape, lo, hi, bins, actions, device=self.device.index, order=order)
        t0 = time.perf_counter()
        self.engine.compile(dynamics_src)
        self.compile_seconds = time.perf_counter() - t0
        self.one_launch_failures = 0          # dataflow-kernel evaluations that gave up (V intact, run sweep by sweep instead)

    def _stream(self):
        return self.torch.cuda.current_stream(self.device).cuda_stream

    def to_memory(self, a):
        return self.engine.to_memory(a)

    def to_user(self, a):
        return self.engine.to_user(a)

    @staticmethod
    def _ptr(term) -> int:
        """Device pointer of the terminal mask, or NULL: `None` = "this grid has no terminal states"
        (the kernels then read no mask and, on sweeps without a residual, no old value either)."""
        return 0 if term is None else term.data_ptr()

    def prepare_mask(self, term, s_begin=None, s_end=None) -> int:
        return self.engine.prepare_mask(self._ptr(term), self._stream(), s_begin, s_end)

    def eval_begin(self, policy, term) -> int:
        return self.engine.eval_begin(policy.data_ptr(), self._ptr(term), self._stream())

    def eval_end(self) -> None:
        self.engine.eval_end()

    def eval_sweeps(self, Va, Vb, policy, term, s_begin, s_end, gamma, n_sweeps, d_delta):
        self.engine.eval_sweeps(Va.data_ptr(), Vb.data_ptr(), policy.data_ptr(), self._ptr(term),
                                s_begin, s_end, gamma, n_sweeps,
                                0 if d_delta is None else d_delta.data_ptr(), self._stream())

    @property
    def resident(self) -> bool:
        """True when the library runs a whole policy evaluation (all sweeps and residual checks) of this grid in one
        launch: the LDS-resident kernel for grids one CU holds, the dataflow kernel for launch-bound grids beyond."""
        if self.one_launch_failures >= 2 and self.engine.info(13) == 0:
            return False                                             # the dataflow kernel keeps timing out on this box
        return (self.engine.info(13) > 0 or self.engine.info(19) > 0) and self.engine.info(14) == 1

    def policy_evaluation(self, V, policy, term, gamma, theta, max_sweeps, check_interval):
        """The whole evaluation loop on the device (pi_policy_evaluation): returns (sweeps done,
        the residuals looked at, in order).  One host synchronisation for the whole evaluation."""
        torch = self.torch
        looks = max_sweeps // check_interval + 2
        out = torch.zeros(looks + 1, dtype=torch.float32, device=self.device)     # [log..., sweeps as int32 bits]
        sweeps = out[looks:].view(torch.int32)
        self.engine.policy_evaluation(V.data_ptr(), policy.data_ptr(), self._ptr(term), gamma, theta,
                                      max_sweeps, check_interval, sweeps.data_ptr(), 0, out.data_ptr(),
                                      self._stream())
        host = out.cpu()
        done = int(host[looks:].view(torch.int32).item())
        if done < 0:
            # a workgroup of the dataflow kernel gave up waiting for its peers (they must all be resident at once: another
            # stream or process holding CUs is enough).  V is untouched — the finish kernel is its only writer — so the
            # caller runs this evaluation sweep by sweep; after two such launches the kernel is not tried again.
            self.one_launch_failures += 1
            logger.warning("pi_policy_evaluation: the one-launch evaluation kernel gave up waiting for its peers "
                           f"(PI_MI355_FLOW_TIMEOUT); falling back to the sweep-by-sweep loop ({self.one_launch_failures} so far)")
            return None
        n_looks = (done - 1) // check_interval + 1 + (1 if (done - 1) % check_interval else 0)
        return done, host[:n_looks].numpy()

    @property
    def whole_run(self) -> bool:
        """True when the library runs evaluation AND improvement rounds in one launch on this grid (pi_info 34: grids
        one CU's LDS holds, and 2-D grids of up to 2^16 states beyond those); PI_MI355_WHOLE_RUN=0 keeps the
        round-by-round loop."""
        return self.resident and self.engine.info(34) > 0 and os.environ.get("PI_MI355_WHOLE_RUN", "1") != "0"

    def policy_iteration(self, V, policy, term, gamma, theta, max_eval_sweeps, check_interval, max_pi_iter, retried=False):
        """The whole run on the device (pi_policy_iteration).  Returns (rounds done, stable, [(sweeps, residual,
        entries changed) per round]) — or None when the launch could not go through (placement, a bounded wait):
        V and the policy are untouched then and the caller runs the loop itself.  One host synchronisation."""
        torch = self.torch
        out = torch.zeros(2 + 4 * max_pi_iter, dtype=torch.int32, device=self.device)
        self.engine.policy_iteration(V.data_ptr(), policy.data_ptr(), self._ptr(term), gamma, theta, max_eval_sweeps,
                                     check_interval, max_pi_iter, out.data_ptr(), out[2:].data_ptr(), self._stream())
        host = out.cpu().numpy()
        rounds = int(host[0])
        if rounds < 0:
            # the XCD-local launch did not go through (placement, a bounded wait): V and the policy are untouched.  The
            # library cannot count that itself (the call is asynchronous): report it, and where one CU's LDS holds the
            # grid run the whole-run kernel that cannot fail instead of going back to the round-by-round loop.
            if not retried and self.engine.info(13) > 0 and self.engine.info(14) == 1:
                self.engine.set_option(8, 2)
                return self.policy_iteration(V, policy, term, gamma, theta, max_eval_sweeps, check_interval, max_pi_iter,
                                             retried=True)
            if not retried:
                self.engine.set_option(8, 1)
            return None
        log = host[2:2 + 4 * rounds].reshape(rounds, 4)
        residuals = log[:, 1].copy().view(np.float32)
        return rounds, bool(host[1]), [(int(log[r, 0]), float(residuals[r]), int(log[r, 2])) for r in range(rounds)]

    def reach_planes(self, term, s_begin, s_end, n_planes, dim=0):
        """bool[n_planes]: planes of V along `dim` the states of the range can read (any action)."""
        words = (n_planes + 31) // 32
        bitmap = self.torch.zeros(words, dtype=self.torch.int32, device=self.device)
        self.engine.reach_planes(self._ptr(term), s_begin, s_end, bitmap.data_ptr(), self._stream(), dim=dim)
        bits = bitmap.cpu().numpy().view(np.uint32)
        return ((bits[np.arange(n_planes) >> 5] >> (np.arange(n_planes) & 31).astype(np.uint32)) & 1).astype(bool)

    def reach_units(self, term, s_begin, s_end, depth):
        """bool[units]: units of the leading `depth` dimensions (planes of dimension 0, or rows
        (i0, i1)) the states of the range can read under any action (pi_reach_units)."""
        n_units = int(np.prod(self.engine._shape[:depth]))
        bitmap = self.torch.zeros((n_units + 31) // 32, dtype=self.torch.int32, device=self.device)
        self.engine.reach_units(self._ptr(term), s_begin, s_end, depth, bitmap.data_ptr(), self._stream())
        bits = bitmap.cpu().numpy().view(np.uint32)
        return ((bits[np.arange(n_units) >> 5] >> (np.arange(n_units) & 31).astype(np.uint32)) & 1).astype(bool)

    def improve_sweep(self, V, policy, term, s_begin, s_end, gamma, d_changed):
        self.engine.improve_sweep(V.data_ptr(), policy.data_ptr(), self._ptr(term), s_begin, s_end,
                                  gamma, 0 if d_changed is None else d_changed.data_ptr(),
                                  self._stream())

    def value_sweep(self, V, Vnew, policy, term, s_begin, s_end, gamma, d_delta, d_changed):
        self.engine.value_sweep(V.data_ptr(), Vnew.data_ptr(), policy.data_ptr(), self._ptr(term),
                                s_begin, s_end, gamma, 0 if d_delta is None else d_delta.data_ptr(),
                                0 if d_changed is None else d_changed.data_ptr(), self._stream())

    def close(self):
        # what the XCD-local kernel did, kept past the handle (pi_info 31 - 33): evaluations run in it, how many of
        # those were run again in the placement-independent kernel, whole runs launched in it
        self.xcd_evaluations, self.xcd_fallbacks, self.whole_runs = (self.engine.info(k) for k in (31, 32, 33))
        self.engine.close()


def _native_reason() -> str:
    try:
        _native.lib()
        return ""
    except _native.NativeError as exc:
        return str(exc)


class _CudaPolicyIterationBase(abc.ABC):
    """Shared implementation; the public classes fix ``_D``."""

    _D: int = 0
    # Memory order of the dimensions on the device (None: the user's order).  MEMORY_ORDER[k] = the dimension stored as
    # memory dimension k, 0 = slowest.  Which dimensions are slow decides how far apart the corners of a successor cell
    # lie and how long a value stays useful in an XCD's L2; the best order is a property of the env's dynamics
    # (tools/dim_order_sweep.py measures it: 7-11 % on the evaluation sweeps of the big BASELINE grids).  Applied to
    # grids of at least _ORDER_MIN_STATES states; PI_MI355_ORDER=user | "0,2,1,3" overrides.  Host-side arrays
    # (value_function, policy, states_space, archives, checkpoints) are always in the user's order; the device
    # tensors (d_value_function, d_policy, d_terminal_mask) are in memory order.
    MEMORY_ORDER = None
    # The order of a SHARDED solver whose transport delivers per state (the peer-to-peer transport's fused exchange):
    # dimension 0 stays slowest — shards are slabs along it — everything else may move, the velocity that couples
    # neighbouring planes included, because the halo is then measured in pairs (i_0, i_v) wherever v lies in memory
    # (csrc/pi_push_kernels.hip: pi_reach_pairs_kernel).  None: the env's own order (what RCCL-sharded solvers keep).
    SHARDED_MEMORY_ORDER = None
    _ORDER_MIN_STATES = 1 << 22

    def __init__(self, bins_space: dict, action_space, config: CudaPIConfig | None = None, *,
                 device=None, process_group=None, transport=None) -> None:
        """
        bins_space   : dict with exactly D keys -> 1-D arrays of grid points (insertion order
                       = dimension order), e.g. {"theta": linspace(-pi, pi, 200), ...}
        action_space : 1-D array of scalar action values
        config       : CudaPIConfig
        device       : torch device of this rank (default: current CUDA device)
        process_group: torch.distributed group to shard over (default: WORLD if initialised)
        transport    : multi-rank transport (transport.py); default: the library's RCCL transport,
                       bootstrapped over `process_group`, when torch.distributed is initialised;
                       False = stay single-rank even then
        """
        if not GPU_AVAILABLE:
            raise RuntimeError(
                f"{type(self).__name__} needs libpi_mi355.so and a ROCm GPU (MI355X, gfx950): "
                + (_native_reason() or "torch.cuda.is_available() is False"))
        self.config = config or CudaPIConfig()
        self.action_space = np.ascontiguousarray(action_space, dtype=np.float32)
        self.n_actions = len(self.action_space)

        keys = list(bins_space.keys())
        assert len(keys) == self._D, (
            f"{type(self).__name__} requires exactly {self._D} state dimensions.")
        # float32 grid points: states_space[:, d] takes exactly these values (:84-87).
        self._bins = [np.asarray(bins_space[k]).astype(np.float32).ravel() for k in keys]
        self._bin_keys = keys
        self.n_states = int(np.prod([len(b) for b in self._bins], dtype=np.int64))
        self._states_space = None
        self._device_arg = device
        self._process_group = process_group
        self._transport_arg = transport
        self.stats = {"eval_sweeps": 0, "improve_sweeps": 0, "pi_iterations": 0,
                      "sweeps_per_iter": [], "eval_seconds": 0.0, "improve_seconds": 0.0}

        self._precompute_grid_metadata()
        self._order = self._choose_memory_order()
        self._allocate_tensors_and_compile()

    # ── grid ────────────────────────────────────────────────────────────────────────
    @property
    def states_space(self) -> np.ndarray:
        """(n_states, D) float32 grid nodes, row-major with the last dimension fastest
        (:84-87, :482-489, :896-904).  Built on first use: the device never needs it."""
        if self._states_space is None:
            grids = np.meshgrid(*self._bins, indexing="ij")
            self._states_space = np.column_stack([g.ravel() for g in grids]).astype(np.float32)
        return self._states_space

    @states_space.setter
    def states_space(self, value) -> None:
        self._states_space = value

    def _will_shard(self) -> bool:
        """Whether this solver is one rank of several (decided as _init_sharding decides it, without side effects)."""
        comm = self._transport_arg
        if comm is False:
            return False
        if comm is not None:
            return getattr(comm, "world", 1) > 1
        try:
            import torch.distributed as dist
            return dist.is_available() and dist.is_initialized() and dist.get_world_size(self._process_group) > 1
        except Exception:  # noqa: BLE001
            return False

    def _shards_over_p2p(self) -> bool:
        """Whether the multi-rank transport of this solver will be the peer-to-peer one (decided as _init_sharding
        decides it): the only one that can deliver a halo that is not made of contiguous rows."""
        import os
        comm = self._transport_arg
        if comm is not None and comm is not False:
            from . import transport as T
            return isinstance(comm, T.P2pTransport)
        return os.environ.get("PI_MI355_TRANSPORT", "rccl").lower() == "p2p" and os.environ.get("PI_MI355_P2P_FUSED", "1") != "0"

    def _choose_memory_order(self):
        """The class's MEMORY_ORDER on big single-rank grids; the env's own order otherwise.  Sharded solvers keep the
        env's order by default: the orders that are fastest on one GPU move the velocity that couples neighbouring planes
        (x' = x + dt x_dot) out of the second-slowest place, and then a shard's reach into its neighbours is a band of
        whole planes instead of a triangle of rows — measured with 8 logical ranks, a rank of the 80^4 grid receives
        15.6-17.6 MiB per sweep instead of 6.9-8.4 and the row-exact swept-first set is the whole shard again
        (profiles/r04/logical_ranks_c4_w8_memory_order.json).  PI_MI355_ORDER forces an order in either case.

        Round 6: a plugin nobody has tuned — a class without a MEMORY_ORDER of its own, i.e. every subclass written for
        the reference (:113-138) — gets its order MEASURED on big single-rank grids (`_tune_memory_order`: a handful of
        candidate orders derived from the plugin's own dynamics, each timed on the device; the decision is cached beside
        the code objects) instead of running in the env's order; PI_MI355_ORDER=user is the opt-out, and so is
        MEMORY_ORDER = "user" on a class whose own order has been measured to be as good as any."""
        import os
        env = os.environ.get("PI_MI355_ORDER", "").strip().lower()
        if env in ("user", "identity", "none"):
            return None
        big = self.n_states >= self._ORDER_MIN_STATES
        if env == "auto":
            order = self._tune_memory_order()
        elif env:
            order = tuple(int(v) for v in env.split(","))
        elif isinstance(self.MEMORY_ORDER, str):                       # "user": measured, the env's own order stays
            return None
        elif self.MEMORY_ORDER is not None and big and not self._will_shard():
            order = tuple(self.MEMORY_ORDER)
        elif (self.SHARDED_MEMORY_ORDER is not None and big and self._will_shard() and self._shards_over_p2p()):
            order = tuple(self.SHARDED_MEMORY_ORDER)
        elif self.MEMORY_ORDER is None and big and not self._will_shard() and self._runs_the_stock_allocation():
            order = self._tune_memory_order()
        else:
            return None
        if sorted(order) != list(range(self._D)):
            raise ValueError(f"memory order {order} is not a permutation of the {self._D} dimensions")
        return None if order == tuple(range(self._D)) else order

    def _runs_the_stock_allocation(self) -> bool:
        """A subclass that replaces _allocate_tensors_and_compile wholesale builds its own device arrays (in the env's
        order, as the reference does): it is left alone."""
        for klass in type(self).__mro__:
            if "_allocate_tensors_and_compile" in vars(klass):
                return klass is _CudaPolicyIterationBase
        return True

    def _lane_spreads(self, dev, samples: int = 192, seed: int = 0):
        """For every candidate lane dimension d: {k: mean spread, in cells, of the successor cells of a wave (up to 64
        consecutive grid points along d, everything else random) over dimension k} — from the plugin's own dynamics,
        evaluated on the device through a throw-away handle (pi_probe_step / pi_probe_interp)."""
        import torch
        D, shape = self._D, [len(b) for b in self._bins]
        eng = _native.Engine(D, self.grid_shape, self.bounds_low, self.bounds_high, self._bins, self.action_space,
                             device=dev.index)
        try:
            eng.compile(self._dynamics_cuda_src())
            rng = np.random.default_rng(seed)
            spread = {}
            for d in range(D):
                L = min(64, shape[d])
                idx = np.stack([rng.integers(0, shape[k], samples) for k in range(D)], axis=1)
                idx = np.repeat(idx[:, None, :], L, axis=1)
                idx[:, :, d] = rng.integers(0, shape[d] - L + 1, samples)[:, None] + np.arange(L)[None, :]
                pts = np.stack([self._bins[k][idx[:, :, k]] for k in range(D)], axis=-1).reshape(-1, D).astype(np.float32)
                act = np.repeat(rng.choice(self.action_space, samples), L).astype(np.float32)
                m = len(pts)
                d_pts, d_act = torch.from_numpy(pts).to(dev), torch.from_numpy(act).to(dev)
                d_next = torch.empty((m, D), dtype=torch.float32, device=dev)
                d_rew = torch.empty(m, dtype=torch.float32, device=dev)
                d_done = torch.empty(m, dtype=torch.uint8, device=dev)
                d_idx = torch.empty((m, 1 << D), dtype=torch.int32, device=dev)
                d_w = torch.empty((m, 1 << D), dtype=torch.float32, device=dev)
                st = torch.cuda.current_stream(dev).cuda_stream
                eng.probe_step(d_pts.data_ptr(), d_act.data_ptr(), d_next.data_ptr(), d_rew.data_ptr(), d_done.data_ptr(), m, st)
                eng.probe_interp(d_next.data_ptr(), d_idx.data_ptr(), d_w.data_ptr(), m, st)
                torch.cuda.synchronize(dev)
                base = d_idx[:, 0].cpu().numpy().astype(np.int64)           # lowest corner of the successor cell (env order here)
                cell = np.stack(np.unravel_index(base, shape), axis=1).reshape(samples, L, D)
                spread[d] = {k: float((cell[:, :, k].max(1) - cell[:, :, k].min(1)).mean()) for k in range(D)}
        finally:
            eng.close()
        return spread

    def _candidate_orders(self, spread):
        """A handful of memory orders worth timing, from the spreads of `_lane_spreads`.  Lane dimension: the one along
        which a wave's successors stay together — the smallest total spread over the OTHER dimensions (double cartpole:
        0 cells with x along the lanes, 0.9 with x_dot, 5-9 with an angle or an angular speed), ties to the env's own
        last dimension — and the runner-up.  Its PARTNER — the dimension a wave's
        successors spread over most, the position a lane velocity moves — is tried second-fastest (double cartpole:
        (.., x, x_dot) is worth 8-10 % over (x, .., x_dot), profiles/r04/dim_order.txt), every remaining dimension is
        tried as the slowest, and the env's own order is always among the candidates."""
        D = self._D
        total = {d: sum(v for k, v in spread[d].items() if k != d) for d in range(D)}
        ranked = sorted(range(D), key=lambda d: (total[d], -d))
        lanes = ranked[:2]                                          # D = 2: both orders
        out = [tuple(range(D))]
        for lane in lanes:
            rest = [k for k in range(D) if k != lane]
            out.append(tuple(rest + [lane]))
            if D < 3:
                continue
            # the partner: where a wave along `lane` spreads most; a wave that does not spread at all (a position along
            # the lanes: every successor moves by the same dt * velocity) is paired with the dimension that spreads
            # over IT — its velocity
            partner = max(rest, key=lambda k: (spread[lane][k], k))
            if spread[lane][partner] < 0.25:
                partner = max(rest, key=lambda k: (spread[k][lane], k))
            others = [k for k in rest if k != partner]
            for slow in others:
                out.append(tuple([slow] + [k for k in others if k != slow] + [partner, lane]))
            if D == 4:
                for slow in rest:
                    out.append(tuple([slow] + [k for k in rest if k != slow] + [lane]))
        seen, uniq = set(), []
        for o in out:
            if o not in seen:
                seen.add(o)
                uniq.append(o)
        return uniq

    def _tune_memory_order(self, sweeps: int = 3):
        """Pick the memory order of a plugin nobody has tuned by MEASURING it: the candidates of `_candidate_orders`,
        each compiled (hipRTC, a second or two) and timed on this grid — one warm-up and `sweeps` evaluation sweeps of a
        random (V, policy), the fastest sweep counts.  The decision is cached beside the code objects, keyed by the
        kernel version, the plugin's source, the grid and the action set (PI_MI355_ORDER_CACHE=0: measure again).
        Results never depend on the order; only the sweeps' speed does."""
        import hashlib
        import json
        import os
        import torch
        D, n = self._D, self.n_states
        dev = torch.device("cuda", torch.cuda.current_device() if self._device_arg is None
                           else torch.device(self._device_arg).index or 0)
        src = self._dynamics_cuda_src()
        key = hashlib.sha256()
        for part in (_native.kernel_source_hash().encode(), src.encode(), np.asarray(self.action_space, np.float32).tobytes(),
                     *[b.tobytes() for b in self._bins]):
            key.update(part)
            key.update(b"|")
        cache = _native.KERNEL_CACHE / f"order_{key.hexdigest()[:24]}.json"
        if os.environ.get("PI_MI355_ORDER_CACHE", "1") != "0" and cache.exists():
            try:
                rec = json.loads(cache.read_text())
                order = tuple(int(v) for v in rec["order"])
                if sorted(order) == list(range(D)):
                    self._order_tuning = rec
                    return order
            except (OSError, ValueError, KeyError):
                pass
        spread = self._lane_spreads(dev)
        cands = self._candidate_orders(spread)
        gen = torch.Generator(device="cpu").manual_seed(0)
        with torch.cuda.device(dev):
            d_V = torch.randn(n, generator=gen, dtype=torch.float32).to(dev)
            d_Vn = torch.empty_like(d_V)
            d_pol = torch.randint(0, self.n_actions, (n,), generator=gen, dtype=torch.int32).to(dev)
            st = torch.cuda.current_stream(dev).cuda_stream
            gamma = float(np.float32(self.config.gamma))
            table = []
            for cand in cands:
                eng = _native.Engine(D, self.grid_shape, self.bounds_low, self.bounds_high, self._bins, self.action_space,
                                     device=dev.index, order=None if cand == tuple(range(D)) else cand)
                try:
                    eng.compile(src)
                    best = float("inf")
                    for i in range(sweeps + 1):
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record()
                        eng.eval_sweep(d_V.data_ptr(), d_Vn.data_ptr(), d_pol.data_ptr(), 0, 0, n, gamma, 0, st)
                        e1.record()
                        e1.synchronize()
                        if i:
                            best = min(best, e0.elapsed_time(e1))
                finally:
                    eng.close()
                table.append({"order": list(cand), "eval_ms": best})
            del d_V, d_Vn, d_pol
        # the env's own order (first candidate) stays unless another one is at least 2 % faster
        pick = min(table, key=lambda r: r["eval_ms"])
        if pick["eval_ms"] > 0.98 * table[0]["eval_ms"]:
            pick = table[0]
        rec = {"order": pick["order"], "dims": self._bin_keys, "candidates": table,
               "lane_spread": {self._bin_keys[d]: round(sum(v for k, v in spread[d].items() if k != d), 3) for d in spread}}
        self._order_tuning = rec
        logger.info("memory order (measured): " + ", ".join(f"{tuple(r['order'])} {r['eval_ms']:.3f} ms" for r in table)
                    + f" -> {tuple(pick['order'])}")
        try:
            cache.parent.mkdir(parents=True, exist_ok=True)
            tmp = cache.with_suffix(f".tmp{os.getpid()}")
            tmp.write_text(json.dumps(rec))
            os.replace(tmp, cache)
        except OSError:
            pass
        return tuple(pick["order"])

    def _to_memory(self, a):
        """Whole-grid array in the user's order -> the device's memory order (identity without MEMORY_ORDER)."""
        return a if self._order is None else self._backend.to_memory(a)

    def _to_user(self, a):
        return a if self._order is None else self._backend.to_user(a)

    def _precompute_grid_metadata(self) -> None:
        # Same quantities as :95-109 / :497-514 / :912-937, from the bin tables instead of
        # the materialised (n, D) array (min/max/unique of a column = those of its table).
        D = self._D
        self.bounds_low = np.array([b.min() for b in self._bins], dtype=np.float32)
        self.bounds_high = np.array([b.max() for b in self._bins], dtype=np.float32)
        self.grid_shape = np.array([len(np.unique(b)) for b in self._bins], dtype=np.int32)
        for d, b in enumerate(self._bins):
            if self.grid_shape[d] != len(b):
                raise ValueError(f"dimension {d} ({self._bin_keys[d]!r}) has repeated grid points")
            if len(b) < 2:
                raise ValueError(f"dimension {d} ({self._bin_keys[d]!r}) needs at least 2 grid points")
        if self.n_states >= 2 ** 31:
            raise ValueError("n_states must be < 2^31 (flat indices are int32, as in the reference)")
        strides = np.ones(D, dtype=np.int64)
        for d in range(D - 2, -1, -1):
            strides[d] = strides[d + 1] * self.grid_shape[d + 1]
        self.strides = strides.astype(np.int32)
        self.corner_bits = np.array(list(product([0, 1], repeat=D)), dtype=np.int32)
        logger.info(f"Grid: shape={self.grid_shape.tolist()}, states={self.n_states:,}, "
                    f"actions={self.n_actions}")

    # ── plugin surface ──────────────────────────────────────────────────────────────
    @abc.abstractmethod
    def _dynamics_cuda_src(self) -> str:
        """C source defining ``__device__ void step_dynamics(...)`` with the arity for D
        (2D: s0,s1,action,*ns0,*ns1,*reward,*terminated; 4D/6D likewise).  Helper
        ``__device__`` functions and ``#define``s are allowed; the text is placed in front of
        the generic kernels and compiled for gfx950."""

    def _terminal_fn(self, states: np.ndarray):
        """(bool mask over grid nodes, scalar terminal value); default: no terminal states."""
        return np.zeros(len(states), dtype=bool), 0.0

    def _terminal_fn_axes(self, axes):
        """Optional, beyond the reference API: the same mask from the bin tables instead of the materialised (n, D)
        grid.  `axes[d]` is dimension d's float32 bin table shaped to broadcast along dimension d (an open mesh, as
        ``np.ix_`` builds it); return (bool array BROADCASTABLE to the grid's shape, scalar terminal value) — the
        expression of ``_terminal_fn`` with ``states[:, d]`` replaced by ``axes[d]``.  The solver then builds the mask
        on the device and never needs ``states_space`` (5.86 GB on the host for a 25^6 grid).  A class that only
        defines ``_terminal_fn`` — every plugin written for the reference — is served through that hook as before."""
        return NotImplemented

    def _terminal_mask_on_device(self, dev):
        """(uint8 device tensor of n states in MEMORY order or None when no node is terminal, terminal value, count)."""
        import torch
        cls = type(self)
        definer = lambda name: next(c for c in cls.__mro__ if name in c.__dict__)     # noqa: E731
        shape = [int(g) for g in self.grid_shape]
        # the bin-table hook counts only when it is at least as derived as _terminal_fn: a subclass that overrides the
        # reference hook of one of this package's envs must get ITS mask
        if issubclass(definer("_terminal_fn_axes"), definer("_terminal_fn")):
            axes = [b.reshape([-1 if k == d else 1 for k in range(self._D)]) for d, b in enumerate(self._bins)]
            got = self._terminal_fn_axes(axes)
            if got is not NotImplemented:
                small, value = got
                small = np.asarray(small, dtype=bool)
                if not small.any():
                    return None, float(value), 0
                small = small.reshape((1,) * (self._D - small.ndim) + small.shape)
                full = torch.from_numpy(np.ascontiguousarray(small)).to(dev).expand(shape)
                if self._order is not None:
                    full = full.permute(*self._order)
                full = full.contiguous().reshape(-1).to(torch.uint8)
                return full, float(value), int(full.sum().item())
        if definer("_terminal_fn") is _CudaPolicyIterationBase:
            return None, 0.0, 0
        mask, value = self._terminal_fn(self.states_space)         # the reference's hook: needs the (n, D) array
        mask = np.ascontiguousarray(mask, dtype=bool)
        if not mask.any():
            return None, float(value), 0
        full = torch.from_numpy(np.ascontiguousarray(self._to_memory(mask.view(np.uint8)))).to(dev)
        return full, float(value), int(mask.sum())

    # ── device state ────────────────────────────────────────────────────────────────
    def _allocate_tensors_and_compile(self) -> None:
        import torch
        logger.info("Allocating device tensors and compiling gfx950 kernels...")
        kw = {} if self._order is None else {"order": self._order}
        self._backend = HipSweepBackend(self._D, self.grid_shape, self.bounds_low, self.bounds_high,      # the one backend
                                        self._bins, self.action_space, self._dynamics_cuda_src(),
                                        device=self._device_arg, **kw)
        if self._order is not None:
            logger.info(f"memory order of the dimensions: {[self._bin_keys[d] for d in self._order]}")
        dev = self._backend.device
        self._init_sharding()

        n, n_pad = self.n_states, self._n_pad
        self.d_policy = torch.zeros(n_pad, dtype=torch.int32, device=dev)
        self.d_value_function = torch.zeros(n_pad, dtype=torch.float32, device=dev)
        self.d_new_value_function = torch.zeros(n_pad, dtype=torch.float32, device=dev)
        self._d_delta = torch.zeros(1, dtype=torch.float32, device=dev)
        self._d_changed = torch.zeros(1, dtype=torch.int32, device=dev)

        device_mask, terminal_value, n_terminal = self._terminal_mask_on_device(dev)
        self.d_terminal_mask = torch.zeros(n_pad, dtype=torch.uint8, device=dev)
        if device_mask is not None:
            self.d_terminal_mask[:n] = device_mask
            del device_mask
            self.d_value_function[:n].masked_fill_(self.d_terminal_mask[:n].bool(), float(terminal_value))
            logger.info(f"Terminal states: {n_terminal:,} (value={terminal_value})")
        self.d_new_value_function.copy_(self.d_value_function)
        # What the sweeps are given as the mask: the tensor, or None when no grid node is terminal — the
        # kernels then stream neither the mask nor (on sweeps without a residual) the old values.
        self._term_arg = self.d_terminal_mask if n_terminal > 0 else None
        # the mask is fixed from here on (as in the reference): the library may list the live states once
        # and visit only those in the later sweeps of an evaluation batch and in the improvement sweeps
        # (big grids whose terminal regions cut through many waves; a rank of a sharded run lists its own shard
        # only — the states its launches visit)
        if self._term_arg is not None and hasattr(self._backend, "prepare_mask"):
            if self._comm is not None:
                import os
                if getattr(self._comm, "delivers_per_state", False) and os.environ.get("PI_MI355_P2P_FUSED", "1") != "0":
                    # the fused exchange of a grid with terminal states delivers from the list sweeps: keep the list
                    # even where it fills no idle lanes
                    self._backend.engine.set_option(6, 1)
                self._backend.prepare_mask(self._term_arg, self._s_begin, self._s_end)
            else:
                self._backend.prepare_mask(self._term_arg)
        if self._comm is not None:
            self._comm.plan(self)
            if self._comm.halo_elems >= 0:
                logger.info(f"halo exchange: rank {self._rank} receives "
                            f"{self._comm.halo_elems * 4 / 2**20:.1f} MiB per sweep instead of "
                            f"{(self._n_pad - self._shard_len) * 4 / 2**20:.1f} MiB")
        logger.success("Kernels compiled. Device memory allocated.")

    def _mask_arg(self):
        """The terminal mask as the sweeps get it: the device tensor, or None when no grid node is
        terminal.  A subclass that replaces _allocate_tensors_and_compile wholesale and never sets
        `_term_arg` gets the tensor it allocated."""
        if hasattr(self, "_term_arg"):
            return self._term_arg
        return self.d_terminal_mask

    def _seed_values(self, mask: np.ndarray, value: float) -> None:
        """Set V (both Jacobi buffers) on the masked grid nodes — supported way to give goal
        cells a non-zero starting value (the crane runner does this by reaching into cupy,
        runners/overhead_crane_cuda.py:193-206)."""
        import torch
        m = torch.from_numpy(np.ascontiguousarray(self._to_memory(np.ascontiguousarray(mask, dtype=bool)))).to(
            self.d_value_function.device)
        self.d_value_function[: self.n_states][m] = float(value)
        self.d_new_value_function[: self.n_states][m] = float(value)

    # ── sharding over ranks ─────────────────────────────────────────────────────────
    def _init_sharding(self) -> None:
        """Pick the transport (None on a single rank) and this rank's contiguous state shard."""
        from . import transport as T
        comm = self._transport_arg
        if comm is False:                               # explicit single-rank solver
            comm = None
        elif comm is None:
            try:
                import torch.distributed as dist
                active = dist.is_available() and dist.is_initialized()
            except Exception:  # noqa: BLE001
                active = False
            if active and dist.get_world_size(self._process_group) > 1:
                comm = T.from_environment(self._process_group)
        self._comm = comm
        self._world, self._rank = (comm.world, comm.rank) if comm is not None else (1, 0)
        self._shard_len, self._s_begin, self._s_end = T.shard_bounds(self.n_states, self._rank, self._world)
        self._n_pad = self._shard_len * self._world
        if comm is not None:
            comm.attach(self)
            logger.info(f"rank {self._rank}/{self._world}: states [{self._s_begin:,}, {self._s_end:,})")

    # ── policy iteration ────────────────────────────────────────────────────────────
    def _evaluation_sweeps(self, n: int, gamma: float) -> None:
        """n Jacobi sweeps under the current policy; afterwards ``d_value_function`` is the
        newest iterate and ``_d_delta`` holds the residual of the last sweep (max over ranks)."""
        if self._comm is not None:
            self._comm.evaluation_sweeps(self, n, gamma)
            return
        self._backend.eval_sweeps(self.d_value_function, self.d_new_value_function, self.d_policy,
                                  self._mask_arg(), self._s_begin, self._s_end, gamma, n,
                                  self._d_delta)
        if n & 1:
            self.d_value_function, self.d_new_value_function = (
                self.d_new_value_function, self.d_value_function)

    def _improvement_sweep(self, gamma: float) -> None:
        """One greedy improvement of this rank's shard; ``_d_changed`` = entries changed (sum
        over ranks)."""
        if self._comm is not None:
            self._comm.improvement_sweep(self, gamma)
            return
        self._backend.improve_sweep(self.d_value_function, self.d_policy, self._mask_arg(),
                                    self._s_begin, self._s_end, gamma, self._d_changed)

    def policy_evaluation(self) -> float:
        """Jacobi sweeps under the current policy until the residual, looked at on sweeps
        0, 25, 50, ... and the last one, drops below theta (:300-336)."""
        cfg = self.config
        gamma = float(np.float32(cfg.gamma))
        delta = float("inf")
        t0 = time.perf_counter()
        if self._comm is None and getattr(self._backend, "resident", False) and cfg.max_eval_iter >= 1:
            delta = self._policy_evaluation_resident(gamma, t0)
            if delta is not None:
                return delta
            delta = float("inf")                                     # the launch gave up with V intact: the loop below
        i = 0
        sweeps = 0
        # the policy is fixed for the whole loop: the library may drop the states whose successor is terminal
        # from the later sweeps (single rank, grids with a live-state list; same results)
        bracket = self._comm is None and hasattr(self._backend, "eval_begin")
        if bracket:
            self._backend.eval_begin(self.d_policy, self._mask_arg())
        try:
            while i < cfg.max_eval_iter:
                check = i if i % SYNC_INTERVAL == 0 else (i // SYNC_INTERVAL + 1) * SYNC_INTERVAL
                check = min(check, cfg.max_eval_iter - 1)
                n = check - i + 1
                self._evaluation_sweeps(n, gamma)
                sweeps += n
                i = check + 1
                delta = float(self._d_delta.item())          # the one host sync per 25 sweeps
                if check % cfg.log_interval == 0:
                    logger.debug(f"  Eval iter {check:5d} | delta = {delta:.4e}")
                if delta < cfg.theta:
                    logger.success(f"  Eval converged at iter {check} | delta = {delta:.2e}")
                    break
            else:
                logger.warning(f"  Eval hit max_eval_iter={cfg.max_eval_iter} | delta = {delta:.2e}")
        finally:
            if bracket:
                self._backend.eval_end()
        self.stats["eval_sweeps"] += sweeps
        self.stats["sweeps_per_iter"].append(sweeps)
        self.stats["eval_seconds"] += time.perf_counter() - t0
        return delta

    def _policy_evaluation_resident(self, gamma: float, t0: float) -> float:
        """Small grids: the same loop, run by ONE kernel launch with V in LDS (same sweeps, same
        residuals, same V); the log lines are written afterwards from the residuals it recorded."""
        cfg = self.config
        res = self._backend.policy_evaluation(self.d_value_function, self.d_policy, self._mask_arg(), gamma,
                                              float(cfg.theta), int(cfg.max_eval_iter), SYNC_INTERVAL)
        if res is None:
            return None
        sweeps, looked = res
        delta = float(looked[-1])
        for k, r in enumerate(looked):
            check = min(k * SYNC_INTERVAL, cfg.max_eval_iter - 1)
            if check % cfg.log_interval == 0:
                logger.debug(f"  Eval iter {check:5d} | delta = {float(r):.4e}")
        last = sweeps - 1
        if delta < cfg.theta:
            logger.success(f"  Eval converged at iter {last} | delta = {delta:.2e}")
        else:
            logger.warning(f"  Eval hit max_eval_iter={cfg.max_eval_iter} | delta = {delta:.2e}")
        self._d_delta.fill_(delta)
        self.stats["eval_sweeps"] += sweeps
        self.stats["sweeps_per_iter"].append(sweeps)
        self.stats["eval_seconds"] += time.perf_counter() - t0
        return delta

    def policy_improvement(self) -> bool:
        """Greedy improvement against the latest V; True when no entry changed (:338-355)."""
        t0 = time.perf_counter()
        gamma = float(np.float32(self.config.gamma))
        self._improvement_sweep(gamma)
        changed = int(self._d_changed.item())
        self.stats["improve_sweeps"] += 1
        self.stats["last_changed"] = changed
        self.stats["improve_seconds"] += time.perf_counter() - t0
        return changed == 0

    def run(self) -> None:
        """Evaluate / improve until the policy is stable or max_pi_iter is reached (:357-370)."""
        if self._run_in_one_launch():
            self._pull_tensors_from_gpu()
            return
        for n in range(self.config.max_pi_iter):
            logger.info(f"-- PI Iteration {n + 1}/{self.config.max_pi_iter} --")
            self.policy_evaluation()
            self.stats["pi_iterations"] = n + 1
            if self.policy_improvement():
                logger.success(f"Policy Iteration converged at iteration {n + 1}.")
                self.stats["stable"] = True
                break
        else:
            logger.warning(f"Policy Iteration hit max_pi_iter={self.config.max_pi_iter}.")
            self.stats["stable"] = False
        self._pull_tensors_from_gpu()

    def _run_in_one_launch(self) -> bool:
        """Launch-bound 2-D grids (BASELINE config C2): the same loop as run(), rounds and all, in ONE kernel launch
        (pi_policy_iteration) — same sweeps per round, same V, same policy; the log lines are written afterwards.
        Only when the loop is the reference's own: a subclass that overrides policy_evaluation / policy_improvement
        gets its methods called round by round.  False: nothing was changed, run() goes on round by round."""
        cfg = self.config
        cls = type(self)
        own = (cls.policy_evaluation is _CudaPolicyIterationBase.policy_evaluation
               and cls.policy_improvement is _CudaPolicyIterationBase.policy_improvement)
        if (self._comm is not None or not own or not getattr(self._backend, "whole_run", False)
                or cfg.max_eval_iter < 1 or cfg.max_pi_iter < 1):
            return False
        t0 = time.perf_counter()
        gamma = float(np.float32(cfg.gamma))
        res = self._backend.policy_iteration(self.d_value_function, self.d_policy, self._mask_arg(), gamma, float(cfg.theta),
                                             int(cfg.max_eval_iter), SYNC_INTERVAL, int(cfg.max_pi_iter))
        if res is None:
            logger.warning("one-launch run could not be placed; running round by round")
            return False
        rounds, stable, log = res
        for n, (sweeps, delta, changed) in enumerate(log):
            logger.info(f"-- PI Iteration {n + 1}/{cfg.max_pi_iter} --")
            if delta < cfg.theta:
                logger.success(f"  Eval converged at iter {sweeps - 1} | delta = {delta:.2e}")
            else:
                logger.warning(f"  Eval hit max_eval_iter={cfg.max_eval_iter} | delta = {delta:.2e}")
            self.stats["eval_sweeps"] += sweeps
            self.stats["sweeps_per_iter"].append(sweeps)
            self.stats["improve_sweeps"] += 1
            self.stats["last_changed"] = changed
        self.stats["pi_iterations"] = rounds
        self.stats["stable"] = stable
        self._d_delta.fill_(log[-1][1])
        if stable:
            logger.success(f"Policy Iteration converged at iteration {rounds}.")
        else:
            logger.warning(f"Policy Iteration hit max_pi_iter={cfg.max_pi_iter}.")
        self.stats["eval_seconds"] += time.perf_counter() - t0
        return True

    # ── extensions beyond the reference API ─────────────
    def value_iteration(self, max_iter: int | None = None) -> float:
        """Fused value-iteration sweeps: V' = max_a Q and policy = argmax in one kernel, until the
        residual (looked at every 25 sweeps like the evaluation loop) drops below theta or
        `max_iter` (default config.max_eval_iter) sweeps are done.  Returns the last residual.
        The reference's README sketches this fused form (:790-799); its code does not have it."""
        cfg = self.config
        gamma = float(np.float32(cfg.gamma))
        limit = cfg.max_eval_iter if max_iter is None else int(max_iter)
        delta = float("inf")
        sweeps = 0
        for i in range(limit):
            check = i % SYNC_INTERVAL == 0 or i == limit - 1
            self._backend.value_sweep(self.d_value_function, self.d_new_value_function, self.d_policy,
                                      self._mask_arg(), self._s_begin, self._s_end, gamma,
                                      self._d_delta if check else None, None)
            if self._comm is not None:
                self._comm.exchange(self, self.d_new_value_function)
            self.d_value_function, self.d_new_value_function = (
                self.d_new_value_function, self.d_value_function)
            sweeps += 1
            if check:
                if self._comm is not None:
                    self._comm.all_reduce_max(self, self._d_delta)
                delta = float(self._d_delta.item())
                if delta < cfg.theta:
                    break
        self.stats["value_sweeps"] = self.stats.get("value_sweeps", 0) + sweeps
        return delta

    def debug_report(self) -> dict:
        """Checked build only (PI_MI355_DEBUG=1 in the environment when the solver was constructed): the
        index violations its sweeps found since the last report — {"violations", "kind", "where", "value"}
        (include/pi_mi355.h, pi_debug_report).  Raises on a solver whose kernels carry no checks."""
        return self._backend.engine.debug_report()

    def save_checkpoint(self, filepath) -> None:
        """Mid-run snapshot (V, policy, counters) that `load_checkpoint` can resume from; the
        reference can only save after run() has dropped its device arrays (:392-409).  Collective
        on several ranks (every rank calls it); rank 0 writes, to a temporary file that is renamed
        into place."""
        import os
        filepath = Path(filepath).with_suffix(".npz")
        n = self.n_states
        if self._comm is not None:
            self._comm.all_gather(self, self.d_policy)
            self._comm.all_gather(self, self.d_value_function)
        if self._rank == 0:
            filepath.parent.mkdir(parents=True, exist_ok=True)
            tmp = filepath.with_name(filepath.name + f".tmp{os.getpid()}.npz")
            np.savez(tmp, value_function=np.ascontiguousarray(self._to_user(self.d_value_function[:n].cpu().numpy())),
                     policy=np.ascontiguousarray(self._to_user(self.d_policy[:n].cpu().numpy())),
                     grid_shape=self.grid_shape,
                     action_space=self.action_space,
                     eval_sweeps=np.int64(self.stats["eval_sweeps"]),
                     improve_sweeps=np.int64(self.stats["improve_sweeps"]),
                     pi_iterations=np.int64(self.stats["pi_iterations"]),
                     sweeps_per_iter=np.asarray(self.stats["sweeps_per_iter"], dtype=np.int64))
            os.replace(tmp, filepath)

    def load_checkpoint(self, filepath) -> None:
        """Restore V and the policy of a `save_checkpoint` file into this (freshly constructed)
        solver; `run()` then continues from there."""
        import torch
        data = np.load(Path(filepath).with_suffix(".npz"))
        if not (np.array_equal(data["grid_shape"], self.grid_shape)
                and np.array_equal(data["action_space"], self.action_space)):
            raise ValueError("checkpoint was written for a different grid or action set")
        n, dev = self.n_states, self.d_value_function.device
        self.d_value_function[:n].copy_(torch.from_numpy(np.ascontiguousarray(self._to_memory(data["value_function"]))).to(dev))
        self.d_new_value_function.copy_(self.d_value_function)
        self.d_policy[:n].copy_(torch.from_numpy(np.ascontiguousarray(self._to_memory(data["policy"]))).to(dev))
        for key in ("eval_sweeps", "improve_sweeps", "pi_iterations"):
            self.stats[key] = int(data[key])
        if "sweeps_per_iter" in data:
            self.stats["sweeps_per_iter"] = [int(x) for x in data["sweeps_per_iter"]]

    def _pull_tensors_from_gpu(self) -> None:
        """Copy V and the policy to host arrays and drop every device array (:372-388)."""
        logger.info("Pulling results from device memory...")
        if self._comm is not None:
            self._comm.all_gather(self, self.d_policy)
            self._comm.all_gather(self, self.d_value_function)    # halo mode: V is only local + halos
        n = self.n_states
        self.value_function = np.ascontiguousarray(self._to_user(self.d_value_function[:n].cpu().numpy()))
        self.policy = np.ascontiguousarray(self._to_user(self.d_policy[:n].cpu().numpy()))
        for attr in ["d_terminal_mask", "_term_arg", "d_value_function", "d_new_value_function", "d_policy",
                     "_d_delta", "_d_changed"]:
            if hasattr(self, attr):
                delattr(self, attr)
        if self._comm is not None:
            self._comm.close()
        self._backend.close()
        logger.success("Device memory released. Results in host memory.")

    # ── persistence (schema of :392-432) ────────────────────────────────────────────
    def save(self, filepath) -> None:
        filepath = Path(filepath).with_suffix(".npz")
        filepath.parent.mkdir(parents=True, exist_ok=True)
        np.savez(filepath, value_function=self.value_function, policy=self.policy,
                 bounds_low=self.bounds_low, bounds_high=self.bounds_high,
                 grid_shape=self.grid_shape, strides=self.strides, corner_bits=self.corner_bits,
                 action_space=self.action_space, states_space=self.states_space)
        logger.success(f"Policy saved to {filepath.resolve()}")

    @classmethod
    def load(cls, filepath):
        """Rebuild an instance from a saved archive; needs neither a GPU nor the library."""
        filepath = Path(filepath).with_suffix(".npz")
        data = np.load(filepath)
        inst = cls.__new__(cls)
        inst._states_space = None
        for key in ("value_function", "policy", "bounds_low", "bounds_high", "grid_shape",
                    "strides", "corner_bits", "action_space"):
            setattr(inst, key, data[key])
        inst.states_space = data["states_space"]
        inst.n_actions = len(inst.action_space)
        inst.n_states = len(inst.states_space)
        inst.config = CudaPIConfig()
        logger.success(f"Policy loaded from {filepath.resolve()}")
        return inst


class CudaPolicyIteration2D(_CudaPolicyIterationBase):
    """2-D grids (reference class :46-432); ``step_dynamics(s0, s1, action, *ns0, *ns1,
    *reward, *terminated)``; 4-corner interpolation."""
    _D = 2


class CudaPolicyIteration4D(_CudaPolicyIterationBase):
    """4-D grids (reference class :439-840); 16-corner interpolation."""
    _D = 4


class CudaPolicyIteration6D(_CudaPolicyIterationBase):
    """6-D grids (reference class :847-1272); 64-corner interpolation."""
    _D = 6
