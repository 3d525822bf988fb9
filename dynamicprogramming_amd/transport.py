"""
Multi-rank transports of the policy-iteration solver.

Partition : rank r owns the contiguous state range
``[r * per, min((r + 1) * per, n))``, ``per = ceil(n / world)``; every rank keeps full-size V
buffers of ``per * world`` floats, sweeps its shard and then makes the new values visible where
the other ranks will read them — a halo exchange of exactly the runs of rows (i0, i1) — planes of
dimension 0 on 2-D grids — each peer can reach, or an all-gather of the shards when those bands cover most of the grid anyway.

* ``NativeTransport`` — the product path.  Everything after construction happens inside
  libpi_mi355.so (csrc/pi_comm.cpp): RCCL communicator, exchange plan, the 25-sweep batch with
  the exchange overlapped on a second HIP stream, scalar all-reduces.  Python only hands over
  the 128-byte RCCL id (bootstrap) and device pointers.  The same C++ driver also runs over an
  in-process transport (``NativeTransport.local``) so that its stream ordering can be tested
  with real kernels on one GPU.

* ``P2pTransport`` — the same C++ driver over the library's peer-to-peer transport (csrc/pi_p2p.cpp): halo rows are
  stored by the sending GPU straight into the receiver's buffers (HIP IPC mappings over xGMI), hand-shaken through
  flag pages; no RCCL communicator.  Python hands the 512-byte descriptors around (torch.distributed, any backend).
  Chosen with ``PI_MI355_TRANSPORT=p2p`` (default ``rccl``) or by passing the transport to the solver.

The CPU test-suite drives the same plan (the library's host-only ``pi_plan_segments``) from Python
with a transport of its own (tests/dist_transport.py); it is not part of this package.
"""
from __future__ import annotations

import os

import numpy as np

from . import _native


def shard_bounds(n: int, rank: int, world: int):
    per = -(-n // world)              # ceil: equal shards, the tail is padding
    s_begin = min(rank * per, n)
    return per, s_begin, min(s_begin + per, n)


class NativeTransport:
    """RCCL (or the in-process test transport) behind the C ABI."""

    is_native = True

    def __init__(self, rank: int, world: int, unique_id: bytes | None = None,
                 local_group: str | None = None):
        self.rank, self.world = int(rank), int(world)
        self._unique_id, self._local_group = unique_id, local_group
        self.info = None

    @classmethod
    def from_torch_distributed(cls, group=None):
        """Bootstrap: rank 0 creates the RCCL id, torch.distributed hands it to the others."""
        import torch.distributed as dist
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        box = [_native.comm_unique_id() if rank == 0 else None]
        src = dist.get_global_rank(group, 0) if group is not None else 0
        dist.broadcast_object_list(box, src=src, group=group)
        return cls(rank, world, unique_id=box[0])

    @classmethod
    def local(cls, rank: int, world: int, group_name: str):
        return cls(rank, world, local_group=group_name)

    # -- wiring ----------------------------------------------------------------
    def attach(self, solver) -> None:
        self.engine = solver._backend.engine
        self._stream = solver._backend._stream
        if self._local_group is not None:
            self.engine.comm_init_local(self.rank, self.world, self._local_group)
        else:
            self.engine.comm_init(self.rank, self.world, self._unique_id)

    def plan(self, solver) -> None:
        mode = {"auto": 0, "allgather": 1, "halo": 2}[os.environ.get("PI_MI355_EXCHANGE", "auto")]
        overlap = os.environ.get("PI_MI355_OVERLAP", "1") != "0"
        self.info = self.engine.exchange_plan(solver._backend._ptr(solver._mask_arg()), solver._shard_len,
                                              mode, overlap, self._stream())
        self.halo_elems = self.info["recv_elems"] if self.info["mode"] == "halo" else -1

    # -- sweeps ----------------------------------------------------------------
    def evaluation_sweeps(self, solver, n: int, gamma: float) -> None:
        self.engine.eval_sweeps_sharded(solver.d_value_function.data_ptr(),
                                        solver.d_new_value_function.data_ptr(),
                                        solver.d_policy.data_ptr(), solver._backend._ptr(solver._mask_arg()),
                                        gamma, n, solver._d_delta.data_ptr(), self._stream())
        if n & 1:
            solver.d_value_function, solver.d_new_value_function = (
                solver.d_new_value_function, solver.d_value_function)

    def improvement_sweep(self, solver, gamma: float) -> None:
        self.engine.improve_sweep_sharded(solver.d_value_function.data_ptr(), solver.d_policy.data_ptr(),
                                          solver._backend._ptr(solver._mask_arg()), gamma,
                                          solver._d_changed.data_ptr(), self._stream())

    def exchange(self, solver, full) -> None:
        self.engine.exchange_V(full.data_ptr(), self._stream())

    def all_gather(self, solver, full) -> None:
        if full.dtype.is_floating_point:
            self.engine.allgather_V(full.data_ptr(), solver._shard_len, self._stream())
        else:
            self.engine.allgather_policy(full.data_ptr(), solver._shard_len, self._stream())

    def all_reduce_max(self, solver, t) -> None:
        self.engine.allreduce_max_f32(t.data_ptr(), self._stream())

    def close(self) -> None:
        pass                     # the communicator dies with the engine handle


class P2pTransport(NativeTransport):
    """Peer-to-peer stores into IPC-mapped buffers behind the C ABI (one process per rank; csrc/pi_p2p.cpp).

    The transport registers the solver's two V buffers and its policy array, so it can only be wired once they exist:
    `attach` remembers the engine, `plan` — which the solver calls after allocating — describes, exchanges the
    descriptors through `exchange` (a callable: bytes -> list of every rank's bytes, ordered by rank; by default
    torch.distributed's all_gather_object on `group`) and initialises the communicator before it makes the plan."""

    delivers_per_state = True          # fused exchange: the sweeps store their rows into the peers themselves

    def __init__(self, rank: int, world: int, exchange=None, group=None):
        super().__init__(rank, world)
        self._exchange = exchange
        self._group = group
        self._connected = False

    @classmethod
    def from_torch_distributed(cls, group=None):
        import torch.distributed as dist
        return cls(dist.get_rank(group), dist.get_world_size(group), group=group)

    def attach(self, solver) -> None:
        self.engine = solver._backend.engine
        self._stream = solver._backend._stream

    def _gather(self, mine: bytes):
        if self._exchange is not None:
            return list(self._exchange(mine))
        import torch.distributed as dist
        box = [None] * self.world
        dist.all_gather_object(box, mine, group=self._group)
        return box

    def plan(self, solver) -> None:
        if not self._connected:
            bufs = [(t.data_ptr(), t.numel() * t.element_size())
                    for t in (solver.d_value_function, solver.d_new_value_function, solver.d_policy)]
            mine = self.engine.p2p_describe(self.rank, self.world, bufs)
            everyone = self._gather(mine)
            status = b"mapped"
            try:
                self.engine.comm_init_p2p(self.rank, self.world, everyone)
            except _native.NativeError as exc:       # tell the peers instead of leaving them in the next collective
                status = f"rank {self.rank}: {exc}".encode()
            # nobody stores into a peer before every rank has mapped its peers — and a rank that could not map fails all
            failed = [s.decode(errors="replace") for s in self._gather(status) if s != b"mapped"]
            if failed:
                raise _native.NativeError("peer-to-peer transport could not be set up: " + "; ".join(failed))
            self._connected = True
        super().plan(solver)


def from_environment(group=None):
    """The transport a sharded solver uses when it is given none: PI_MI355_TRANSPORT = rccl (default) | p2p."""
    kind = os.environ.get("PI_MI355_TRANSPORT", "rccl").lower()
    if kind == "p2p":
        return P2pTransport.from_torch_distributed(group)
    if kind != "rccl":
        raise ValueError(f"PI_MI355_TRANSPORT={kind!r}: expected 'rccl' or 'p2p'")
    return NativeTransport.from_torch_distributed(group)
