"""
Test transport for the CPU suite: the library's exchange plan (the host-only ``pi_plan_segments`` of
libpi_mi355.so) driven from Python over ``torch.distributed`` with the CPU checker backend of
tests/helpers.py.  Test infrastructure only — the product's multi-rank path is
``dynamicprogramming_amd.transport.NativeTransport`` (RCCL inside the library); tests pass an
instance of this class through the solver's ``transport=`` argument.
"""
from __future__ import annotations

import os

import numpy as np

from dynamicprogramming_amd import _native


class TorchDistTransport:
    """Test transport (gloo): the library's plan, driven from Python over torch.distributed."""

    is_native = False

    def __init__(self, group=None):
        import torch.distributed as dist
        self.group = group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self.segments = None
        self.halo_elems = -1

    def attach(self, solver) -> None:
        import torch.distributed as dist
        self._peer = [dist.get_global_rank(self.group, r) if self.group is not None else r
                      for r in range(self.world)]

    def _all_gather_shards(self, solver, full) -> None:
        import torch.distributed as dist
        per = solver._shard_len
        mine = full[self.rank * per:(self.rank + 1) * per].clone()   # gloo: no aliasing
        dist.all_gather_into_tensor(full, mine, group=self.group)

    def plan(self, solver) -> None:
        import torch
        import torch.distributed as dist
        mode = os.environ.get("PI_MI355_EXCHANGE", "auto")
        self.segments = self._send_ranges = self._interior = self._poison = None
        if mode == "allgather":
            return
        n, per = solver.n_states, solver._shard_len
        # same units as the library: rows (i0, i1) where the grid has them, planes otherwise
        shape = [int(x) for x in solver.grid_shape]
        depth = 2 if (len(shape) >= 3 and shape[0] * shape[1] <= (1 << 17)) else 1
        depth = max(1, min(depth, int(os.environ.get("PI_MI355_REACH_DEPTH", depth))))
        g0 = int(np.prod(shape[:depth]))
        stride0 = n // g0
        mine = np.zeros(g0, dtype=bool)
        if solver._s_end > solver._s_begin:
            mine = solver._backend.reach_units(solver._mask_arg(), solver._s_begin, solver._s_end, depth)
        allbits = torch.zeros(self.world * g0, dtype=torch.uint8)
        dist.all_gather_into_tensor(allbits, torch.from_numpy(mine.astype(np.uint8)), group=self.group)
        reach = allbits.numpy().astype(bool).reshape(self.world, g0)
        segs = _native.plan_segments(self.world, g0, stride0, n, per, reach)    # the C++ planner
        recv = [int(sum(b - a for (_, d, a, b) in segs if d == r)) for r in range(self.world)]
        if mode != "halo" and max(recv) > 0.6 * per * (self.world - 1):
            return
        self.segments = [tuple(int(x) for x in s) for s in segs]
        self.halo_elems = recv[self.rank]
        cuts = sorted({(a, b) for (src, _, a, b) in self.segments if src == self.rank})
        merged = []
        for a, b in cuts:
            if merged and a <= merged[-1][1]:
                merged[-1][1] = max(merged[-1][1], b)
            else:
                merged.append([a, b])
        interior, pos = [], solver._s_begin
        for a, b in merged:
            if a > pos:
                interior.append((pos, a))
            pos = max(pos, b)
        if pos < solver._s_end:
            interior.append((pos, solver._s_end))
        self._send_ranges, self._interior = [tuple(m) for m in merged], interior
        # Debug aid: PI_MI355_POISON_UNREACHED=1 overwrites, after every exchange, all of V' this
        # rank neither owns nor declared reachable with NaN — a read outside the planned band then
        # poisons the result instead of silently using stale data.
        if os.environ.get("PI_MI355_POISON_UNREACHED") == "1":
            keep = torch.zeros(solver._n_pad, dtype=torch.bool)
            keep[solver._s_begin:solver._s_end] = True
            for p in np.flatnonzero(mine):
                keep[int(p) * stride0:min((int(p) + 1) * stride0, n)] = True
            self._poison = ~keep

    def _post(self, full):
        import torch.distributed as dist
        ops = []
        for src, dst, a, b in self.segments:
            if src == self.rank:
                ops.append(dist.P2POp(dist.isend, full[a:b], self._peer[dst], self.group))
            elif dst == self.rank:
                ops.append(dist.P2POp(dist.irecv, full[a:b], self._peer[src], self.group))
        return dist.batch_isend_irecv(ops) if ops else []

    def exchange(self, solver, full) -> None:
        if self.segments is None:
            self._all_gather_shards(solver, full)
            return
        for req in self._post(full):
            req.wait()
        if self._poison is not None and full.dtype.is_floating_point:
            full[self._poison] = float("nan")

    def evaluation_sweeps(self, solver, n: int, gamma: float) -> None:
        import torch
        be = solver._backend
        for k in range(n):
            last = k == n - 1
            src, dst = solver.d_value_function, solver.d_new_value_function
            if self.segments is not None:
                parts = torch.zeros(len(self._send_ranges) + len(self._interior) + 1)
                i = 0
                for a, b in self._send_ranges:
                    be.eval_sweeps(src, dst, solver.d_policy, solver._mask_arg(), a, b, gamma, 1,
                                   parts[i:i + 1] if last else None)
                    i += 1
                reqs = self._post(dst)
                for a, b in self._interior:
                    be.eval_sweeps(src, dst, solver.d_policy, solver._mask_arg(), a, b, gamma, 1,
                                   parts[i:i + 1] if last else None)
                    i += 1
                for req in reqs:
                    req.wait()
                if self._poison is not None:
                    dst[self._poison] = float("nan")
                if last:
                    solver._d_delta[0] = parts[:i].max() if i else 0.0
            else:
                be.eval_sweeps(src, dst, solver.d_policy, solver._mask_arg(), solver._s_begin,
                               solver._s_end, gamma, 1, solver._d_delta if last else None)
                self._all_gather_shards(solver, dst)
            solver.d_value_function, solver.d_new_value_function = dst, src
        self.all_reduce_max(solver, solver._d_delta)

    def improvement_sweep(self, solver, gamma: float) -> None:
        import torch.distributed as dist
        solver._backend.improve_sweep(solver.d_value_function, solver.d_policy, solver._mask_arg(),
                                      solver._s_begin, solver._s_end, gamma, solver._d_changed)
        dist.all_reduce(solver._d_changed, op=dist.ReduceOp.SUM, group=self.group)

    def all_gather(self, solver, full) -> None:
        self._all_gather_shards(solver, full)

    def all_reduce_max(self, solver, t) -> None:
        import torch.distributed as dist
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)

    def close(self) -> None:
        pass
