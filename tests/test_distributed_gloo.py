"""
Multi-rank host path on CPU: world_size 2, 3 and 4 over gloo.  On the GPU the exchange runs
inside libpi_mi355.so over RCCL (csrc/pi_comm.cpp, tested with its in-process transport in
tests/test_gpu_parity.py); here the SAME plan — the library's host-only ``pi_plan_segments`` —
is driven from Python by the test transport (``tests/dist_transport.py``).  Each rank
sweeps its contiguous state shard (sizes NOT divisible by the world size, so the padded tail
is exercised), the reachable planes (or the whole shards) travel after every evaluation sweep,
residual / changed-count are all-reduced, and the result must be bit-identical to the
single-rank run.  Everything a rank did not declare reachable is poisoned with
NaN after each exchange, so a plan that misses a plane cannot pass.  The sweep backend is the
CPU checker of tests/helpers.py (test-only hook).
"""
from __future__ import annotations

import socket
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parents[1]


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank: int, world: int, port: int, name: str, shape, cfg_kw: dict, out_dir: str,
            exchange: str = "auto") -> None:
    sys.path.insert(0, str(ROOT))
    import os
    os.environ["PI_MI355_EXCHANGE"] = exchange
    os.environ["PI_MI355_POISON_UNREACHED"] = "1"     # reads outside the planned band -> NaN
    import torch
    import torch.distributed as dist
    from dynamicprogramming_amd import envs
    from dynamicprogramming_amd.solver import CudaPIConfig
    from tests import helpers as H
    from tests.dist_transport import TorchDistTransport
    torch.set_num_threads(1)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        cls = envs.ENVS[name]
        s = H.with_checker_backend(cls)(H.env_bins_space(name, shape), cls.ACTIONS, CudaPIConfig(**cfg_kw),
                                        transport=TorchDistTransport())
        assert s._world == world and s._rank == rank
        n = s.n_states
        per = -(-n // world)
        assert (s._s_begin, s._s_end) == (min(rank * per, n), min((rank + 1) * per, n))
        s.run()
        np.savez(Path(out_dir) / f"rank{rank}.npz", V=s.value_function, policy=s.policy,
                 sweeps=np.asarray(s.stats["sweeps_per_iter"]), evals=s._backend.calls["eval"],
                 halo=np.int64(s._comm.halo_elems))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("exchange", ["allgather", "halo"])
@pytest.mark.parametrize("world,name,shape", [(2, "mountain_car", (23, 19)),      # 437 states: odd
                                               (3, "cartpole", (9, 4, 7, 3)),       # 756 = 3*252
                                               (2, "double_cartpole", (3, 2, 3, 3, 3, 3)),   # 486
                                               (4, "pendulum", (41, 13))])          # angle wraps
def test_sharded_run_is_bit_identical_to_single_rank(world, name, shape, exchange, tmp_path):
    import torch.multiprocessing as mp
    from dynamicprogramming_amd import envs
    from dynamicprogramming_amd.solver import CudaPIConfig
    from tests import helpers as H
    cfg_kw = {**envs.ENVS[name].CONFIG, "max_pi_iter": 3, "max_eval_iter": 60}
    mp.spawn(_worker, args=(world, _free_port(), name, shape, cfg_kw, str(tmp_path), exchange),
             nprocs=world, join=True)
    cls = envs.ENVS[name]
    single = H.with_checker_backend(cls)(H.env_bins_space(name, shape), cls.ACTIONS, CudaPIConfig(**cfg_kw))
    single.run()
    for r in range(world):
        got = np.load(tmp_path / f"rank{r}.npz")
        H.assert_bits_equal(got["V"], single.value_function, f"rank {r} V")
        assert np.array_equal(got["policy"], single.policy)
        assert got["sweeps"].tolist() == single.stats["sweeps_per_iter"]
        if exchange == "allgather":
            assert int(got["evals"]) == single.stats["eval_sweeps"]  # one launch per sweep per rank
        else:                                                        # boundary + interior launches
            assert int(got["evals"]) >= single.stats["eval_sweeps"]
        assert (int(got["halo"]) >= 0) == (exchange == "halo")


def _worker_vi(rank: int, world: int, port: int, name: str, shape, cfg_kw: dict, out_dir: str, exchange: str) -> None:
    sys.path.insert(0, str(ROOT))
    import os
    os.environ["PI_MI355_EXCHANGE"] = exchange
    os.environ["PI_MI355_POISON_UNREACHED"] = "1"
    import torch
    import torch.distributed as dist
    from dynamicprogramming_amd import envs
    from dynamicprogramming_amd.solver import CudaPIConfig
    from tests import helpers as H
    from tests.dist_transport import TorchDistTransport
    torch.set_num_threads(1)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        cls = envs.ENVS[name]
        make = lambda: H.with_checker_backend(cls)(H.env_bins_space(name, shape), cls.ACTIONS, CudaPIConfig(**cfg_kw),
                                                   transport=TorchDistTransport())
        s = make()
        delta = s.value_iteration(max_iter=31)               # residual looked at on sweeps 0, 25 and 30
        s.save_checkpoint(Path(out_dir) / "vi_ckpt")          # collective: rank 0 writes
        dist.barrier()
        r = make()                                            # every rank resumes from rank 0's file
        r.load_checkpoint(Path(out_dir) / "vi_ckpt")
        delta2 = r.value_iteration(max_iter=7)
        r._pull_tensors_from_gpu()
        np.savez(Path(out_dir) / f"vi_rank{rank}.npz", V=r.value_function, policy=r.policy,
                 deltas=np.asarray([delta, delta2], dtype=np.float64))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("exchange", ["allgather", "halo"])
@pytest.mark.parametrize("world,name,shape", [(2, "cartpole", (9, 4, 7, 3)), (3, "pendulum", (31, 13))])
def test_sharded_value_iteration_and_checkpoint_resume(world, name, shape, exchange, tmp_path):
    """The 'next' rows of SURVEY section 8f.4 on several ranks: fused value-iteration sweeps with the exchange
    after every sweep, a collective checkpoint written by rank 0, every rank resuming from it — bit-identical
    to one rank doing 31 + 7 sweeps without a break."""
    import torch.multiprocessing as mp
    from dynamicprogramming_amd import envs
    from dynamicprogramming_amd.solver import CudaPIConfig
    from tests import helpers as H
    cfg_kw = {**envs.ENVS[name].CONFIG, "theta": 1e-30}       # never converges early: sweep counts are exact
    mp.spawn(_worker_vi, args=(world, _free_port(), name, shape, cfg_kw, str(tmp_path), exchange),
             nprocs=world, join=True)
    cls = envs.ENVS[name]
    single = H.with_checker_backend(cls)(H.env_bins_space(name, shape), cls.ACTIONS, CudaPIConfig(**cfg_kw))
    d1 = single.value_iteration(max_iter=31)
    d2 = single.value_iteration(max_iter=7)
    single._pull_tensors_from_gpu()
    for r in range(world):
        got = np.load(tmp_path / f"vi_rank{r}.npz")
        H.assert_bits_equal(got["V"], single.value_function, f"rank {r} V")
        assert np.array_equal(got["policy"], single.policy)
        assert got["deltas"].tolist() == [d1, d2]
