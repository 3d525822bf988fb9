"""
The peer-to-peer transport (csrc/pi_p2p.cpp, csrc/pi_p2p_kernels.hip) on the GPU: one PROCESS per rank, as in a real
multi-GPU run — here every rank's process uses the box's one GPU, which is all HIP IPC needs: each rank maps its peers'
V buffers, policy array and flag page with hipIpcOpenMemHandle, halo rows are stored by the sender's kernel straight
into the receiver's buffer, and the two sides hand-shake through counters in the flag pages; the scalar reductions go
through the same pages.  No RCCL communicator exists in these runs (RCCL refuses two ranks on one GPU anyway).

Bar: every rank's full run() equals the single-rank run bit for bit — V, policy, sweep counts — over the halo plan
(coarse ranges, row-exact lists, with and without overlap) and the all-gather, 2-D / 4-D / 6-D, with and without
terminal states, shards whose borders are not line-aligned.  And a peer that never arrives is an error within
PI_MI355_COMM_TIMEOUT, not a hang.

No reference counterpart: src/cuda_policy_iteration.py:300-336 is a single-device loop.
"""
from __future__ import annotations

import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parents[1]
pytestmark = pytest.mark.gpu


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank: int, world: int, port: int, name: str, shape, cfg_kw: dict, out_dir: str, env: dict) -> None:
    sys.path.insert(0, str(ROOT))
    os.environ.update(env)
    os.environ["PI_MI355_TRANSPORT"] = "p2p"
    os.environ.setdefault("PI_MI355_COMM_TIMEOUT", "30")
    import torch
    import torch.distributed as dist
    from dynamicprogramming_amd import envs
    from dynamicprogramming_amd.solver import CudaPIConfig
    from tests import helpers as H
    torch.set_num_threads(1)
    import datetime
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world,
                            timeout=datetime.timedelta(seconds=120))       # a rank that dies fails its peers, in time
    try:
        cls = envs.ENVS[name]
        s = cls(H.env_bins_space(name, shape), cls.ACTIONS, CudaPIConfig(**cfg_kw), device="cuda:0")
        assert s._world == world and s._rank == rank
        eng = s._backend.engine
        assert eng.comm_info(2) == 3, "the solver is not on the peer-to-peer transport"
        info = dict(s._comm.info)
        row_exact, fused, live, pairs = eng.comm_info(5), eng.comm_info(6), eng.info(16), eng.comm_info(7)   # before run()
        if env.get("TEST_POISON") == "1":
            # Everything this rank neither owns nor is DELIVERED becomes NaN: both Jacobi buffers outside its shard are
            # poisoned, then its peers deliver the starting values it can reach (one whole-row exchange).  From here on a
            # destination mask that misses a state some sweep reads — under any policy the run visits — puts a NaN into V
            # and the comparison with the single-rank run fails.  (A cell's corners are delivered whatever their weights:
            # the reach probes mark cells, not weights.)
            nan = float("nan")
            for buf in (s.d_value_function, s.d_new_value_function):
                buf[: s._s_begin] = nan
                buf[s._s_end:] = nan
            torch.cuda.synchronize()
            dist.barrier()                      # nobody delivers into a buffer that is still being poisoned
            s._comm.exchange(s, s.d_value_function)
            torch.cuda.synchronize()
            dist.barrier()
        s.run()
        np.savez(Path(out_dir) / f"rank{rank}.npz", V=s.value_function, policy=s.policy,
                 sweeps=np.asarray(s.stats["sweeps_per_iter"]), mode=np.asarray(info["mode"]),
                 recv=np.int64(info["recv_elems"]), row_exact=np.int64(row_exact), fused=np.int64(fused), live=np.int64(live), pairs=np.int64(pairs),
                 send=np.int64(info["send_elems"]), order=np.asarray(eng.order))
        dist.barrier()
    finally:
        dist.destroy_process_group()


CASES = [
    # world, env, grid, exchange mode, extra environment
    (2, "pendulum", (41, 13), "halo", {}),
    (3, "double_pendulum_swingup", (14, 9, 11, 8), "halo", {}),                       # no terminal states: row-exact lists
    (3, "double_pendulum_swingup", (14, 9, 11, 8), "halo", {"PI_MI355_ROW_EXACT": "0"}),
    # row-exact plan forced: the FUSED exchange (pi_eval_push_kernel stores the rows into the peers itself) ...
    (3, "double_pendulum_swingup", (14, 9, 11, 8), "halo", {"PI_MI355_ROW_EXACT": "1"}),
    (4, "double_pendulum_swingup", (40, 6, 8, 6), "halo", {"PI_MI355_ROW_EXACT": "1"}),   # C4 @ 8 in small, 4 ranks
    (2, "pendulum", (41, 13), "halo", {"PI_MI355_ROW_EXACT": "1"}),
    # ... and the same plan with the copy kernel on the second stream
    (3, "double_pendulum_swingup", (14, 9, 11, 8), "halo", {"PI_MI355_ROW_EXACT": "1", "PI_MI355_P2P_FUSED": "0"}),
    (2, "double_cartpole", (6, 4, 5, 4, 5, 4), "halo", {}),                            # terminal states, 6-D
    # the velocity that couples neighbouring planes moved out of memory dimension 1 (what the fast single-GPU orders do):
    # rows (i0, i1) are then all reachable, and the fused exchange cuts its destination masks down to the pairs
    # (i_0, i_v) each peer really reads (pi_reach_pairs_kernel) — state-exact lists, delivered per state
    (3, "double_pendulum_swingup", (14, 9, 11, 8), "halo", {"PI_MI355_ORDER": "0,2,1,3"}),
    (4, "double_pendulum_swingup", (40, 6, 8, 6), "halo", {"PI_MI355_ORDER": "0,2,3,1"}),
    (2, "double_cartpole", (6, 4, 5, 4, 5, 4), "halo", {"PI_MI355_ORDER": "0,2,3,5,4,1", "PI_MI355_LIVE_MIN": "1"}),
    (2, "double_cartpole", (6, 4, 5, 4, 5, 4), "halo", {"PI_MI355_ORDER": "0,2,3,5,4,1", "PI_MI355_LIVE_MIN": "1",
                                                       "PI_MI355_PAIR_REACH": "0"}),
    # the same plans with everything a rank is not delivered POISONED (NaN): the masks are sufficient, not just plausible
    (3, "double_pendulum_swingup", (14, 9, 11, 8), "halo", {"PI_MI355_ROW_EXACT": "1", "TEST_POISON": "1"}),
    (3, "double_pendulum_swingup", (14, 9, 11, 8), "halo", {"PI_MI355_ORDER": "0,2,1,3", "TEST_POISON": "1"}),
    (4, "double_pendulum_swingup", (40, 6, 8, 6), "halo", {"PI_MI355_ORDER": "0,2,3,1", "TEST_POISON": "1"}),
    (2, "double_cartpole", (6, 4, 5, 4, 5, 4), "halo", {"PI_MI355_ORDER": "0,2,3,5,4,1", "PI_MI355_LIVE_MIN": "1",
                                                       "TEST_POISON": "1"}),
    (3, "cartpole_swingup", (18, 7, 9, 8), "halo", {"PI_MI355_LIVE_MIN": "1", "TEST_POISON": "1"}),
    (4, "cartpole_swingup", (18, 7, 9, 8), "halo", {"PI_MI355_OVERLAP": "0", "TEST_POISON": "1"}),     # the copy kernel
    # grids WITH terminal states whose shards keep a live-state list: the later sweeps of every batch go through the
    # fused exchange (push kernel over the live spans of the ranges peers wait for), the first one through the copy kernel
    (2, "double_cartpole", (6, 4, 5, 4, 5, 4), "halo", {"PI_MI355_LIVE_MIN": "1"}),
    (3, "cartpole_swingup", (18, 7, 9, 8), "halo", {"PI_MI355_LIVE_MIN": "1"}),
    (3, "cartpole_swingup", (18, 7, 9, 8), "halo", {"PI_MI355_LIVE_MIN": "1", "PI_MI355_P2P_FUSED": "0"}),
    (4, "cartpole_swingup", (18, 7, 9, 8), "halo", {"PI_MI355_OVERLAP": "0"}),
    (4, "cartpole_swingup", (18, 7, 9, 8), "allgather", {}),
    (2, "mountain_car", (23, 19), "allgather", {}),                                    # 437 states: padded tail
]


@pytest.mark.parametrize("world,name,shape,mode,extra", CASES)
def test_p2p_sharded_run_is_bit_identical_to_single_rank(world, name, shape, mode, extra, cuda_device, tmp_path):
    import torch.multiprocessing as mp
    from dynamicprogramming_amd import envs
    from tests import helpers as H
    cls = envs.ENVS[name]
    cfg_kw = {**cls.CONFIG, "max_pi_iter": 3, "max_eval_iter": 60}
    single = cls(H.env_bins_space(name, shape), cls.ACTIONS, envs.CudaPIConfig(**cfg_kw), device=cuda_device,
                 transport=False)
    single.run()
    env = {"PI_MI355_EXCHANGE": mode, **extra}
    mp.spawn(_worker, args=(world, _free_port(), name, shape, cfg_kw, str(tmp_path), env), nprocs=world, join=True)
    for r in range(world):
        got = np.load(tmp_path / f"rank{r}.npz")
        H.assert_bits_equal(got["V"], single.value_function, f"rank {r} V")
        assert np.array_equal(got["policy"], single.policy)
        assert got["sweeps"].tolist() == single.stats["sweeps_per_iter"]
        assert str(got["mode"]) == mode
        if mode == "halo":
            assert 0 < int(got["recv"]) < (world - 1) * -(-single.n_states // world)
        if "PI_MI355_ORDER" in extra:
            assert got["order"].tolist() == [int(v) for v in extra["PI_MI355_ORDER"].split(",")]
            assert int(got["pairs"]) == (0 if extra.get("PI_MI355_PAIR_REACH") == "0" else 1)
            if "PI_MI355_LIVE_MIN" not in extra:
                assert int(got["row_exact"]) == 1 and int(got["fused"]) == 1      # state-exact lists, fused
        elif extra.get("PI_MI355_ROW_EXACT") == "1":
            assert int(got["row_exact"]) == 1
            assert int(got["fused"]) == (0 if extra.get("PI_MI355_P2P_FUSED") == "0" else 1)
        elif extra.get("PI_MI355_LIVE_MIN") == "1":
            # every rank decides for itself whether its shard keeps a list (>= 3 % idle lanes otherwise); ranks with and
            # without one — fused and unfused — interoperate: same message numbers
            assert int(got["fused"]) == (1 if int(got["live"]) > 0 and extra.get("PI_MI355_P2P_FUSED") != "0" else 0)
        elif extra.get("PI_MI355_ROW_EXACT") == "0" or mode != "halo" or extra.get("PI_MI355_OVERLAP") == "0":
            assert int(got["fused"]) == 0
    if extra.get("PI_MI355_LIVE_MIN") == "1":
        lives = [int(np.load(tmp_path / f"rank{r}.npz")["live"]) for r in range(world)]
        assert max(lives) > 0, "no shard kept a live-state list: the case does not test what it is for"


def _worker_lonely(rank: int, world: int, port: int, out_dir: str) -> None:
    """Rank 1 connects and then never sweeps; rank 0 must get an error, not hang."""
    sys.path.insert(0, str(ROOT))
    os.environ["PI_MI355_TRANSPORT"] = "p2p"
    os.environ["PI_MI355_COMM_TIMEOUT"] = "3"
    os.environ["PI_MI355_EXCHANGE"] = "halo"
    import time
    import torch
    import torch.distributed as dist
    from dynamicprogramming_amd import _native, envs
    from dynamicprogramming_amd.solver import CudaPIConfig
    from tests import helpers as H
    import datetime
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world,
                            timeout=datetime.timedelta(seconds=120))
    try:
        name, shape = "cartpole_swingup", (18, 7, 9, 8)
        cls = envs.ENVS[name]
        s = cls(H.env_bins_space(name, shape), cls.ACTIONS, CudaPIConfig(**cls.CONFIG), device="cuda:0")
        outcome = "idle"
        if rank == 0:
            t0 = time.time()
            try:
                s._evaluation_sweeps(4, 0.99)
                torch.cuda.synchronize()
                outcome = "no error"
            except _native.NativeError as exc:
                outcome = f"error after {time.time() - t0:.1f} s: {exc}"
            try:                                            # the communicator is spent: the next call says so at once
                s._evaluation_sweeps(1, 0.99)
                outcome += " | second call: no error"
            except _native.NativeError as exc:
                outcome += f" | second call: {exc}"
            (Path(out_dir) / "rank0.txt").write_text(outcome)
        dist.barrier()                                      # rank 1 keeps its buffers mapped until rank 0 is done
    finally:
        dist.destroy_process_group()


def test_p2p_missing_peer_is_an_error_not_a_hang(cuda_device, tmp_path):
    import torch.multiprocessing as mp
    mp.spawn(_worker_lonely, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    text = (tmp_path / "rank0.txt").read_text()
    assert text.startswith("error after"), text
    assert "gave up waiting for a peer" in text
    seconds = float(text.split("error after ")[1].split(" s")[0])
    assert 2.0 < seconds < 30.0, text
    assert "second call: " in text and "second call: no error" not in text


def test_p2p_descriptor_validation(cuda_device):
    """pi_p2p_describe / pi_comm_init_p2p refuse what they cannot serve, with messages."""
    torch = pytest.importorskip("torch")
    from dynamicprogramming_amd import _native, envs
    from tests import helpers as H
    name, shape = "pendulum", (21, 11)
    cls = envs.ENVS[name]
    s = cls(H.env_bins_space(name, shape), cls.ACTIONS, envs.CudaPIConfig(**cls.CONFIG), device=cuda_device, transport=False)
    eng = s._backend.engine
    bufs = [(t.data_ptr(), t.numel() * t.element_size()) for t in (s.d_value_function, s.d_new_value_function, s.d_policy)]
    with pytest.raises(_native.NativeError, match="call pi_p2p_describe first"):
        eng.comm_init_p2p(0, 2, [b"\0" * 512] * 2)
    with pytest.raises(_native.NativeError, match="2 <= world <= 16"):
        eng.p2p_describe(0, 1, bufs)
    with pytest.raises(_native.NativeError, match="at most 4 buffers"):
        eng.p2p_describe(0, 2, bufs + bufs)
    mine = eng.p2p_describe(0, 2, bufs)
    assert len(mine) == 512
    with pytest.raises(_native.NativeError, match="live in one process"):
        other = bytearray(mine)
        other[8:12] = (1).to_bytes(4, "little")             # rank field of a copy of this rank's own descriptor
        eng.comm_init_p2p(0, 2, [mine, bytes(other)])
    mine = eng.p2p_describe(0, 2, bufs)
    with pytest.raises(_native.NativeError, match="not one of pi_p2p_describe"):
        eng.comm_init_p2p(0, 2, [mine, b"\0" * 512])
    assert eng.comm_info(0) == -1                           # no communicator was installed by the failed attempts


def _worker_vi(rank: int, world: int, port: int, name: str, shape, cfg_kw: dict, out_dir: str, exchange: str) -> None:
    sys.path.insert(0, str(ROOT))
    os.environ.update({"PI_MI355_TRANSPORT": "p2p", "PI_MI355_COMM_TIMEOUT": "30", "PI_MI355_EXCHANGE": exchange})
    import datetime
    import torch.distributed as dist
    from dynamicprogramming_amd import envs
    from dynamicprogramming_amd.solver import CudaPIConfig
    from tests import helpers as H
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world,
                            timeout=datetime.timedelta(seconds=120))
    try:
        cls = envs.ENVS[name]
        make = lambda: cls(H.env_bins_space(name, shape), cls.ACTIONS, CudaPIConfig(**cfg_kw), device="cuda:0")  # noqa: E731
        s = make()
        d1 = s.value_iteration(max_iter=31)                  # residual looked at on sweeps 0, 25 and 30
        s.save_checkpoint(Path(out_dir) / "vi_ckpt")         # collective (all-gathers over the transport); rank 0 writes
        dist.barrier()
        r = make()                                           # a SECOND communicator of this process: new flag page, new maps
        r.load_checkpoint(Path(out_dir) / "vi_ckpt")
        d2 = r.value_iteration(max_iter=7)
        r._pull_tensors_from_gpu()
        np.savez(Path(out_dir) / f"vi_rank{rank}.npz", V=r.value_function, policy=r.policy,
                 deltas=np.asarray([d1, d2], dtype=np.float64))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("exchange", ["halo", "allgather"])
def test_p2p_sharded_value_iteration_checkpoint_and_resume(exchange, cuda_device, tmp_path):
    """The fused max-backup sweeps with the exchange after every sweep, a collective checkpoint and
    a resume in a second solver of the same processes — all over the peer-to-peer transport, bit-identical to one rank
    doing 31 + 7 sweeps without a break."""
    import torch.multiprocessing as mp
    from dynamicprogramming_amd import envs
    from tests import helpers as H
    world, name, shape = 3, "cartpole", (9, 4, 7, 3)
    cls = envs.ENVS[name]
    cfg_kw = {**cls.CONFIG, "theta": 1e-30}                  # never converges early: sweep counts are exact
    mp.spawn(_worker_vi, args=(world, _free_port(), name, shape, cfg_kw, str(tmp_path), exchange), nprocs=world, join=True)
    single = cls(H.env_bins_space(name, shape), cls.ACTIONS, envs.CudaPIConfig(**cfg_kw), device=cuda_device, transport=False)
    d1 = single.value_iteration(max_iter=31)
    d2 = single.value_iteration(max_iter=7)
    single._pull_tensors_from_gpu()
    for r in range(world):
        got = np.load(tmp_path / f"vi_rank{r}.npz")
        H.assert_bits_equal(got["V"], single.value_function, f"rank {r} V")
        assert np.array_equal(got["policy"], single.policy)
        assert got["deltas"].tolist() == [float(d1), float(d2)]


@pytest.mark.parametrize("world,name,shape,extra", [
    (3, "double_pendulum_swingup", (14, 9, 11, 8), {"PI_MI355_ORDER": "0,2,1,3"}),
    (2, "double_cartpole", (6, 4, 5, 4, 5, 4), {"PI_MI355_ORDER": "0,2,3,5,4,1", "PI_MI355_LIVE_MIN": "1"}),
])
def test_p2p_poisoning_check_fails_when_a_delivery_is_missing(world, name, shape, extra, cuda_device, tmp_path):
    """Negative control of the TEST_POISON cases above: with every 7th destination mask cleared (fault injection in the
    planner: the state is swept, its value is not delivered) the poisoned runs must NOT equal the single-rank run."""
    import torch.multiprocessing as mp
    from dynamicprogramming_amd import envs
    from tests import helpers as H
    cls = envs.ENVS[name]
    cfg_kw = {**cls.CONFIG, "max_pi_iter": 3, "max_eval_iter": 60}
    single = cls(H.env_bins_space(name, shape), cls.ACTIONS, envs.CudaPIConfig(**cfg_kw), device=cuda_device, transport=False)
    single.run()
    env = {"PI_MI355_EXCHANGE": "halo", "TEST_POISON": "1", "PI_MI355_DEBUG_DROP_DELIVERY": "7", **extra}
    mp.spawn(_worker, args=(world, _free_port(), name, shape, cfg_kw, str(tmp_path), env), nprocs=world, join=True)
    differs = False
    for r in range(world):
        got = np.load(tmp_path / f"rank{r}.npz")
        assert int(got["fused"]) == 1
        differs |= not np.array_equal(got["V"].view(np.uint32), single.value_function.view(np.uint32))
    assert differs, "values were dropped from the deliveries and nothing noticed: the poisoning check checks nothing"
