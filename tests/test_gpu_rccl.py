"""First contact with RCCL at N > 1 ranks — in a TEST, not inside the timed scaling bench.

The development boxes of this repository have one GPU, where RCCL refuses a second rank ("Duplicate GPU detected"), so
`ncclCommInitRank` with world > 1, the grouped ncclSend / ncclRecv halo exchange and `ncclAllGather`
(csrc/pi_comm.cpp) have only ever run with one rank.  These tests enable themselves when at least two GPUs are visible:
they start the ranks as FRESH child processes through the launcher of the bench contract (never an exec of a process
that has touched a GPU), bounded by a time limit that kills the whole process group, and demand bit-identity with the
single-rank run.  On a one-GPU box they skip and say so.
"""
from __future__ import annotations

import json
import os
import signal
import socket
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
pytestmark = pytest.mark.gpu


def _gpus() -> int:
    import torch
    return torch.cuda.device_count()            # counting devices does not initialise them


def _free_port() -> int:
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        return sock.getsockname()[1]


def _run_bounded(cmd, env, limit):
    """Run `cmd` in a session of its own; on overrun kill its whole process group.  Returns (rc, stdout, stderr)."""
    proc = subprocess.Popen(cmd, env=env, text=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE, start_new_session=True)
    try:
        out, err = proc.communicate(timeout=limit)
        return proc.returncode, out, err
    except subprocess.TimeoutExpired:
        for sig in (signal.SIGTERM, signal.SIGKILL):
            try:
                os.killpg(proc.pid, sig)
            except ProcessLookupError:
                break
            try:
                proc.wait(timeout=10)
                break
            except subprocess.TimeoutExpired:
                continue
        out, err = proc.communicate()
        return 124, out, err


def _need_two_gpus():
    n = _gpus()
    if n < 2:
        reason = f"{n} GPU visible: RCCL with N > 1 ranks needs at least 2 (this test runs by itself on a multi-GPU node)"
        print("SKIP: " + reason)
        pytest.skip(reason)
    return n


@pytest.mark.parametrize("exchange", ["halo", "allgather"])
def test_rccl_ranks_reproduce_the_single_rank_run(exchange):
    """2 ranks (4 when 4 GPUs are visible), the halo exchange and the all-gather: V, policy and the sweep count of every
    evaluation equal the single-rank run on every rank; the communicator reports the launcher's world size over RCCL."""
    n = _need_two_gpus()
    world = 4 if n >= 4 else 2
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update({"PI_MI355_EXCHANGE": exchange, "HSA_ENABLE_IPC_MODE_LEGACY": "0", "PI_MI355_COMM_TIMEOUT": "60"})
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), str(ROOT / "tests" / "rccl_ranks.py")]
    rc, out, err = _run_bounded(cmd, env, 420)
    reports = [json.loads(line.split("RCCL_RANK ", 1)[1]) for line in out.splitlines() if line.startswith("RCCL_RANK ")]
    assert rc == 0, f"rc {rc}\n{out[-3000:]}\n{err[-3000:]}"
    assert sorted(r["rank"] for r in reports) == list(range(world))
    for r in reports:
        assert r["ok"] and r["world"] == world
        for case in r["cases"]:
            assert case["identical"] and case["comm"]["world"] == world and case["comm"]["transport"] == "rccl"
            assert case["comm"]["plan"]["mode"] == exchange


def test_bench_line_of_two_ranks_says_what_rccl_did():
    """`python bench.py --gpus 2` — the command of the scaling run, shortened — through its own launcher, supervisors
    and ladder: the line's check.exchange names 2 ranks over RCCL and sharded == unsharded before and after the timed steps."""
    _need_two_gpus()
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    cmd = [sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--no-cpu-baseline",
           "--attempt-timeout", "200"]
    rc, out, err = _run_bounded(cmd, env, 1500)
    assert rc == 0, f"rc {rc}\n{out[-3000:]}\n{err[-3000:]}"
    line = [json.loads(l) for l in out.splitlines() if l.startswith("{") and '"metric"' in l][-1]
    x = line["check"]["exchange"]
    assert line["n_gpus"] == 2 and x["world"] == 2
    assert x["attempts"][0]["ok"] is True, x["attempts"]           # the first rung (halo + overlap over RCCL) worked
    assert x["transport"] == "rccl" and x["bit_identical"]["ok"] and x["bit_identical"]["after_timed_region"]["ok"]
    assert line["value"] == line["value_by_transport"]["rccl"] if "value_by_transport" in line else True
