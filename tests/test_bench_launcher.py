"""bench.py's launcher logic (no GPU): `python bench.py --gpus N` outside torch.distributed.run starts
the N ranks as a CHILD process with the launch line of the bench contract, relays rank 0's JSON line
and propagates a failing rank's exit code."""
from __future__ import annotations

import importlib.util
import json
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]


def _bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", ROOT / "bench.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_dry_run_prints_the_contract_launch_line():
    res = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                          "--launch-dry-run"], capture_output=True, text=True, timeout=120,
                         env={k: v for k, v in __import__("os").environ.items()
                              if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")})
    assert res.returncode == 0, res.stderr
    cmd = json.loads(res.stdout.strip().splitlines()[-1])["launch"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=2" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert 1024 <= int(cmd[cmd.index("--master-port") + 1]) < 65536
    i = cmd.index(str(ROOT / "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "2", "--steps", "3", "--warmup", "1"]      # same args, dry-run flag dropped


def test_ranks_started_by_a_launcher_must_match_gpus():
    import os
    env = {**os.environ, "WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"}
    res = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2"], capture_output=True, text=True,
                         timeout=300, env=env)
    assert res.returncode != 0 and "--nproc-per-node must equal --gpus" in res.stderr


def test_child_line_is_relayed_and_failures_propagate(monkeypatch, capsys):
    bench = _bench()
    noise = "import sys; print('rank noise'); print('{\"not\": \"the line\"}'); "
    ok = noise + "print('{\"metric\": \"m\", \"value\": 1.0, \"n_gpus\": 2}'); print('trailing noise')"
    monkeypatch.setattr(bench, "launch_command", lambda n, argv, port=None: [sys.executable, "-c", ok])
    assert bench.self_launch(2, ["--gpus", "2"], dry_run=False) == 0
    out = capsys.readouterr().out.strip().splitlines()
    assert out == ['{"metric": "m", "value": 1.0, "n_gpus": 2}']               # exactly one line, rank 0's

    bad = noise + "print('{\"metric\": \"m\", \"value\": 1.0}'); sys.exit(3)"
    monkeypatch.setattr(bench, "launch_command", lambda n, argv, port=None: [sys.executable, "-c", bad])
    assert bench.self_launch(2, ["--gpus", "2"], dry_run=False) == 3             # a failed rank fails the bench

    monkeypatch.setattr(bench, "launch_command", lambda n, argv, port=None: [sys.executable, "-c", noise])
    assert bench.self_launch(2, ["--gpus", "2"], dry_run=False) == 1             # no result line -> failure


# ── the ranks' supervisors (N > 1): fallback ladder, time limit, lockstep over gloo — with fake ranks ──────────────
FAKE_RANK = r'''
import json, os, sys, time
rank, attempt, mode = int(os.environ["RANK"]), int(os.environ["PI_BENCH_ATTEMPT"]), os.environ["PI_BENCH_MODE"]
assert os.environ["PI_BENCH_WORKER"] == "1"
plan = json.loads(os.environ["FAKE_PLAN"])            # per attempt: what each rank does
what = plan[attempt][rank] if attempt < len(plan) else "ok"
if mode == "halo":
    assert os.environ.get("PI_MI355_OVERLAP") == "0"
if mode.startswith("allgather"):
    assert os.environ.get("PI_MI355_EXCHANGE") == "allgather"
assert (os.environ.get("PI_BENCH_MINIMAL") == "1") == (mode == "allgather-minimal")
over_p2p = mode.endswith(" over p2p")                 # the ladder's last rung (attempt 4) or the best-effort rerun (5)
bonus = over_p2p and attempt >= 5
assert (os.environ.get("PI_MI355_TRANSPORT") == "p2p") == over_p2p and (os.environ.get("PI_BENCH_BONUS") == "1") == bonus
if what == "hang":
    time.sleep(600)
if what.startswith("exit"):
    sys.exit(int(what[4:]))
if rank == 0:
    print("noise before")
    print(json.dumps({"metric": "m", "value": float(os.environ.get("FAKE_P2P_VALUE", "1.0")) if bonus else 1.0,
                      "n_gpus": int(os.environ["WORLD_SIZE"]), "ms_per_step": 2.0,
                      "check": {"exchange": {"mode": mode, "port": os.environ["MASTER_PORT"]}}}))
'''


def _supervisor_rank(rank, world, port, plan, timeout, queue):
    import contextlib
    import io
    import os
    os.environ.update({"RANK": str(rank), "LOCAL_RANK": str(rank), "WORLD_SIZE": str(world),
                       "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "FAKE_PLAN": json.dumps(plan)})
    bench = _bench()
    bench.worker_command = lambda argv: [sys.executable, "-c", FAKE_RANK]
    out = io.StringIO()
    with contextlib.redirect_stdout(out):
        rc = bench.supervise(["--gpus", str(world)], attempt_timeout=timeout)
    queue.put((rank, rc, out.getvalue()))


def _run_supervisors(plan, timeout=30.0, world=2):
    import multiprocessing as mp
    bench = _bench()
    ctx = mp.get_context("spawn")
    queue = ctx.Queue()
    port = bench._free_port()
    procs = [ctx.Process(target=_supervisor_rank, args=(r, world, port, plan, timeout, queue)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted(queue.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    return got


def test_supervisors_relay_the_first_rung_when_it_works():
    (r0, rc0, out0), (r1, rc1, out1) = _run_supervisors([["ok", "ok"]])
    assert (rc0, rc1) == (0, 0) and out1.strip() == ""
    (line,) = out0.strip().splitlines()                                  # exactly one line, on rank 0
    obj = json.loads(line)
    assert obj["check"]["exchange"]["mode"] == "halo+overlap"
    assert [a["ok"] for a in obj["check"]["exchange"]["attempts"]] == [True]
    # ... and the same exchange ran once more over the peer-to-peer transport, reported beside it
    p2p = obj["check"]["exchange"]["p2p"]
    assert p2p["ok"] is True and p2p["mode"] == "halo+overlap over p2p" and p2p["value"] == 1.0


def test_value_is_the_primary_rung_and_the_p2p_rerun_stands_beside_it(monkeypatch):
    """`value` is the transport chosen up front (the first rung that works); a faster best-effort rerun over the
    peer-to-peer transport is reported under value_by_transport / check.exchange.p2p and never replaces it."""
    for p2p_value in ("3.5", "0.5"):
        monkeypatch.setenv("FAKE_P2P_VALUE", p2p_value)
        (_, rc0, out0), (_, rc1, _) = _run_supervisors([["ok", "ok"]])
        assert (rc0, rc1) == (0, 0)
        (line,) = out0.strip().splitlines()
        obj = json.loads(line)
        assert obj["value"] == 1.0 and obj["value_by_transport"]["rccl"] == 1.0
        assert obj["value_by_transport"]["p2p"] == float(p2p_value) and obj["value_by_transport"]["reported"].startswith("rccl")
        x = obj["check"]["exchange"]
        assert x["mode"] == "halo+overlap" and x["p2p"]["ok"] is True and x["p2p"]["value"] == float(p2p_value)
        assert [a["mode"] for a in x["attempts"]] == ["halo+overlap"]


def test_a_failing_p2p_rerun_costs_nothing():
    plan = [["ok", "ok"]] * 5 + [["ok", "exit7"]]                        # index 5 = len(LADDER) = the p2p rerun
    (_, rc0, out0), (_, rc1, out1) = _run_supervisors(plan)
    assert (rc0, rc1) == (0, 0) and out1.strip() == ""
    obj = json.loads(out0.strip().splitlines()[-1])
    assert obj["value"] == 1.0 and [a["ok"] for a in obj["check"]["exchange"]["attempts"]] == [True]
    p2p = obj["check"]["exchange"]["p2p"]
    assert p2p["ok"] is False and p2p["exit_codes"][1] == 7 and "value" not in p2p


def test_a_failing_rank_moves_every_rank_to_the_next_rung():
    # rung 0: rank 1 exits 3 while rank 0 would hang in a collective; rung 1 (halo, no overlap) works
    got = _run_supervisors([["hang", "exit3"], ["ok", "ok"]], timeout=60.0)
    (_, rc0, out0), (_, rc1, _) = got
    assert (rc0, rc1) == (0, 0)
    obj = json.loads(out0.strip().splitlines()[-1])
    att = obj["check"]["exchange"]["attempts"]
    assert [a["mode"] for a in att] == ["halo+overlap", "halo"] and [a["ok"] for a in att] == [False, True]
    assert att[0]["exit_codes"][1] == 3 and att[0]["timeout"] is False
    assert att[0]["seconds"] < 30                                        # the hung rank was killed at once, not at the limit
    assert obj["check"]["exchange"]["mode"] == "halo"


def test_a_hung_rung_is_killed_at_the_time_limit_and_the_ladder_ends_in_allgather():
    got = _run_supervisors([["hang", "hang"], ["exit1", "ok"], ["ok", "ok"]], timeout=4.0)
    (_, rc0, out0), (_, rc1, _) = got
    assert (rc0, rc1) == (0, 0)
    att = json.loads(out0.strip().splitlines()[-1])["check"]["exchange"]["attempts"]
    assert [a["mode"] for a in att] == ["halo+overlap", "halo", "allgather"]
    assert att[0]["timeout"] is True and att[1]["ok"] is False and att[2]["ok"] is True
    # every rung had its own rendezvous port


def test_the_peer_to_peer_rung_rescues_a_run_whose_rccl_rungs_all_fail():
    got = _run_supervisors([["exit2", "ok"], ["ok", "exit2"], ["exit2", "exit2"], ["ok", "exit5"], ["ok", "ok"]], timeout=30.0)
    (_, rc0, out0), (_, rc1, _) = got
    assert (rc0, rc1) == (0, 0)
    obj = json.loads(out0.strip().splitlines()[-1])
    att = obj["check"]["exchange"]["attempts"]
    assert [a["mode"] for a in att] == ["halo+overlap", "halo", "allgather", "allgather-minimal", "halo+overlap over p2p"]
    assert [a["ok"] for a in att] == [False, False, False, False, True]
    assert "p2p" not in obj["check"]["exchange"] and "value_by_transport" not in obj      # no rerun of a p2p rung over p2p


def test_when_every_rung_fails_every_rank_fails():
    got = _run_supervisors([["exit2", "ok"], ["ok", "exit2"], ["exit2", "exit2"], ["ok", "exit5"], ["exit9", "ok"]], timeout=30.0)
    assert [rc for _, rc, _ in got] == [1, 1]
    assert all(out.strip() == "" for _, _, out in got)                  # no result line is invented


def test_dry_run_shows_the_ladder():
    res = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "4", "--launch-dry-run"], capture_output=True,
                         text=True, timeout=120, env={k: v for k, v in __import__("os").environ.items()
                                                      if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")})
    assert res.returncode == 0, res.stderr
    obj = json.loads(res.stdout.strip().splitlines()[-1])
    assert [r["mode"] for r in obj["ladder"]] == ["halo+overlap", "halo", "allgather", "allgather-minimal",
                                                 "halo+overlap over p2p"]
    assert obj["ladder"][4]["env"]["PI_MI355_TRANSPORT"] == "p2p" and obj["ladder"][4]["env"]["PI_BENCH_BONUS"] == "0"
    assert obj["bonus_after_first_success"]["env"]["PI_BENCH_BONUS"] == "1"
    assert obj["ladder"][3]["env"]["PI_BENCH_MINIMAL"] == "1"
    assert obj["ladder"][2]["env"]["PI_MI355_EXCHANGE"] == "allgather" and obj["ladder"][0]["timeout_s"] == 240.0
    assert obj["worker"][1].endswith("bench.py")


def test_dry_run_of_the_one_gpu_rehearsal_shows_its_own_ladder():
    env = {k: v for k, v in __import__("os").environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["PI_BENCH_SHARE_GPU"] = "1"
    res = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--launch-dry-run"], capture_output=True,
                         text=True, timeout=120, env=env)
    assert res.returncode == 0, res.stderr
    obj = json.loads(res.stdout.strip().splitlines()[-1])
    assert [r["mode"] for r in obj["ladder"]] == ["halo+overlap over p2p", "halo over p2p", "allgather over p2p"]
    assert all(r["env"]["PI_MI355_TRANSPORT"] == "p2p" for r in obj["ladder"]) and "rehearsal" in obj


def test_self_launch_kills_a_child_that_overruns(monkeypatch):
    bench = _bench()
    monkeypatch.setattr(bench, "launch_command", lambda n, argv, port=None: [sys.executable, "-c", "import time; time.sleep(600)"])
    monkeypatch.setattr(bench, "LADDER", [])                            # limit = 0 x timeout + 300 -> patch the constant part too
    import time as _t
    t0 = _t.monotonic()
    real_popen = subprocess.Popen

    class Quick(real_popen):
        def communicate(self, input=None, timeout=None):                # noqa: A002
            return super().communicate(input=input, timeout=min(timeout or 2.0, 2.0))
    monkeypatch.setattr(bench.subprocess, "Popen", Quick)
    assert bench.self_launch(2, ["--gpus", "2"], dry_run=False) == 124
    assert _t.monotonic() - t0 < 60


def test_under_the_real_launcher_the_ranks_children_can_rendezvous(tmp_path):
    """The contract's launch line for real: torch.distributed.run starts two supervisors, each starts a child rank, and
    the CHILDREN form their own process group (gloo here) on the port the supervisors agreed on — which only works if the
    child does not inherit the launcher's agent store (TORCHELASTIC_USE_AGENT_STORE points at the launcher's port)."""
    import os
    fake = tmp_path / "fake_rank.py"
    fake.write_text(
        "import json, os, torch, torch.distributed as dist\n"
        "assert os.environ['PI_BENCH_WORKER'] == '1' and 'TORCHELASTIC_USE_AGENT_STORE' not in os.environ\n"
        "dist.init_process_group('gloo')\n"
        "t = torch.ones(1)\n"
        "dist.all_reduce(t)\n"
        "assert t.item() == dist.get_world_size() == 2\n"
        "if dist.get_rank() == 0:\n"
        "    print(json.dumps({'metric': 'm', 'value': 2.0, 'n_gpus': 2, 'check': {'exchange': {'mode': os.environ['PI_BENCH_MODE']}}}))\n"
        "dist.destroy_process_group()\n")
    driver = tmp_path / "supervisor.py"
    driver.write_text(
        "import importlib.util, sys\n"
        f"spec = importlib.util.spec_from_file_location('bench_under_test', r'{ROOT / 'bench.py'}')\n"
        "bench = importlib.util.module_from_spec(spec)\n"
        "spec.loader.exec_module(bench)\n"
        f"bench.worker_command = lambda argv: [sys.executable, r'{fake}']\n"
        "sys.exit(bench.supervise(['--gpus', '2'], attempt_timeout=90.0))\n")
    bench = _bench()
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    res = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", str(bench._free_port()), str(driver)],
                         capture_output=True, text=True, timeout=300, env=env)
    assert res.returncode == 0, res.stderr[-3000:]
    line = bench.result_line(res.stdout)
    assert line is not None, res.stdout
    obj = json.loads(line)
    assert obj["value"] == 2.0 and [a["ok"] for a in obj["check"]["exchange"]["attempts"]] == [True]
    assert obj["check"]["exchange"]["attempts"][0]["mode"] == "halo+overlap"


def test_the_minimal_rung_is_the_last_resort():
    got = _run_supervisors([["exit1", "ok"], ["ok", "exit1"], ["exit1", "exit1"], ["ok", "ok"]], timeout=30.0)
    (_, rc0, out0), (_, rc1, _) = got
    assert (rc0, rc1) == (0, 0)
    att = json.loads(out0.strip().splitlines()[-1])["check"]["exchange"]["attempts"]
    assert [a["mode"] for a in att] == ["halo+overlap", "halo", "allgather", "allgather-minimal"]
    assert [a["ok"] for a in att] == [False, False, False, True]


SIGNALLED_SUPERVISOR = r"""
import importlib.util, os, sys
spec = importlib.util.spec_from_file_location('bench_under_test', sys.argv[1])
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)
marker = sys.argv[2]
child = "import os, sys, time; open(sys.argv[1], 'w').write(str(os.getpid())); time.sleep(600)"
bench.worker_command = lambda argv: [sys.executable, '-c', child, marker]
sys.exit(bench.supervise(['--gpus', '1'], attempt_timeout=300.0, ladder=[('only', {})], bonus_p2p=False))
"""


def test_a_terminated_supervisor_takes_its_worker_with_it(tmp_path):
    """The worker of a rung lives in a session of its own, so nobody else reaps it: SIGTERM to the supervisor (the launcher
    tearing the remaining ranks down, self_launch killing an overrun) must end the worker before the supervisor exits —
    an orphaned rank would keep its GPU busy, inside an RCCL collective for minutes."""
    import os
    import signal
    import time
    bench = _bench()
    marker = tmp_path / "worker.pid"
    env = {**os.environ, "RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1", "MASTER_ADDR": "127.0.0.1",
           "MASTER_PORT": str(bench._free_port())}
    sup = subprocess.Popen([sys.executable, "-c", SIGNALLED_SUPERVISOR, str(ROOT / "bench.py"), str(marker)], env=env)
    try:
        deadline = time.monotonic() + 120
        while not (marker.exists() and marker.read_text().strip()) and time.monotonic() < deadline:
            assert sup.poll() is None, "the supervisor ended before its worker started"
            time.sleep(0.2)
        worker = int(marker.read_text())
        os.kill(worker, 0)                                               # alive
        sup.send_signal(signal.SIGTERM)
        assert sup.wait(timeout=60) == 128 + signal.SIGTERM
        for _ in range(100):                                             # ... and the worker is gone with it
            try:
                os.kill(worker, 0)
            except ProcessLookupError:
                break
            time.sleep(0.1)
        else:
            os.kill(worker, signal.SIGKILL)
            raise AssertionError("the worker outlived its supervisor")
    finally:
        if sup.poll() is None:
            sup.kill()


def test_a_rank_over_the_peer_to_peer_transport_never_touches_the_nccl_backend(monkeypatch):
    """The ladder's last rung is what is left when RCCL itself fails: such a rank joins torch.distributed over gloo and
    keeps the small tensors of its collectives on the host — as the rehearsal and the best-effort rerun do."""
    import torch
    bench = _bench()
    for env_name in ("PI_MI355_TRANSPORT", "PI_BENCH_SHARE_GPU"):
        monkeypatch.delenv(env_name, raising=False)
    assert not bench.over_p2p() and bench._collective_device(torch.device("cuda", 0)).type == "cuda"
    monkeypatch.setenv("PI_MI355_TRANSPORT", "p2p")
    assert bench.over_p2p() and bench._collective_device(torch.device("cuda", 0)).type == "cpu"
    # every rung of the ladder that runs over the peer-to-peer transport says so in its environment (what over_p2p reads)
    assert [m for m, e in bench.LADDER if e.get("PI_MI355_TRANSPORT") == "p2p"] == ["halo+overlap over p2p"]
    assert bench.BONUS_P2P["PI_MI355_TRANSPORT"] == "p2p"
    src = (ROOT / "bench.py").read_text()
    assert src.count('init_process_group("nccl"') == 1 and 'if over_p2p():\n            dist.init_process_group("gloo")' in src
