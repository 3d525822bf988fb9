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
