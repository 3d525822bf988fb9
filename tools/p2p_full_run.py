"""tools/p2p_full_run.py env bins ranks — a full run() to convergence with `ranks` rank PROCESSES on ONE GPU over the
peer-to-peer transport (csrc/pi_p2p.cpp; fused exchange where the plan allows), digests of V / policy as tools/full_run.py
prints them for one rank: the same digests mean that ~10^5 sweeps of hand-shaken halo deliveries moved not one bit.

The ranks share the box's GPU (HIP IPC needs nothing else), torch.distributed runs over the CPU backend.  Times are
N processes on one device — they say what the exchange machinery costs there, nothing about N GPUs.

usage: python tools/p2p_full_run.py double_pendulum_swingup 80 2        (PI_MI355_P2P_FUSED=0: copy kernel)
"""
import hashlib
import json
import os
import socket
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))


def worker(rank, world, port, name, bins, out):
    os.environ["PI_MI355_TRANSPORT"] = "p2p"
    os.environ.setdefault("PI_MI355_COMM_TIMEOUT", "60")
    import datetime
    import logging
    import torch
    import torch.distributed as dist
    from dynamicprogramming_amd import envs
    if rank == 0:                                  # progress (one line per policy evaluation) on stderr
        logging.basicConfig(level=logging.INFO, stream=sys.stderr, format="%(asctime)s %(message)s")
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world,
                            timeout=datetime.timedelta(seconds=600))
    try:
        s = envs.make(name, bins, device="cuda:0")
        eng = s._backend.engine
        info = {"transport": eng.comm_info(2), "plan": dict(s._comm.info), "row_exact": eng.comm_info(5), "fused": eng.comm_info(6),
                "memory_order": list(eng.order)}
        torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        s.run()
        dt = time.perf_counter() - t0
        st = s.stats
        if rank == 0:
            bk = s.n_states * (st["eval_sweeps"] + st["improve_sweeps"] * s.n_actions)
            Path(out).write_text(json.dumps({
                "env": name, "bins": bins, "states": s.n_states, "rank_processes_on_one_gpu": world,
                "pi_iterations": st["pi_iterations"], "eval_sweeps": st["eval_sweeps"], "improve_sweeps": st["improve_sweeps"],
                "stable": st.get("stable"), "seconds": round(dt, 2), "backups_per_s": bk / dt,
                "us_per_eval_sweep": dt / max(st["eval_sweeps"], 1) * 1e6,
                "sha256_V": hashlib.sha256(s.value_function.tobytes()).hexdigest()[:16],
                "sha256_policy": hashlib.sha256(s.policy.tobytes()).hexdigest()[:16], "exchange": info}))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def main():
    import tempfile
    import torch.multiprocessing as mp
    name, bins, world = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    assert 2 <= world <= 5, "at most 5 rank processes on one GPU box"
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    with tempfile.TemporaryDirectory(prefix="p2p_full_") as tmp:
        out = str(Path(tmp) / "rank0.json")
        mp.spawn(worker, args=(world, port, name, bins, out), nprocs=world, join=True)
        print(Path(out).read_text(), flush=True)


if __name__ == "__main__":
    main()
