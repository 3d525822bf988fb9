"""
bench.py — state-action Bellman backups/s of the policy-iteration hot path on MI355X.

Workload (BASELINE.json metric config, SURVEY.md §8d "C4"): double-pendulum swing-up, 4-D
grid 80^4 = 40.96 M states x 11 torques, gamma 0.999, fp32.  Synthetic inputs: V ~ N(0,1),
policy ~ U{0..10} (seeded), no terminal states (the env has none).  Everything is resident
in HBM before the timed region.

A STEP = one pass of the hot path over the whole grid in the sweep mix of SURVEY §8(d)'s
protocol (100 evaluation + 10 improvement sweeps): 10 evaluation sweeps (each with the fused
residual on the last) followed by 1 greedy improvement sweep (with the fused changed-count)
= 10 n + 11 n = 21 n state-action backups, through the product path (solver -> C ABI -> HIP).

N > 1 (`python -m torch.distributed.run ... bench.py --gpus N`): one process per GPU, the
SAME grid sharded into N contiguous state ranges (strong scaling), V shards all-gathered over
RCCL/xGMI after every evaluation sweep — the solver's own multi-rank path.

Prints ONE JSON line on rank 0 (contract in the task statement), plus
  roofline     — dominant kernel = pi_eval_sweep_kernel; algorithmic bytes/backup from
                 SURVEY §8(d): 4*2^D + 4*D + 9 = 89 B (4-D), one backup per state per launch;
                 achieved = 89 B * states-per-launch / mean launch time (HIP events on the
                 launch stream around each 10-sweep group); peak = 8 TB/s HBM3E.
  cpu_baseline — the oracle (oracle/pi_oracle.cpp, OpenMP) on the same workload restricted
                 to a bounded sample of states, on this box's host cores (rank 0, N = 1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
ENV = "double_pendulum_swingup"
BINS = 80
EVAL_PER_STEP = 10
IMPROVE_PER_STEP = 1


def algorithmic_bytes_eval(D: int) -> int:
    return 4 * (1 << D) + 4 * D + 9          # 89 B for D = 4


def cpu_baseline(env: str, bins: int, sample_states: int, seed: int = 0) -> dict:
    """Oracle timed on host cores over states [0, sample) of the same grid / V / policy."""
    import oracle
    from dynamicprogramming_amd import envs
    cls = envs.ENVS[env]
    tables = [np.asarray(b, np.float32) for b in cls.bins_space(bins).values()]
    lo, hi, shape, strides = oracle.grid_metadata(tables)
    n = int(np.prod(shape))
    m = min(sample_states, n)
    idx = np.stack(np.unravel_index(np.arange(m), tuple(shape)), axis=1)
    states = np.stack([tables[d][idx[:, d]] for d in range(len(tables))], axis=1).astype(np.float32)
    rng = np.random.default_rng(seed)
    V = rng.standard_normal(n).astype(np.float32)
    pol = rng.integers(0, len(cls.ACTIONS), size=m).astype(np.int32)
    term = np.zeros(m, dtype=np.uint8)
    chk = oracle.build(cls._D, envs.dynamics_source(env))
    gamma = np.float32(cls.CONFIG["gamma"])
    out = np.zeros(m, dtype=np.float32)
    for _ in range(2):   # warm the OpenMP team and the caches before timing
        chk.eval_sweep(states, cls.ACTIONS, pol, V, term, lo, hi, shape, strides, gamma, 0, m, out=out)
    chk.improve_sweep(states, cls.ACTIONS, pol, V, term, lo, hi, shape, strides, gamma, 0, min(m, 1 << 16))
    t0 = time.perf_counter()
    for _ in range(EVAL_PER_STEP):
        chk.eval_sweep(states, cls.ACTIONS, pol, V, term, lo, hi, shape, strides, gamma, 0, m, out=out)
    for _ in range(IMPROVE_PER_STEP):
        chk.improve_sweep(states, cls.ACTIONS, pol, V, term, lo, hi, shape, strides, gamma, 0, m)
    dt = time.perf_counter() - t0
    backups = m * (EVAL_PER_STEP + IMPROVE_PER_STEP * len(cls.ACTIONS))
    threads = chk.threads
    return {"value": backups / dt, "unit": "backups/s", "cores": threads, "kind": "port",
            "sample": f"one step (10 eval + 1 improve sweeps) over states [0, {m}) of the same "
                      f"{bins}^{cls._D} grid, oracle/pi_oracle.cpp with OpenMP, {dt:.1f} s wall"}


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--bins", type=int, default=None, help="grid points per dimension (default: the BASELINE config)")
    ap.add_argument("--env", default=ENV, help="env plugin (default: the BASELINE metric config, "
                    "double_pendulum_swingup at 80 bins; other BASELINE configs: cartpole_swingup@50, "
                    "double_cartpole@25, double_cartpole_swingup@25, pendulum@200)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--transition-cache", action="store_true",
                    help="record transitions on the first evaluation sweep of a step and replay them "
                         "on the others (default: recompute the dynamics every sweep, as the reference)")
    ap.add_argument("--cpu-sample", type=int, default=1 << 23,
                    help="states of the same grid the CPU baseline sweeps (about 15 CPU-seconds)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from dynamicprogramming_amd import envs

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=dev)

    cls = envs.ENVS[args.env]
    if args.bins is None:
        args.bins = BINS if args.env == ENV else cls.DEFAULT_BINS
    cfg = envs.CudaPIConfig(**cls.CONFIG, cache_transitions=args.transition_cache)
    solver = envs.make(args.env, args.bins, config=cfg, device=dev)
    n, nA, D = solver.n_states, solver.n_actions, cls._D
    gamma = float(np.float32(solver.config.gamma))

    # synthetic resident inputs (identical on every rank)
    gen = torch.Generator(device="cpu").manual_seed(0)
    V0 = torch.randn(n, generator=gen, dtype=torch.float32)
    P0 = torch.randint(0, nA, (n,), generator=gen, dtype=torch.int32)
    solver.d_value_function[:n].copy_(V0)
    solver.d_new_value_function.copy_(solver.d_value_function)
    solver.d_policy[:n].copy_(P0)
    del V0, P0

    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(args.steps)]
    cached = getattr(solver._backend, "_cache", None) is not None

    def step(k=None):
        # One policy evaluation of EVAL_PER_STEP sweeps (sweep 0 also records the transitions
        # when the cache is on; the rest replay them), then one improvement.
        if k is not None:
            ev[k][0].record()
        solver._evaluation_sweeps(1, gamma)
        if k is not None:
            ev[k][1].record()
        solver._evaluation_sweeps(EVAL_PER_STEP - 1, gamma)
        if k is not None:
            ev[k][2].record()
        solver._improvement_sweep(gamma)
        if k is not None:
            ev[k][3].record()

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(k)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    first_ms = float(np.mean([e[0].elapsed_time(e[1]) for e in ev]))
    rest_ms = float(np.mean([e[1].elapsed_time(e[2]) for e in ev])) / (EVAL_PER_STEP - 1)
    improve_ms = float(np.mean([e[2].elapsed_time(e[3]) for e in ev])) / IMPROVE_PER_STEP
    # sanity: the sweeps really ran (residual and change count of the last step)
    last_delta = float(solver._d_delta.item())
    last_changed = int(solver._d_changed.item())

    backups_per_step = n * (EVAL_PER_STEP + IMPROVE_PER_STEP * nA)
    value = backups_per_step * args.steps / elapsed
    states_per_launch = solver._s_end - solver._s_begin
    bytes_eval = algorithmic_bytes_eval(D)
    bytes_improve = 4 * (1 << D) + (4 * D + 1 + 4) / nA

    tiled = bool(solver._backend.engine.info(10))

    def roof(kernel, bytes_per_backup, backups, ms):
        ach = bytes_per_backup * backups / (ms * 1e-3) / 1e9
        return {"bound": "hbm", "kernel": kernel, "achieved": ach, "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": None,
                "bytes_per_backup": bytes_per_backup, "backups_per_launch": backups,
                "avg_launch_ms": ms}

    kernels = {
        "first_eval_sweep": roof(("pi_tile_build_kernel" if tiled else "pi_eval_build_kernel") if cached
                                 else ("pi_tile_eval_kernel" if tiled else "pi_eval_sweep_kernel"),
                                 bytes_eval, states_per_launch, first_ms),
        "other_eval_sweeps": roof(("pi_tile_replay_kernel" if tiled else "pi_eval_replay_kernel") if cached
                                  else ("pi_tile_eval_kernel" if tiled else "pi_eval_sweep_kernel"),
                                  bytes_eval, states_per_launch, rest_ms),
        "improve_sweep": roof("pi_improve_sweep_kernel", bytes_improve, states_per_launch * nA,
                              improve_ms),
    }
    # HBM traffic per launch from the committed PMC passes of this same command
    # (tools/profile_bench.sh -> profiles/rNN/traffic_bench_c4.json), gfx950-corrected.
    if world == 1:
        for tf in sorted(ROOT.glob("profiles/r*/traffic_bench_c4.json"), reverse=True):
            t = json.loads(tf.read_text())
            if t.get("states") == n:
                for k in kernels.values():
                    hit = t["kernels"].get(k["kernel"])
                    if hit:
                        k["traffic"] = hit["hbm_bytes_corrected"]
                        k["traffic_source"] = str(tf.relative_to(ROOT))
                break
    share = {"first_eval_sweep": first_ms, "other_eval_sweeps": rest_ms * (EVAL_PER_STEP - 1),
             "improve_sweep": improve_ms * IMPROVE_PER_STEP}
    dominant = max(share, key=share.get)
    eng = solver._backend.engine
    out = {
        "metric": "state-action Bellman backups/sec",
        "value": value,
        "unit": "backups/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": f"{args.env} {D}D grid bins={args.bins}/dim "
                               f"({n} states) x {nA} actions, gamma={solver.config.gamma}; step = {EVAL_PER_STEP} "
                               f"eval sweeps + {IMPROVE_PER_STEP} improve sweep = {backups_per_step} backups",
                   "states": n, "actions": nA, "eval_sweeps_per_step": EVAL_PER_STEP,
                   "improve_sweeps_per_step": IMPROVE_PER_STEP, "transition_cache": cached,
                   "parallelism": f"state-range shards x{world}" + (", RCCL all-gather of V per eval sweep" if world > 1 else "")},
        "roofline": kernels[dominant],
        "kernels": kernels,
        "time_share_ms": share,
        "eval_backups_per_s": states_per_launch * world * EVAL_PER_STEP / ((first_ms + rest_ms * (EVAL_PER_STEP - 1)) * 1e-3),
        "improve_backups_per_s": states_per_launch * world * nA / (improve_ms * 1e-3),
        "check": {"last_residual": last_delta, "last_changed": last_changed,
                  "vgpr_eval": eng.info(4), "vgpr_improve": eng.info(5), "vgpr_replay": eng.info(8),
                  "replay_states_per_thread": eng.info(9), "tiled": bool(eng.info(10)),
                  "tile": [eng.info(30 + d) for d in range(D)], "box": [eng.info(20 + d) for d in range(D)],
                  "reach": [eng.info(40 + d) for d in range(D)], "tiled_blocks": eng.info(11),
                  "eval_blocks_per_cu": eng.info(12), "improve_blocks_per_cu": eng.info(13)},
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args.env, args.bins, args.cpu_sample)
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
