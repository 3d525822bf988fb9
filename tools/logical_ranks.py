"""tools/logical_ranks.py ENV BINS WORLD [SWEEPS] — the sharded driver of libpi_mi355.so at FULL size
with WORLD logical ranks on ONE GPU (GPU box; one host thread, one stream and one full set of V buffers
per rank; the in-process transport of csrc/pi_comm.cpp, same planner, kernels and stream ordering as the
RCCL transport).  Everything printed is "one GPU, logical ranks": it shows what each rank of a real
N-GPU run would sweep, send and receive, NOT how fast N GPUs are.

For the config it prints one JSON line with
  * the plan of every rank: exchange mode, bytes received / sent per evaluation sweep, the launch
    ranges (swept first / interior) with their state counts;
  * bit-identity: after SWEEPS evaluation sweeps + one improvement sweep from a seeded V / policy,
    every rank's shard of V and of the policy equals the single-rank result;
  * the measured single-GPU time split: each rank's swept-first launches and interior launches timed
    alone on the GPU (HIP events), next to the whole-grid sweep time — compute a rank has to finish
    before its halo can leave vs compute that can hide the transfer.
"""
import json
import sys
import threading
import uuid
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
import torch

from dynamicprogramming_amd import envs
from dynamicprogramming_amd import transport as T

env, bins, world = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
sweeps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
dev = torch.device("cuda:0")
cls = envs.ENVS[env]
cfg = envs.CudaPIConfig(**cls.CONFIG)
gamma = float(np.float32(cfg.gamma))


def seed(solver):
    n = solver.n_states
    gen = torch.Generator(device="cpu").manual_seed(0)
    V = torch.randn(n, generator=gen, dtype=torch.float32)
    P = torch.randint(0, solver.n_actions, (n,), generator=gen, dtype=torch.int32)
    solver.d_value_function[:n].copy_(solver._to_memory(V))        # the same problem whatever the solver's memory order
    term = solver.d_terminal_mask[:n].bool()
    solver.d_value_function[:n][term] = 0.0
    solver.d_new_value_function.copy_(solver.d_value_function)
    solver.d_policy[:n].copy_(solver._to_memory(P))
    solver.d_policy[:n][term] = 0


single = envs.make(env, bins, config=cfg, device=dev, transport=False)
seed(single)
n = single.n_states
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
single._evaluation_sweeps(2, gamma)                         # warm-up (also: code objects, clocks)
seed(single)
e0.record()
single._evaluation_sweeps(sweeps, gamma)
e1.record()
e1.synchronize()
whole_ms = e0.elapsed_time(e1) / sweeps
single._improvement_sweep(gamma)
torch.cuda.synchronize()
# the single-rank solver may hold its arrays in the class's fast memory order; sharded ranks keep the env's order
V_ref = single._to_user(single.d_value_function[:n]).clone()
P_ref = single._to_user(single.d_policy[:n]).clone()
changed_ref = int(single._d_changed.item())
delta_ref = float(single._d_delta.item())

group = f"logical-{uuid.uuid4().hex}"
out, errors = [None] * world, []


def rank_main(r):
    try:
        stream = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(stream):
            s = envs.make(env, bins, config=cfg, device=dev, transport=T.NativeTransport.local(r, world, group))
            seed(s)
            stream.synchronize()
            info = dict(s._comm.info)
            ranges = s._backend.engine.plan_ranges()
            s._evaluation_sweeps(sweeps, gamma)
            s._improvement_sweep(gamma)
            stream.synchronize()
            a, b = s._s_begin, s._s_end
            same_V = bool(torch.equal(s.d_value_function[a:b], V_ref[a:b]))
            same_P = bool(torch.equal(s.d_policy[a:b], P_ref[a:b]))
            ok_scalars = int(s._d_changed.item()) == changed_ref and float(s._d_delta.item()) == delta_ref
            out[r] = {"rank": r, "states": b - a, "mode": info["mode"], "reach_units": info["reach_units"],
                      "row_exact_plan": info.get("row_exact"),
                      "bytes_received_per_sweep": 4 * info["recv_elems"], "bytes_sent_per_sweep": 4 * info["send_elems"],
                      "ranges": ranges, "bit_identical_V": same_V, "bit_identical_policy": same_P,
                      "reduced_scalars_equal": ok_scalars, "_solver": s}
    except Exception as exc:  # noqa: BLE001
        errors.append((r, repr(exc)))


threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
for t in threads:
    t.start()
for t in threads:
    t.join(timeout=600)
if errors or any(o is None for o in out):
    print(json.dumps({"env": env, "bins": bins, "world": world, "errors": errors}))
    sys.exit(1)

# time split, each rank alone on the GPU: its swept-first launches, then its interior launches
for o in out:
    s = o.pop("_solver")
    eng = s._backend.engine
    st = torch.cuda.current_stream(dev).cuda_stream
    parts = {0: 0.0, 1: 0.0}
    later = {0: 0.0, 1: 0.0}
    for kind in (0, 1):
        rs = [(a, b) for k, a, b in o["ranges"] if k == kind]
        if not rs:
            continue
        term_ptr = s._backend._ptr(s._mask_arg())
        for rep in range(2):                                # first repetition warms up
            e0.record()
            for _ in range(5):                              # exactly the launches the sharded driver makes for this part
                eng.eval_sweep_part(s.d_value_function.data_ptr(), s.d_new_value_function.data_ptr(),
                                    s.d_policy.data_ptr(), term_ptr, kind, gamma, st)
            e1.record()
            e1.synchronize()
        parts[kind] = e0.elapsed_time(e1) / 5
        # the LATER sweeps of a batch (no terminal copies; over the live-state list where the handle holds one):
        # (a 9-sweep batch - a 1-sweep batch) / 8 per range
        def batches(k):
            e0.record()
            for a, b in rs:
                eng.eval_sweeps(s.d_value_function.data_ptr(), s.d_new_value_function.data_ptr(), s.d_policy.data_ptr(),
                                s.d_terminal_mask.data_ptr(), a, b, gamma, k, 0, st)
            e1.record()
            e1.synchronize()
            return e0.elapsed_time(e1)
        batches(2)
        later[kind] = (min(batches(9) for _ in range(2)) - min(batches(1) for _ in range(2))) / 8.0
    o["first_ms"], o["interior_ms"] = parts[0], parts[1]
    o["first_later_sweeps_ms"], o["interior_later_sweeps_ms"] = later[0], later[1]
    o["live_states_listed"] = eng.info(16)
    o["row_exact"] = bool(eng.comm_info(5) == 1)
    o["first_launches"] = sum(1 for k, _, _ in o["ranges"] if k == 0)
    o["interior_launches"] = sum(1 for k, _, _ in o["ranges"] if k == 1)
    o["first_states"] = sum(b - a for k, a, b in o["ranges"] if k == 0)
    o["interior_states"] = sum(b - a for k, a, b in o["ranges"] if k == 1)
    s._backend.close()

print(json.dumps({"env": env, "bins": bins, "world": world, "states": n, "label": "one GPU, logical ranks",
                  "whole_grid_eval_ms_single_rank": whole_ms, "sweeps_checked": sweeps,
                  "all_bit_identical": all(o["bit_identical_V"] and o["bit_identical_policy"] and o["reduced_scalars_equal"]
                                           for o in out),
                  "ranks": out}), flush=True)
