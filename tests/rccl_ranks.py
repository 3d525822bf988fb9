"""tests/rccl_ranks.py — one RANK of a multi-GPU RCCL check, started by tests/test_gpu_rccl.py through
`python -m torch.distributed.run --nproc-per-node N tests/rccl_ranks.py` (fresh processes: nothing here has touched a
GPU before it is started; test infrastructure, not part of the package).

Every rank solves small grids twice — sharded over the N ranks through the library's RCCL transport (communicator, exchange
plan, sharded evaluation batches, scalar all-reduces, final all-gathers: csrc/pi_comm.cpp) and alone — and demands the same
bits: a C4-shaped 4-D grid (wrapping angles, no terminal states) and a 25^6-shaped odd 6-D grid (terminal states, n not a
multiple of the rank count: padded shards).  PI_MI355_EXCHANGE (halo | allgather) comes from the test.  Prints one JSON line
per rank; exit code 0 only when everything was identical and the communicator says what the launcher said.
"""
import json
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
import torch
import torch.distributed as dist

from dynamicprogramming_amd import envs

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", 0)))
dev = torch.device("cuda", torch.cuda.current_device())
dist.init_process_group("nccl", device_id=dev)
report = {"rank": rank, "world": world, "exchange": os.environ.get("PI_MI355_EXCHANGE", "auto"), "cases": []}
ok = True
for name, shape in (("double_pendulum_swingup", (16, 10, 12, 10)), ("double_cartpole", (5, 5, 5, 5, 5, 5))):
    cls = envs.ENVS[name]
    keys = list(cls.bins_space(2).keys())
    bins = {k: np.asarray(list(cls.bins_space(int(g)).values())[d], np.float32) for d, (k, g) in enumerate(zip(keys, shape))}
    cfg = dict(cls.CONFIG, max_pi_iter=3, max_eval_iter=60)
    sharded = cls(bins, cls.ACTIONS, envs.CudaPIConfig(**cfg), device=dev)
    eng = sharded._backend.engine
    comm = {"world": eng.comm_info(1), "transport": {1: "rccl", 2: "in-process", 3: "p2p"}.get(eng.comm_info(2), "none"),
            "rank": eng.comm_info(0), "plan": dict(sharded._comm.info) if sharded._comm is not None else None}
    sharded.run()
    alone = cls(bins, cls.ACTIONS, envs.CudaPIConfig(**cfg), device=dev, transport=False)
    alone.run()
    same = bool(np.array_equal(sharded.value_function.view(np.uint32), alone.value_function.view(np.uint32))
                and np.array_equal(sharded.policy, alone.policy)
                and sharded.stats["sweeps_per_iter"] == alone.stats["sweeps_per_iter"])
    right = comm["world"] == world and comm["transport"] == "rccl" and comm["rank"] == rank
    ok = ok and same and right
    report["cases"].append({"env": name, "shape": list(shape), "states": int(np.prod(shape)), "identical": same,
                            "comm": comm, "sweeps": sharded.stats["sweeps_per_iter"]})
report["ok"] = ok
print("RCCL_RANK " + json.dumps(report), flush=True)
torch.cuda.synchronize()
dist.barrier()
dist.destroy_process_group()
sys.exit(0 if ok else 1)
