"""tools/rccl_smoke.py — the library's RCCL path (csrc/pi_comm.cpp) on whatever ranks are available,
world size 1 included:  python -m torch.distributed.run --nproc-per-node N tools/rccl_smoke.py

Every rank solves a small pendulum grid twice — sharded over the N ranks through
NativeTransport (communicator, exchange plan, sharded evaluation batches, scalar all-reduces,
final all-gathers, all inside libpi_mi355.so) and alone — and checks that both give the same
bits.  torch.distributed is only used to hand the 128-byte RCCL id around."""
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
import torch
import torch.distributed as dist

from dynamicprogramming_amd import envs

rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
# PI_SMOKE_ONE_GPU=1: every rank on device 0 (rehearsal on a one-GPU box, if RCCL accepts it)
torch.cuda.set_device(0 if os.environ.get("PI_SMOKE_ONE_GPU") == "1" else int(os.environ.get("LOCAL_RANK", 0)))
dev = torch.device("cuda", torch.cuda.current_device())
dist.init_process_group("nccl", device_id=dev)
cls = envs.ENVS["pendulum"]
cfg = dict(cls.CONFIG, max_pi_iter=3, max_eval_iter=101)
sharded = cls(cls.bins_space(96), cls.ACTIONS, envs.CudaPIConfig(**cfg), device=dev)
plan = dict(sharded._comm.info) if sharded._comm is not None else None
sharded.run()
alone = cls(cls.bins_space(96), cls.ACTIONS, envs.CudaPIConfig(**cfg), device=dev, transport=False)
alone.run()
same = (np.array_equal(sharded.value_function.view(np.uint32), alone.value_function.view(np.uint32))
        and np.array_equal(sharded.policy, alone.policy)
        and sharded.stats["sweeps_per_iter"] == alone.stats["sweeps_per_iter"])
print(f"rank {rank}/{world}: plan={plan} sweeps={sharded.stats['sweeps_per_iter']} identical={same}", flush=True)
dist.destroy_process_group()
sys.exit(0 if same else 1)
