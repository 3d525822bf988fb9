"""tools/nccl_smoke.py — RCCL sanity on whatever ranks are available (also world size 1): the
collective forms the multi-rank solver uses (in-place all_gather_into_tensor of a slice, uint8
all_gather, float MAX / int32 SUM all_reduce).  Launch with torch.distributed.run."""
import os, torch, torch.distributed as dist
rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", 0)))
dev = torch.device("cuda", torch.cuda.current_device())
dist.init_process_group("nccl", device_id=dev)
per = 1000
full = torch.full((per * world,), -1.0, device=dev)
full[rank * per:(rank + 1) * per] = float(rank)
dist.all_gather_into_tensor(full, full[rank * per:(rank + 1) * per])
assert all(float(full[r * per]) == r and float(full[(r + 1) * per - 1]) == r for r in range(world))
bits = torch.zeros(world * 80, dtype=torch.uint8, device=dev)
dist.all_gather_into_tensor(bits, torch.full((80,), rank + 1, dtype=torch.uint8, device=dev))
assert int(bits[-1]) == world
d = torch.tensor([float(rank)], device=dev); dist.all_reduce(d, op=dist.ReduceOp.MAX)
c = torch.tensor([rank + 1], dtype=torch.int32, device=dev); dist.all_reduce(c, op=dist.ReduceOp.SUM)
assert float(d) == world - 1 and int(c) == world * (world + 1) // 2
if world > 1:
    nxt, prv = (rank + 1) % world, (rank - 1) % world
    buf = torch.zeros(per * world, device=dev)
    buf[rank * per:(rank + 1) * per] = rank + 10.0
    ops = [dist.P2POp(dist.isend, buf[rank * per:(rank + 1) * per], nxt),
           dist.P2POp(dist.irecv, buf[prv * per:(prv + 1) * per], prv)]
    for r in dist.batch_isend_irecv(ops):
        r.wait()
    torch.cuda.synchronize()
    assert float(buf[prv * per]) == prv + 10.0
dist.barrier()
if rank == 0:
    print(f"nccl smoke ok, world={world}")
dist.destroy_process_group()
