"""tools/tiled_list_ab.py --env E --bins B [--blocks 3,5,8] — does a TILED traversal order of the live-state list cut the
L2 misses of the 6-D sweeps, and does that buy time?  (GPU box; experiment behind pi_set_live_order.)

The list sweeps take lane k = state list[k]; the list is ascending, i.e. the sweep walks memory dimension 0, then 1, then
2 ...  A successor cell spans two planes of EVERY dimension: the pair along dimension 2 (62 KB apart on 25^6) is re-read
while it is still in L2, the pairs along dimensions 1 (1.6 MB) and 0 (39 MB) are not — V is fetched 11x per sweep past L2
on the double-cartpole swing-up grid (profiles/r06/counters_bench_c5_swingup.json).  Variant "tile B": the list sorted by
(i0, i2 // B, i1, i2 % B, i3, i4, i5) — for a block of B rows of dimension 2 the walk runs along dimension 1 first, so that
the two sub-planes a block shares with its successor along dimension 1 are still in L2 (4 blocks of targets x (B + halo)
rows must fit it).  Times evaluation / improvement sweeps per variant on the bench state, checks that V' is bit-identical
to the ascending list's, one JSON line per variant.  PI_MI355_STRIP (environment) picks the workgroup -> XCD schedule."""
import argparse, json, os, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
import torch
from dynamicprogramming_amd import envs

ap = argparse.ArgumentParser()
ap.add_argument("--env", default="double_cartpole_swingup")
ap.add_argument("--bins", type=int, default=25)
ap.add_argument("--blocks", default="3,5,8,13")
ap.add_argument("--sweeps", type=int, default=10)
args = ap.parse_args()
cls = envs.ENVS[args.env]
s = envs.make(args.env, args.bins, config=envs.CudaPIConfig(**cls.CONFIG), device="cuda:0")
eng = s._backend.engine
n, nA, D = s.n_states, s.n_actions, cls._D
gamma = float(np.float32(s.config.gamma))
order = eng.order
mem_shape = [int(s.grid_shape[d]) for d in order]
stride = [int(np.prod(mem_shape[k + 1:])) for k in range(D)]
m = eng.live_list()
assert m > 0, "this grid keeps no live-state list"
asc = torch.empty(m, dtype=torch.int32, device="cuda:0")
eng.live_list(asc.data_ptr(), m)
torch.cuda.synchronize()


def tiled(B):
    x = asc.to(torch.int64)
    i0, i1, i2, rest = x // stride[0], (x // stride[1]) % mem_shape[1], (x // stride[2]) % mem_shape[2], x % stride[2]
    nb = -(-mem_shape[2] // B)
    key = ((((i0 * nb + i2 // B) * mem_shape[1] + i1) * B + i2 % B) * stride[2]) + rest
    return asc[torch.argsort(key)].contiguous()


def reset():
    gen = torch.Generator(device="cpu").manual_seed(0)
    s.d_value_function[:n].copy_(torch.randn(n, generator=gen, dtype=torch.float32))
    s.d_policy[:n].copy_(torch.randint(0, nA, (n,), generator=gen, dtype=torch.int32))
    t = s.d_terminal_mask[:n].bool()
    s.d_value_function[:n][t] = 0.0
    s.d_new_value_function.copy_(s.d_value_function)


def measure(label, lst):
    if lst is not None:
        eng.set_live_order(lst.data_ptr(), m)
    reset()
    s._evaluation_sweeps(args.sweeps + 1, gamma)
    torch.cuda.synchronize()
    digest = int(s.d_value_function[:n].view(torch.int32).to(torch.int64).sum().item())
    ms = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        s._evaluation_sweeps(args.sweeps, gamma)
        e1.record()
        e1.synchronize()
        ms.append(e0.elapsed_time(e1) / args.sweeps)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    s._improvement_sweep(gamma)
    s._improvement_sweep(gamma)
    e1.record()
    e1.synchronize()
    out = {"env": args.env, "bins": args.bins, "strip": os.environ.get("PI_MI355_STRIP", "auto"), "strip_states": eng.info(35),
           "variant": label, "eval_ms": ms, "improve_ms": e0.elapsed_time(e1) / 2, "V_digest": digest,
           "changed": int(s._d_changed.item())}
    print(json.dumps(out), flush=True)
    return digest


base = measure("ascending", None)
for B in [int(v) for v in args.blocks.split(",") if v]:
    d = measure(f"tile {B}", tiled(B))
    assert d == base, f"tile {B}: V differs from the ascending list's"
measure("ascending again", asc)
