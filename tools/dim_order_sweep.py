"""tools/dim_order_sweep.py ENV BINS [perm ...] — does another memory / traversal order of the grid's dimensions make the
evaluation sweep cheaper?  (GPU box; experiment, not product code.)

The kernels sweep states in flat (row-major) order and an XCD's 4 MiB L2 sees one advancing window of V.  A value is
re-read by every state whose successor cell contains it; whether those re-reads hit L2 depends on which dimensions are
the SLOW ones of the order.  On the double-cartpole swing-up grid (25^6 x 9 actions) the L2-side traffic of the evaluation
sweep is 4.4 x the compulsory bytes (profiles/r03/counters_bench_c5_swingup.json).  This tool measures the alternative
orders directly, with the product's own kernels: it builds the SAME env with its dimensions permuted — bin tables in the
new order and a wrapper around the plugin's step_dynamics that un-permutes the arguments — so both the layout of V in HBM
and the order the sweep walks it change, and nothing else does.  Per permutation: evaluation / improvement sweep time on
the bench state (V ~ N(0,1), random policy, 2 bench steps), and a checksum showing the permuted problem IS the same problem
(sum and max of V' equal up to summation order).

perm: comma-separated dimension order, new dimension k = old dimension perm[k]; default: a set of candidates for D = 6.
Run under `rocprofv3 --pmc FETCH_SIZE WRITE_SIZE` for the traffic of each (one process per permutation: `--one`).
"""
import json
import re
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
import torch

from dynamicprogramming_amd import envs
from dynamicprogramming_amd.solver import CudaPolicyIteration2D, CudaPolicyIteration4D, CudaPolicyIteration6D

env, bins = sys.argv[1], int(sys.argv[2])
cls = envs.ENVS[env]
D = cls._D
perms = [tuple(int(v) for v in a.split(",")) for a in sys.argv[3:] if re.fullmatch(r"[0-9,]+", a)]
if not perms:
    perms = {6: [(0, 1, 2, 3, 4, 5), (0, 2, 1, 3, 4, 5), (0, 1, 3, 2, 4, 5), (1, 0, 2, 3, 4, 5), (0, 2, 3, 1, 4, 5),
                 (0, 1, 2, 3, 5, 4), (2, 3, 0, 1, 4, 5), (0, 1, 4, 5, 2, 3)],
             4: [(0, 1, 2, 3), (1, 0, 2, 3), (0, 2, 1, 3), (0, 1, 3, 2), (2, 3, 0, 1)]}.get(D, [tuple(range(D))])
base = {2: CudaPolicyIteration2D, 4: CudaPolicyIteration4D, 6: CudaPolicyIteration6D}[D]
dyn = envs.dynamics_source(env)
keys = list(cls.bins_space(bins).keys())
tables = list(cls.bins_space(bins).values())


def permuted_class(perm):
    inv = [perm.index(d) for d in range(D)]                   # old dimension d is new dimension inv[d]
    p = ", ".join(f"float p{k}" for k in range(D))
    n = ", ".join(f"float* n{k}" for k in range(D))
    call_s = ", ".join(f"p{inv[d]}" for d in range(D))
    call_n = ", ".join(f"n{inv[d]}" for d in range(D))
    wrapper = (f"\n#define step_dynamics env_step_dynamics_original\n{dyn}\n#undef step_dynamics\n"
               f"__device__ void step_dynamics({p}, float act, {n}, float* rew, bool* done) {{\n"
               f"    env_step_dynamics_original({call_s}, act, {call_n}, rew, done);\n}}\n")

    class Permuted(base):
        def _dynamics_cuda_src(self):
            return wrapper

        def _terminal_fn(self, states):
            inst = object.__new__(cls)
            if env == "overhead_crane":
                inst.target_x = 0.0
            for attr in ("_TH_LIMIT", "_TH_FAIL", "_RAIL"):
                if hasattr(cls, attr):
                    setattr(inst, attr, getattr(cls, attr))

            class Columns:                                     # column d of the ORIGINAL order, without copying 6 GB
                def __getitem__(self, key):
                    rows, col = key
                    return states[rows, inv[col]]

                def __len__(self):
                    return len(states)
            return cls._terminal_fn(inst, Columns())
    if cls._terminal_fn is base._terminal_fn:
        del Permuted._terminal_fn
    return Permuted


out = []
for perm in perms:
    P = permuted_class(perm)
    s = P({keys[perm[k]]: tables[perm[k]] for k in range(D)}, cls.ACTIONS, envs.CudaPIConfig(**cls.CONFIG), device="cuda:0")
    n, nA = s.n_states, s.n_actions
    gamma = float(np.float32(s.config.gamma))
    # the same problem: seed V / policy in the ORIGINAL order and permute them into this layout
    gen = torch.Generator(device="cpu").manual_seed(0)
    shape = [len(t) for t in tables]
    V0 = torch.randn(n, generator=gen, dtype=torch.float32).reshape(shape).permute(*perm).contiguous().reshape(-1)
    P0 = torch.randint(0, nA, (n,), generator=gen, dtype=torch.int32).reshape(shape).permute(*perm).contiguous().reshape(-1)
    s.d_value_function[:n].copy_(V0)
    s.d_new_value_function.copy_(s.d_value_function)
    s.d_policy[:n].copy_(P0)
    del V0, P0
    for _ in range(2):
        s._evaluation_sweeps(10, gamma)
        s._improvement_sweep(gamma)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ms = []
    for _ in range(3):
        e0.record()
        s._evaluation_sweeps(10, gamma)
        e1.record()
        e1.synchronize()
        ms.append(e0.elapsed_time(e1) / 10)
    e0.record()
    s._improvement_sweep(gamma)
    e1.record()
    e1.synchronize()
    imp = e0.elapsed_time(e1)
    V = s.d_value_function[:n]
    row = {"perm": list(perm), "dims": [keys[d] for d in perm], "eval_ms": min(ms), "improve_ms": imp,
           "checksum_sum": float(V.double().sum().item()), "checksum_max": float(V.max().item()),
           "residual": float(s._d_delta.item()), "changed": int(s._d_changed.item()), "live_list": s._backend.engine.info(16)}
    out.append(row)
    print(json.dumps(row), flush=True)
    s._backend.close()
    del s
    torch.cuda.empty_cache()
print(json.dumps({"env": env, "bins": bins, "rows": out}))
