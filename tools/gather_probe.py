"""
tools/gather_probe.py — how much of the replay sweep's time is the SHAPE of the V gather?

Runs pi_eval_sweeps_cached on the 80^4 grid for synthetic 4-D dynamics whose displacement
pattern is controlled, and for the real double pendulum:
  identity   s' = s                      (every lane gathers its own cell: contiguous)
  shift      s' = s + const cells        (contiguous, displaced)
  shear2     q2' = q2 + dt*w2            (lanes of a wave spread over ~8 rows of dim 2)
  shear02    both angle dims sheared by their velocities (double-pendulum kinematics, no accel)
  pendulum   the real env
Prints ms per replay sweep.  Usage (GPU box): python tools/gather_probe.py [bins]
"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
import torch
from dynamicprogramming_amd import _native, envs

bins_n = int(sys.argv[1]) if len(sys.argv) > 1 else 80
only = [a for a in sys.argv[2:] if not a.startswith("policy=")]
pol_mode = ([a.split("=")[1] for a in sys.argv[2:] if a.startswith("policy=")] or ["random"])[0]
SIG = "float a, float b, float c, float d, float u, float* na, float* nb, float* nc, float* nd, float* r, bool* t"
DYN = {
    "identity": f"__device__ void step_dynamics({SIG}) {{ *na=a; *nb=b; *nc=c; *nd=d; *r=1.0f; *t=false; }}",
    "shift": f"__device__ void step_dynamics({SIG}) {{ *na=a+0.2f; *nb=b+1.0f; *nc=c+0.2f; *nd=d+1.0f; *r=1.0f; *t=false; }}",
    "shear2": f"__device__ void step_dynamics({SIG}) {{ *na=a; *nb=b; *nc=c+0.02f*d; *nd=d; *r=1.0f; *t=false; }}",
    "shear02": f"__device__ void step_dynamics({SIG}) {{ *na=a+0.02f*b; *nb=b; *nc=c+0.02f*d; *nd=d; *r=1.0f; *t=false; }}",
    "pendulum": envs.dynamics_source("double_pendulum_swingup"),
}
cls = envs.ENVS["double_pendulum_swingup"]
tables = [np.asarray(b, np.float32) for b in cls.bins_space(bins_n).values()]
n = bins_n ** 4
dev = torch.device("cuda:0")
gen = torch.Generator(device="cpu").manual_seed(0)
V = torch.randn(n, generator=gen).to(dev)
Vb = torch.empty_like(V)
pol = torch.randint(0, 11, (n,), generator=gen, dtype=torch.int32).to(dev)
if pol_mode == "const":
    pol.fill_(5)
elif pol_mode == "blocks":      # piecewise-constant policy: action changes every 8 cells of dim 1
    idx = torch.arange(n, device=dev)
    pol = ((idx // (bins_n ** 2 * 8)) % 11).to(torch.int32)
term = torch.zeros(n, dtype=torch.uint8, device=dev)
for name, dyn in DYN.items():
    if only and name not in only:
        continue
    eng = _native.Engine(4, [bins_n] * 4, [t.min() for t in tables], [t.max() for t in tables], tables,
                         cls.ACTIONS, device=0)
    eng.compile(dyn)
    need = eng.transition_cache_bytes(0, n)
    cache = torch.empty(need, dtype=torch.uint8, device=dev)
    eng.eval_sweeps_cached(V.data_ptr(), Vb.data_ptr(), pol.data_ptr(), term.data_ptr(), 0, n, 0.999, 3, True,
                           cache.data_ptr(), need)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    eng.eval_sweeps_cached(V.data_ptr(), Vb.data_ptr(), pol.data_ptr(), term.data_ptr(), 0, n, 0.999, 20, False,
                           cache.data_ptr(), need)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    e0.record()
    eng.eval_sweeps(V.data_ptr(), Vb.data_ptr(), pol.data_ptr(), term.data_ptr(), 0, n, 0.999, 10)
    e1.record()
    torch.cuda.synchronize()
    ms2 = e0.elapsed_time(e1) / 10
    print(f"[{pol_mode}] tiled={eng.info(10)} box={[eng.info(20 + d) for d in range(4)]} ", end="")
    print(f"{name:10s} replay {ms:.3f} ms/sweep   recompute {ms2:.3f} ms/sweep   ({n / ms / 1e6:.1f} G states/s replay)")
    eng.close()
