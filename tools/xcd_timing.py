"""tools/xcd_timing.py [env@bins ...] — cycles per phase of the XCD-local kernel (GPU box; diagnostics).

Builds pi_xcd_kernel with its cycle counters (PI_MI355_XCD_TIMING), runs run() of a launch-bound 2-D grid in one launch and
prints what the library's trace (PI_MI355_XCD_TRACE) reports for the first and the last workgroup: cycles per sweep in the
gather, in the backup + store, the barriers' share, polls per sweep x 1000.  Other knobs are taken from the environment
(PI_MI355_XCD_FIRST_SLEEP, PI_MI355_XCD_RING).
"""
import os, sys, time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
os.environ["PI_MI355_XCD_TRACE"] = "1"
os.environ["PI_MI355_XCD_TIMING"] = "1"
import torch

from dynamicprogramming_amd import envs

cases = [(a.split("@")[0], int(a.split("@")[1])) for a in sys.argv[1:]] or [("pendulum", 200)]
for name, bins in cases:
    best = None
    for _ in range(3):
        s = envs.make(name, bins, device="cuda:0")
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        s.run()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    print(f"{name}@{bins}: run() best of 3 {best * 1e3:.2f} ms, {s.stats['eval_sweeps']} sweeps, {s.stats['pi_iterations']} rounds", flush=True)
