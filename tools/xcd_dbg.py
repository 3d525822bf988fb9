import os, sys, time
sys.path.insert(0, ".")
os.environ["PI_MI355_XCD_TRACE"] = "1"
os.environ["PI_MI355_XCD_TIMING"] = "1"
import torch
from dynamicprogramming_amd import envs
for name, bins in (("pendulum", 200),):
    s = envs.make(name, bins, device="cuda:0")
    e = s._backend.engine
    print("info", [e.info(k) for k in (6, 19, 30, 31, 32, 33)])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    s.run()
    print("seconds", time.perf_counter() - t0, s.stats["pi_iterations"], s.stats["eval_sweeps"], s.stats["stable"], s.stats["sweeps_per_iter"][:5])
