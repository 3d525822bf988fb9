import os, sys
sys.path.insert(0, ".")
os.environ["PI_MI355_XCD_TRACE"] = "1"
os.environ["PI_MI355_XCD_TIMING"] = "1"
from dynamicprogramming_amd import envs
for ring in sys.argv[1:] or ["128"]:
    os.environ["PI_MI355_XCD_RING"] = ring
    for name, bins in (("pendulum", 200),):
        s = envs.make(name, bins, device="cuda:0")
        e = s._backend.engine
        print("ring", ring, "info", [e.info(k) for k in (6, 19, 30, 31, 32)])
        d = s.policy_evaluation()
        print("delta", d, s.stats["sweeps_per_iter"], [e.info(k) for k in (30, 31, 32)])
