"""tools/time_construction.py — wall time of constructing a solver on the big grids (GPU box): with the mask from the bin
tables (`_terminal_fn_axes`, built on the device) and with the reference's `_terminal_fn(states_space)` hook, which
materialises the (n, D) grid on the host (what every plugin written for the reference gets)."""
import json, sys, time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch

from dynamicprogramming_amd import envs

for name, bins in (("double_cartpole", 25), ("double_cartpole_swingup", 25), ("cartpole_swingup", 50)):
    cls = envs.ENVS[name]
    row = {"env": name, "bins": bins}
    for label, klass in (("bin_tables", cls), ("reference_hook", type(cls.__name__ + "RefHook", (cls,), {"_terminal_fn": cls._terminal_fn}))):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        s = klass(cls.bins_space(bins), cls.ACTIONS, envs.CudaPIConfig(**cls.CONFIG), device="cuda:0")
        torch.cuda.synchronize()
        row[label] = {"seconds": time.perf_counter() - t0, "live_list": s._backend.engine.info(16),
                      "states_space_materialised": s._states_space is not None}
        s._backend.close()
        del s
    print(json.dumps(row), flush=True)
