"""tools/fetch_calibration_identity.py — the product's evaluation sweep with a plugin whose successor is the state itself.

Second half of the FETCH_SIZE / WRITE_SIZE calibration (tools/fetch_calibration.hip is the first): `step_dynamics` returns
its own state, so the successor cell of grid node s is the cell at (or just below) s and the sweep's compulsory traffic is
KNOWN — every V value read once through the sweeps' own 8-byte corner-pair gathers (4 B per state), the policy entry once
through the 4-byte non-temporal stream (4 B), V' written once (4 B) — provided the far corners' lines stay in the XCD's
L2 until the states that own them come by.  That is a property of the grid's shape, so two 4-D shapes of the same 40.96 M
states are swept: (640, 40, 40, 40), whose slowest plane is 256 KB (stays), and the metric config's own 80^4, whose slowest
plane is 2 MB (the same reuse distance the real sweep has to live with), plus the 6-D 25^6 grid.  Run under
`rocprofv3 --pmc ...` by tools/fetch_calibration.sh (kernel name: pi_eval_sweep_kernel); prints one JSON line with the
known bytes per launch.
"""
import argparse, json, sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
import torch

from dynamicprogramming_amd import _native

ap = argparse.ArgumentParser()
ap.add_argument("--shapes", default="640,40,40,40;80,80,80,80;25,25,25,25,25,25")
ap.add_argument("--sweeps", type=int, default=6)
args = ap.parse_args()

out = []
dev = torch.device("cuda:0")
for spec in args.shapes.split(";"):
    shape = [int(v) for v in spec.split(",")]
    D = len(shape)
    names = [f"s{d}" for d in range(D)]
    src = ("__device__ void step_dynamics(" + ", ".join(f"float {v}" for v in names) + ", float a, "
           + ", ".join(f"float* n{d}" for d in range(D)) + ", float* r, bool* t) { "
           + " ".join(f"*n{d} = s{d};" for d in range(D)) + " *r = a; *t = false; }")
    bins = [np.linspace(-1.0, 1.0, g, dtype=np.float32) for g in shape]
    eng = _native.Engine(D, shape, [b.min() for b in bins], [b.max() for b in bins], bins, np.array([0.0, 1.0], np.float32),
                         device=0)
    eng.compile(src)
    n = int(np.prod(shape))
    V = torch.randn(n, dtype=torch.float32, device=dev)
    Vn = torch.empty_like(V)
    pol = torch.randint(0, 2, (n,), dtype=torch.int32, device=dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st = torch.cuda.current_stream().cuda_stream
    ms = []
    for i in range(args.sweeps):                  # one launch per call: pi_eval_sweep_kernel, no residual, no mask
        e0.record()
        eng.eval_sweep(V.data_ptr(), Vn.data_ptr(), pol.data_ptr(), 0, 0, n, 0.99, 0, st)
        e1.record()
        e1.synchronize()
        ms.append(e0.elapsed_time(e1))
        V, Vn = Vn, V
    out.append({"shape": shape, "states": n, "known_read_bytes": 8 * n, "known_write_bytes": 4 * n,
                "ms_per_sweep": min(ms[1:]), "threads_per_workgroup": eng.info(11), "chunks_per_workgroup": eng.info(3)})
    eng.close()
    del V, Vn, pol
print(json.dumps({"identity_sweeps": out, "sweeps_each": args.sweeps}))
