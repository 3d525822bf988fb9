"""tools/infer_bench.py — throughput of the batched inference kernel on one MI355X.

For the 2-D / 4-D / 6-D BASELINE grids (pendulum 200^2, double pendulum 80^4, double cartpole 25^6): a random
policy table and uniformly random query points resident in HBM, `pi_infer_query` timed with events on the launch
stream (action only, and weights + indices), against the numpy twin of utils/barycentric.py on the host (one
thread, 2^16 points).  Prints one JSON object per grid; HBM-side bytes per query are the compulsory ones
(4 D in, 4 out, and for the policy gather one 64-byte sector per corner PAIR, corners being adjacent along the
last dimension: 2^(D-1) x 64 B — random points share no lines).
usage: python tools/infer_bench.py [--log2-batch 22] [--repeat 20]
"""
import argparse
import json
import sys
import time
from itertools import product
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
import torch

from dynamicprogramming_amd import _native, envs
from utils import barycentric as B


def grid_of(env, bins):
    cls = envs.ENVS[env]
    tabs = [np.asarray(b, np.float32) for b in cls.bins_space(bins).values()]
    shape = np.array([len(t) for t in tabs], np.int32)
    lo = np.array([t.min() for t in tabs], np.float32)
    hi = np.array([t.max() for t in tabs], np.float32)
    strides = np.array([int(np.prod(shape[d + 1:])) for d in range(len(shape))], np.int32)
    return cls, lo, hi, shape, strides


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log2-batch", type=int, default=22)
    ap.add_argument("--repeat", type=int, default=20)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(0)
    for env, bins in (("pendulum", 200), ("double_pendulum_swingup", 80), ("double_cartpole", 25)):
        cls, lo, hi, shape, strides = grid_of(env, bins)
        D = len(shape)
        n = int(np.prod(shape.astype(np.int64)))
        bits = np.array(list(product([0, 1], repeat=D)), dtype=np.int32)
        acts = np.asarray(cls.ACTIONS, np.float32)
        policy = rng.integers(0, len(acts), size=n, dtype=np.int32)
        m = 1 << args.log2_batch
        pts = (lo + (hi - lo) * rng.random((m, D), dtype=np.float32)).astype(np.float32)
        eng = _native.InferenceEngine(lo, hi, shape, strides, bits, device=0)
        eng.set_policy(policy, acts)
        d_pts = torch.from_numpy(pts).to(dev)
        d_act = torch.empty(m, dtype=torch.float32, device=dev)
        d_w = torch.empty((m, 1 << D), dtype=torch.float32, device=dev)
        d_idx = torch.empty((m, 1 << D), dtype=torch.int32, device=dev)
        st = torch.cuda.current_stream(dev).cuda_stream

        def timed(**kw):
            for _ in range(3):
                eng.query(d_pts.data_ptr(), m, stream=st, **kw)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(args.repeat):
                eng.query(d_pts.data_ptr(), m, stream=st, **kw)
            b.record()
            torch.cuda.synchronize()
            return a.elapsed_time(b) / args.repeat

        ms_act = timed(d_actions=d_act.data_ptr())
        ms_wi = timed(d_weights=d_w.data_ptr(), d_indices=d_idx.data_ptr())
        # host twin on a sample (and a spot check of the device result against it)
        k = 1 << 16
        t0 = time.perf_counter()
        w, idx = B.get_barycentric_weights_and_indices(pts[:k], lo, hi, shape, strides, bits)
        a_host = (w * acts[policy[idx]]).astype(np.float32)
        t_host = time.perf_counter() - t0
        eng.query(d_pts.data_ptr(), m, d_weights=d_w.data_ptr(), d_indices=d_idx.data_ptr(), stream=st)
        torch.cuda.synchronize()
        same = bool(np.array_equal(d_w[:k].cpu().numpy().view(np.uint32), w.view(np.uint32))
                    and np.array_equal(d_idx[:k].cpu().numpy(), idx))
        bytes_act = 4 * D + 4 + (1 << (D - 1)) * 64
        bytes_wi = 4 * D + 8 * (1 << D)
        print(json.dumps({
            "grid": f"{env} {bins}^{D}", "batch": m,
            "action_only": {"ms": round(ms_act, 4), "queries_per_s": m / ms_act * 1e3,
                            "compulsory_bytes_per_query": bytes_act,
                            "hbm_frac_of_8TBps": bytes_act * m / ms_act * 1e3 / 8e12},
            "weights_and_indices": {"ms": round(ms_wi, 4), "queries_per_s": m / ms_wi * 1e3,
                                    "compulsory_bytes_per_query": bytes_wi,
                                    "hbm_frac_of_8TBps": bytes_wi * m / ms_wi * 1e3 / 8e12},
            "numpy_twin_one_thread": {"points": k, "queries_per_s": k / t_host},
            "device_equals_twin_on_sample": same}), flush=True)
        assert same
        eng.close()
        del d_pts, d_act, d_w, d_idx


if __name__ == "__main__":
    main()
