"""tools/ab_lane_argmax.py — A/B for the improvement sweep's argmax (GPU box; experiment, not product code).

north_star suggests a "wavefront-shuffle argmax": spread the actions of a state over lanes and reduce
with shuffles.  The product kernel (pi_improve_sweep_kernel) instead loops over the actions in ONE lane,
because everything in step_dynamics that does not depend on the action is then hoisted out of the loop
by the compiler.  This script measures both on the same inputs:

  A  product kernel: one lane per state, serial action loop.
  B  experimental kernel, built here from the product's own translation unit (`pi_kernel_source`) plus the
     text below: L = next power of two >= n_actions lanes per state, every lane runs the full backup
     for its action, then log2(L) shuffle steps pick the first maximum (same tie rule: lowest index wins,
     NaN never wins).  Loaded with hipModuleLoad through ctypes; policy output must equal A's.

usage: python tools/ab_lane_argmax.py [env] [bins]     (default: double_pendulum_swingup 80)
"""
import ctypes
import json
import subprocess
import sys
import tempfile
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
import torch

from dynamicprogramming_amd import envs

AB_KERNEL = r'''
// ---- experiment: lane-parallel argmax (tools/ab_lane_argmax.py) ----
extern "C" __global__ void __launch_bounds__(256)
ab_improve_lanes_kernel(const float* __restrict__ V, int* __restrict__ policy,
                        const unsigned char* __restrict__ term, const float* __restrict__ tab,
                        long long n, float gamma, unsigned int* __restrict__ changed) {
    __shared__ float lds_tab[PI_GRID.tab_len];
    pi_stage_table<256>(tab, lds_tab);
    __syncthreads();
    constexpr int kPer = 256 / AB_L;                       // states per workgroup
    const long long s = (long long)blockIdx.x * kPer + threadIdx.x / AB_L;
    const int a = threadIdx.x % AB_L;
    const long long sc = s < n ? s : n - 1;
    const bool live = s < n && !term[sc];
    float q = -1.0e30f;
    int best = a;
    if (live && a < PI_NA) {
        float x[PI_D];
        pi_state_coords((unsigned int)sc, lds_tab, x);
        const float qq = pi_backup(x, lds_tab[PI_TAB_ACT + a], V, gamma);
        q = qq > -1.0e30f ? qq : -1.0e30f;                 // the serial loop starts from -1e30 with a strict '>'
    }
#pragma unroll
    for (int o = AB_L / 2; o > 0; o >>= 1) {
        const float oq = __shfl_xor(q, o, 64);
        const int oa = __shfl_xor(best, o, 64);
        if (oq > q || (oq == q && oa < best)) { q = oq; best = oa; }
    }
    unsigned int ch = 0;
    if (live && a == 0) {
        ch = policy[s] != best ? 1u : 0u;
        policy[s] = best;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ch += __shfl_xor(ch, o, 64);
    if ((threadIdx.x & 63) == 0 && ch) atomicAdd(changed, ch);
}
'''


def main():
    env = sys.argv[1] if len(sys.argv) > 1 else "double_pendulum_swingup"
    bins = int(sys.argv[2]) if len(sys.argv) > 2 else 80
    cls = envs.ENVS[env]
    solver = envs.make(env, bins, device="cuda:0")
    eng = solver._backend.engine
    n, nA = solver.n_states, solver.n_actions
    L = 1
    while L < nA:
        L *= 2
    gamma = float(np.float32(solver.config.gamma))
    gen = torch.Generator(device="cpu").manual_seed(0)
    solver.d_value_function[:n].copy_(torch.randn(n, generator=gen, dtype=torch.float32))
    V = solver.d_value_function
    term = solver.d_terminal_mask
    tmp = Path(tempfile.mkdtemp(prefix="ab_lane_"))
    src = tmp / "ab.hip"
    src.write_text(f"#define AB_L {L}\n" + eng.kernel_source(envs.dynamics_source(env)) + AB_KERNEL)
    # AB_L must be visible inside the appended kernel only; the define in front is harmless for the template
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-std=c++17", "--genco",
                    "-include", "hip/hip_runtime.h", str(src), "-o", str(tmp / "ab.hsaco")], check=True)
    hip = ctypes.CDLL("libamdhip64.so")
    mod, fn = ctypes.c_void_p(), ctypes.c_void_p()
    assert hip.hipModuleLoad(ctypes.byref(mod), str(tmp / "ab.hsaco").encode()) == 0
    assert hip.hipModuleGetFunction(ctypes.byref(fn), mod, b"ab_improve_lanes_kernel") == 0
    user_tables = [np.asarray(b, np.float32) for b in cls.bins_space(bins).values()]
    d_tab = torch.from_numpy(np.concatenate([cls.ACTIONS.astype(np.float32)] +
                                            [user_tables[d] for d in eng.order])).cuda()        # the engine's memory order
    pol_a = torch.zeros(n, dtype=torch.int32, device="cuda:0")
    pol_b = torch.zeros(n, dtype=torch.int32, device="cuda:0")
    ch_a = torch.zeros(1, dtype=torch.int32, device="cuda:0")
    ch_b = torch.zeros(1, dtype=torch.int32, device="cuda:0")

    def run_a():
        eng.improve_sweep(V.data_ptr(), pol_a.data_ptr(), term.data_ptr(), 0, n, gamma, ch_a.data_ptr(),
                          torch.cuda.current_stream().cuda_stream)

    per = 256 // L
    blocks = (n + per - 1) // per
    vals = [ctypes.c_void_p(V.data_ptr()), ctypes.c_void_p(pol_b.data_ptr()), ctypes.c_void_p(term.data_ptr()),
            ctypes.c_void_p(d_tab.data_ptr()), ctypes.c_longlong(n), ctypes.c_float(gamma),
            ctypes.c_void_p(ch_b.data_ptr())]
    args = (ctypes.c_void_p * len(vals))(*[ctypes.cast(ctypes.byref(v), ctypes.c_void_p) for v in vals])

    def run_b():
        rc = hip.hipModuleLaunchKernel(fn, ctypes.c_uint(blocks), 1, 1, 256, 1, 1, 0,
                                       ctypes.c_void_p(torch.cuda.current_stream().cuda_stream), args, None)
        assert rc == 0, rc

    def timed(f, reps):
        f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            f()
        e1.record()
        e1.synchronize()
        return e0.elapsed_time(e1) / reps

    ms_a = timed(run_a, 5)
    ch_b.zero_()
    ms_b = timed(run_b, 3)
    same = bool(torch.equal(pol_a, pol_b))
    print(json.dumps({"env": env, "bins": bins, "states": n, "actions": nA, "lanes_per_state": L,
                      "A_serial_loop_ms": ms_a, "B_lane_parallel_ms": ms_b, "policies_identical": same,
                      "ratio_B_over_A": ms_b / ms_a}))


if __name__ == "__main__":
    main()
