"""tools/ab_eval_replay.py — A/B for policy-EVALUATION sweeps: recompute vs replay (GPU box; experiment).

Under a fixed policy the successor of a state — step_dynamics(s, pi(s)) -> (s', reward, done) — is the same in every
sweep of the evaluation, and an evaluation runs thousands of sweeps.  The product recomputes it each sweep (~400 VALU
instructions per state on the double pendulum, the longest phase of a wave's life: profiles/r04/phase_timeline_c4.txt).
Variant B computes it ONCE per policy into a record in HBM (s' as D floats + the reward: 20 B per state in 4-D, 32 B in
6-D — 0.8 GB at 80^4, 7.8 GB at 25^6, of 288 GB) and every sweep then streams the record instead of the policy entry,
runs the cell search, the 2^D-corner gather and the fmaf chain as before — same operations on the same values, so V' is
the product's bit for bit (checked) — with no LDS table, no barrier and ~100 instead of ~400 VALU instructions per state.
Costs 16-28 more HBM bytes per state and sweep.

  A  product kernel through the C ABI (pi_eval_sweep)
  B  prepare once (ab_prepare_kernel), then ab_replay_kernel per sweep; built from the product's translation unit

usage: python tools/ab_eval_replay.py [env] [bins] [threads x cpw,...]   (default double_pendulum_swingup 80 1024x2,512x2,256x2,256x4)
"""
import ctypes
import json
import re
import subprocess
import sys
import tempfile
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
import torch

from dynamicprogramming_amd import envs

AB_KERNELS = r'''
// ---- experiment: successor records (tools/ab_eval_replay.py) ----
// record of state s: ns[0..3] in recA[s]; D == 4: reward in recR[s]; D == 6: (ns[4], ns[5], reward, done) in recB[s]
typedef float AbF4 __attribute__((ext_vector_type(4)));
extern "C" __global__ void __launch_bounds__(256)
ab_prepare_kernel(const int* __restrict__ policy, const unsigned char* __restrict__ term, const float* __restrict__ tab,
                  long long n, AbF4* __restrict__ recA, AbF4* __restrict__ recB, float* __restrict__ recR,
                  unsigned char* __restrict__ flag) {
    __shared__ float lds_tab[PI_GRID.tab_len];
    pi_stage_table<256>(tab, lds_tab);
    __syncthreads();
    const long long s = (long long)blockIdx.x * 256 + threadIdx.x;
    if (s >= n) return;
    float x[PI_D], ns[PI_D], reward = 0.0f;
    bool done = false;
#pragma unroll
    for (int d = 0; d < PI_D; ++d) ns[d] = 0.0f;
    const bool is_term = term != nullptr && term[s];
    if (!is_term) {
        pi_state_coords((unsigned int)s, lds_tab, x);
        pi_dynamics(x, lds_tab[PI_TAB_ACT + policy[s]], ns, &reward, &done);
    }
    recA[s] = AbF4{ns[0], ns[1], PI_D > 2 ? ns[2 % PI_D] : 0.0f, PI_D > 2 ? ns[3 % PI_D] : 0.0f};
#if PI_D == 6
    recB[s] = AbF4{ns[4], ns[5], reward, 0.0f};
#else
    recR[s] = reward;
#endif
    flag[s] = is_term ? 2 : done ? 1 : 0;
}

extern "C" __global__ void __launch_bounds__(PI_BLOCK_EVAL) __attribute__((amdgpu_num_sgpr(80)))
ab_replay_kernel(const float* __restrict__ V, float* __restrict__ Vn, const AbF4* __restrict__ recA,
                 const AbF4* __restrict__ recB, const float* __restrict__ recR, const unsigned char* __restrict__ flag,
                 long long s_begin, long long s_end, float gamma, int cpw) {
    long long chunk0, n_chunks;
    if (!pi_first_chunk<PI_BLOCK_EVAL>(s_end - s_begin, cpw, chunk0, n_chunks)) return;
    const int n_here = (int)(min(chunk0 + cpw, n_chunks) - chunk0);
    const unsigned int tid = threadIdx.x;
    long long sb = s_begin + chunk0 * PI_BLOCK_EVAL;
    unsigned int lane = min(tid, (unsigned int)(min(s_end - sb, (long long)PI_BLOCK_EVAL) - 1));
    struct Rec { AbF4 a; AbF4 b; float r; unsigned char f; };
    auto load = [&](long long base, unsigned int l) {
        Rec q;
        q.a = __builtin_nontemporal_load(recA + base + l);
        q.b = AbF4{0.0f, 0.0f, 0.0f, 0.0f};
        q.r = 0.0f;
#if PI_D == 6
        q.b = __builtin_nontemporal_load(recB + base + l);
#else
        q.r = __builtin_nontemporal_load(recR + base + l);
#endif
        q.f = flag != nullptr ? __builtin_nontemporal_load(flag + base + l) : (unsigned char)0;
        return q;
    };
    Rec nxt = load(sb, lane);
    for (int k = 0; k < n_here; ++k) {
        const Rec cur = nxt;
        const long long sb_c = sb;
        const unsigned int lane_c = lane;
        if (k + 1 < n_here) {
            sb += PI_BLOCK_EVAL;
            lane = min(tid, (unsigned int)(min(s_end - sb, (long long)PI_BLOCK_EVAL) - 1));
            nxt = load(sb, lane);
        }
        float ns[PI_D];
        ns[0] = cur.a.x; ns[1] = cur.a.y;
#if PI_D >= 4
        ns[2] = cur.a.z; ns[3] = cur.a.w;
#endif
#if PI_D == 6
        ns[4] = cur.b.x; ns[5] = cur.b.y;
        const float reward = cur.b.z;
#else
        const float reward = cur.r;
#endif
        float e = 0.0f;
        if (cur.f == 0) {
            unsigned int base;
            float fr[PI_D];
            pi_locate(ns, base, fr);
            e = pi_interpolate(V, base, fr);
        }
        const float nv = reward + gamma * e;
        if (tid == lane_c && cur.f != 2) pi_store_lane(Vn + sb_c, lane_c, nv);
    }
}
'''


def main():
    env = sys.argv[1] if len(sys.argv) > 1 else "double_pendulum_swingup"
    bins = int(sys.argv[2]) if len(sys.argv) > 2 else 80
    geoms = [tuple(int(v) for v in g.split("x")) for g in (sys.argv[3] if len(sys.argv) > 3 else "1024x2,512x2,256x2,256x4,1024x1").split(",")]
    cls = envs.ENVS[env]
    solver = envs.make(env, bins, device="cuda:0")
    eng = solver._backend.engine
    n, nA, D = solver.n_states, solver.n_actions, cls._D
    assert D in (4, 6)
    gamma = float(np.float32(solver.config.gamma))
    gen = torch.Generator(device="cpu").manual_seed(0)
    solver.d_value_function[:n].copy_(torch.randn(n, generator=gen, dtype=torch.float32))
    solver.d_policy[:n].copy_(torch.randint(0, nA, (n,), generator=gen, dtype=torch.int32))
    if solver._mask_arg() is not None:
        m = solver.d_terminal_mask[:n].bool()
        solver.d_value_function[:n][m] = 0.0
        solver.d_policy[:n][m] = 0
    solver.d_new_value_function.copy_(solver.d_value_function)
    eng.prepare_mask(0)                                       # state-order sweeps on both sides of the comparison
    for _ in range(2):
        solver._evaluation_sweeps(10, gamma)
        solver._improvement_sweep(gamma)
    V, pol = solver.d_value_function, solver.d_policy
    term = solver._mask_arg()
    tptr = 0 if term is None else term.data_ptr()
    user_tables = [np.asarray(b, np.float32) for b in cls.bins_space(bins).values()]
    d_tab = torch.from_numpy(np.concatenate([np.asarray(solver.action_space, np.float32)] +
                                            [user_tables[d] for d in eng.order])).cuda()
    stream = torch.cuda.current_stream().cuda_stream
    Vref = torch.zeros_like(V)
    Vb = torch.zeros_like(V)
    Vref.copy_(V)                                             # terminal states: "copy" == already there (keep_terminals)
    Vb.copy_(V)

    def timed(f, reps):
        f()
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                f()
            e1.record()
            e1.synchronize()
            best = min(best, e0.elapsed_time(e1) / reps)
        return best

    def run_a():
        eng.eval_sweep(V.data_ptr(), Vref.data_ptr(), pol.data_ptr(), tptr, 0, n, gamma, 0, stream)

    reps = 20 if n < (1 << 27) else 5
    out = {"env": env, "bins": bins, "states": n, "memory_order": list(eng.order), "A_product_ms": timed(run_a, reps),
           "product_geometry": [eng.info(11), eng.info(3)], "B": []}
    recA = torch.empty((n, 4), dtype=torch.float32, device="cuda:0")
    recB = torch.empty((n if D == 6 else 1, 4), dtype=torch.float32, device="cuda:0")
    recR = torch.empty(n if D == 4 else 1, dtype=torch.float32, device="cuda:0")
    flag = torch.empty(n, dtype=torch.uint8, device="cuda:0")
    src_text = eng.kernel_source(envs.dynamics_source(env))
    hip = ctypes.CDLL("libamdhip64.so")
    tmp = Path(tempfile.mkdtemp(prefix="ab_replay_"))
    first = True
    for blk, cpw in geoms:
        text, hits = re.subn(r"#define PI_BLOCK_EVAL \d+", f"#define PI_BLOCK_EVAL {blk}", src_text, count=1)
        assert hits == 1
        src = tmp / f"ab_{blk}.hip"
        src.write_text(text + AB_KERNELS)
        res = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-std=c++17",
                              "--genco", "-include", "hip/hip_runtime.h", "-Rpass-analysis=kernel-resource-usage",
                              str(src), "-o", str(tmp / f"ab_{blk}.hsaco")], capture_output=True, text=True)
        if res.returncode != 0:
            raise SystemExit(res.stderr[-3000:])
        vg, fn = None, None
        for line in res.stderr.splitlines():
            if "Function Name:" in line:
                fn = line.split("Function Name:")[1].split()[0]
            elif fn == "ab_replay_kernel" and " VGPRs:" in line:
                vg = int(line.split("VGPRs:")[1].split()[0])
        mod, f_prep, f_rep = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
        assert hip.hipModuleLoad(ctypes.byref(mod), str(tmp / f"ab_{blk}.hsaco").encode()) == 0
        assert hip.hipModuleGetFunction(ctypes.byref(f_prep), mod, b"ab_prepare_kernel") == 0
        assert hip.hipModuleGetFunction(ctypes.byref(f_rep), mod, b"ab_replay_kernel") == 0

        def launch(f, grid, block, vals):
            args = (ctypes.c_void_p * len(vals))(*[ctypes.cast(ctypes.byref(v), ctypes.c_void_p) for v in vals])
            rc = hip.hipModuleLaunchKernel(f, ctypes.c_uint(grid), 1, 1, ctypes.c_uint(block), 1, 1, 0, ctypes.c_void_p(stream), args, None)
            assert rc == 0, rc

        pvals = [ctypes.c_void_p(pol.data_ptr()), ctypes.c_void_p(tptr), ctypes.c_void_p(d_tab.data_ptr()), ctypes.c_longlong(n),
                 ctypes.c_void_p(recA.data_ptr()), ctypes.c_void_p(recB.data_ptr()), ctypes.c_void_p(recR.data_ptr()),
                 ctypes.c_void_p(flag.data_ptr())]
        prep = lambda: launch(f_prep, (n + 255) // 256, 256, pvals)       # noqa: E731
        if first:
            out["prepare_ms"] = timed(prep, 3)
            torch.cuda.synchronize()
            out["flags"] = {int(k): int(v) for k, v in zip(*[t.tolist() for t in torch.unique(flag, return_counts=True)])}
            first = False
        need_flag = bool(flag.any().item())
        chunks = (n + blk - 1) // blk
        groups = (chunks + cpw - 1) // cpw
        grid = max(8, 8 * ((groups + 7) // 8))
        rvals = [ctypes.c_void_p(V.data_ptr()), ctypes.c_void_p(Vb.data_ptr()), ctypes.c_void_p(recA.data_ptr()),
                 ctypes.c_void_p(recB.data_ptr()), ctypes.c_void_p(recR.data_ptr()),
                 ctypes.c_void_p(flag.data_ptr() if need_flag else 0), ctypes.c_longlong(0), ctypes.c_longlong(n),
                 ctypes.c_float(gamma), ctypes.c_int(cpw)]
        rep = lambda: launch(f_rep, grid, blk, rvals)          # noqa: E731
        Vb.copy_(V)
        run_a()
        rep()
        torch.cuda.synchronize()
        same = bool(torch.equal(Vb[:n].view(torch.int32), Vref[:n].view(torch.int32)))
        row = {"threads": blk, "cpw": cpw, "vgprs": vg, "ms": timed(rep, reps), "identical": same, "flag_stream": need_flag}
        out["B"].append(row)
        print(json.dumps(row), flush=True)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
