// pi_onelaunch_kernels.hip — the ONE-LAUNCH kernels of launch-bound grids (round 5; frozen in round 6):
//   pi_eval_resident_kernel / pi_run_resident_kernel   grids one CU's LDS holds: a whole evaluation, or the whole run()
//   pi_eval_flow_kernel (+ pi_flow_finish_kernel)      2-D / 4-D grids of up to 2^17 states: a whole evaluation, the iterates
//                                                      travelling between workgroups as tagged granules
//   pi_xcd_kernel (+ pi_xcd_finish_kernel)             2-D grids of ~4 000 to 2^16 states on the CUs of ONE XCD: an
//                                                      evaluation or the whole run()
// A device-code TEMPLATE like pi_sweep_kernels.hip and never compiled on its own: libpi_mi355.so (pi_api.cpp, build_source)
// appends it to the translation unit <defines> + pi_math.h + <step_dynamics> + pi_sweep_kernels.hip — whose helpers
// (pi_state_coords, pi_dynamics, pi_locate, pi_corner_weights, pi_stage_table, pi_wave_max, ...) it uses — ONLY for handles
// whose grid qualifies (PI_RESIDENT_K > 0, PI_FLOW or PI_XCD among the generated defines): the units of the big grids the
// BASELINE metric is quoted on do not carry these ~1 000 lines, and the hot template stays readable.
// Same arithmetic as the sweep kernels, hence the same bits; reference loops restated: policy_evaluation :300-336,
// policy_improvement :338-355, run :357-370 of src/cuda_policy_iteration.py.

// ---- LDS-resident evaluation batch for small grids ----------------------------------------
// Grids of a few thousand states (pi_create decides: up to 12 288 in 2-D, 4 096 in 4-D, 1 024 in
// 6-D) are launch-bound: one sweep is a few microseconds of launch, load -> compute -> gather
// latency and kernel boundary for very little work (profiles/r02/full_runs_mi355x.txt).  With a
// fixed policy the successor cell, its D fractional offsets and the reward of a state do not change
// from sweep to sweep, and the whole value table fits in LDS, so ONE workgroup (1024 threads in
// 2-D, 512 above) runs the entire batch of `n_sweeps` sweeps: the dynamics once per state
// (results kept in registers, PI_RESIDENT_K states per thread), then per
// sweep 2^D LDS reads and the fmaf chain per state and two workgroup barriers (Jacobi: the new
// values wait in registers until every lane has read the old ones).  The last two iterates go to
// the caller's buffers exactly where the ping-pong of pi_eval_sweeps puts them (sweep i writes Vb
// for even i, Va for odd i); the residual of the last sweep goes straight to *delta_out.
// Whole-grid batches only (the host checks): a partial range would read the other buffer's values
// outside the range on odd sweeps.  Arithmetic identical to pi_eval_sweep_kernel's.
// Second mode (sweeps_out != nullptr): the reference's whole policy_evaluation loop (:300-336) in
// this one launch — up to n_sweeps sweeps, the residual looked at on sweeps 0, check_interval,
// 2 check_interval, ... and the last one, stop as soon as it is below theta (compared in double,
// like the host's `float(delta) < theta`); every residual looked at goes to residual_log, the
// number of sweeps done to *sweeps_out, the newest iterate to Va (Vb is not touched).
#ifndef PI_RESIDENT_K
#define PI_RESIDENT_K 0
#endif
#if PI_RESIDENT_K > 0
// 1024 threads (128 VGPRs each) hold 12 two-dimensional states per thread without spilling; 4-D and
// 6-D states carry more per-state data and 2^D weights in flight: 512 threads with 256 VGPRs each.
#ifndef PI_RESIDENT_BLOCK
#define PI_RESIDENT_BLOCK (PI_D == 2 ? 1024 : 512)
#endif
#ifndef PI_RESIDENT_OVERLAP
#define PI_RESIDENT_OVERLAP (PI_D == 2 ? 4 : PI_D == 4 ? 2 : 1)
#endif
__device__ __forceinline__ float pi_interpolate_lds(const float* lv, unsigned int base, const float (&fr)[PI_D]) {
    float w[PI_C];
    pi_corner_weights(fr, w);
    float v[PI_C];
#pragma unroll
    for (int c = 0; c < PI_C; ++c) v[c] = lv[base + (unsigned int)pi_corner_offset(c)];
    float e = 0.0f;
#pragma unroll
    for (int c = 0; c < PI_C; ++c) e = fmaf(w[pi_corner_mask(c)], v[c], e);
    return e;
}
extern "C" __global__ void __launch_bounds__(PI_RESIDENT_BLOCK)
pi_eval_resident_kernel(float* __restrict__ Va, float* __restrict__ Vb, const int* __restrict__ policy,
                        const unsigned char* __restrict__ term, const float* __restrict__ tab,
                        float gamma, int n_sweeps, float* __restrict__ delta_out, double theta,
                        int check_interval, int* __restrict__ sweeps_out, float* __restrict__ residual_log) {
    __shared__ float lds_tab[PI_GRID.tab_len];
    __shared__ float lv[PI_GRID.n];                       // the value table
    __shared__ float lds_red[PI_RESIDENT_BLOCK / 64 + 1];
    const bool converge = sweeps_out != nullptr;
    constexpr unsigned int N = (unsigned int)PI_GRID.n;
    const unsigned int tid = threadIdx.x;
    for (unsigned int i = tid; i < N; i += PI_RESIDENT_BLOCK) lv[i] = Va[i];
    pi_stage_table<PI_RESIDENT_BLOCK>(tab, lds_tab);
    __syncthreads();

    // per state: 0 = no state (tail), 1 = terminal (copies its value), 2 = done successor (no
    // bootstrap), 3 = interpolates
    unsigned int kind[PI_RESIDENT_K], base[PI_RESIDENT_K];
    float fr[PI_RESIDENT_K][PI_D], reward[PI_RESIDENT_K], v_cur[PI_RESIDENT_K];
#pragma unroll
    for (int j = 0; j < PI_RESIDENT_K; ++j) {
        const unsigned int s = (unsigned int)j * PI_RESIDENT_BLOCK + tid;
        kind[j] = 0u;
        base[j] = 0u;
        reward[j] = 0.0f;
        v_cur[j] = 0.0f;
#pragma unroll
        for (int d = 0; d < PI_D; ++d) fr[j][d] = 0.0f;
        if (s < N) {
            v_cur[j] = lv[s];
            kind[j] = 1u;
            if (term == nullptr || !term[s]) {
                float x[PI_D], ns[PI_D];
                pi_state_coords(s, lds_tab, x);
                bool done;
                pi_dynamics(x, lds_tab[PI_TAB_ACT + pi_checked_action(policy[s], s)], ns, &reward[j], &done);
                kind[j] = 2u;
                if (!done) {
                    pi_locate(ns, base[j], fr[j]);
                    kind[j] = 3u;
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);                // one state at a time: the per-state results
    }                                                     // fill the register file, not the temporaries

    float dmax = 0.0f;
    int done_sweeps = 0;
    for (int i = 0; i < n_sweeps; ++i) {
        // Branch-free per state, so that the LDS reads of several states are in flight together:
        // states that do not interpolate read cell 0 (their `base`) and drop the result by select.
        float nv[PI_RESIDENT_K];
#pragma unroll
        for (int j = 0; j < PI_RESIDENT_K; ++j) {
            const float e = pi_interpolate_lds(lv, base[j], fr[j]);
            const float q = reward[j] + gamma * (kind[j] == 3u ? e : 0.0f);
            nv[j] = kind[j] >= 2u ? q : v_cur[j];
            // let PI_RESIDENT_OVERLAP states overlap, no more (registers)
            if ((j + 1) % PI_RESIDENT_OVERLAP == 0) __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();                                  // every lane has read the old table
        const bool last = i == n_sweeps - 1;
        const bool keep = !converge && i >= n_sweeps - 2; // batch mode: the two iterates the caller sees
        const bool look = last || (converge && i % check_interval == 0);
        float* dst = (i & 1) ? Va : Vb;
        if (look) dmax = 0.0f;
#pragma unroll
        for (int j = 0; j < PI_RESIDENT_K; ++j) {
            const unsigned int s = (unsigned int)j * PI_RESIDENT_BLOCK + tid;
            if (kind[j] != 0u) {
                lv[s] = nv[j];
                if (keep) dst[s] = nv[j];
            }
            const float dlt = fabsf(nv[j] - v_cur[j]);    // 0 for lanes without a state (nv = v_cur = 0)
            dmax = (look & (dlt > dmax)) ? dlt : dmax;
            v_cur[j] = nv[j];
        }
        done_sweeps = i + 1;
        if (converge && look) {                           // workgroup-wide residual, then decide together
            const float wmax = pi_wave_max(dmax);
            if ((tid & 63u) == 0u) lds_red[tid >> 6] = wmax;
            __syncthreads();
            if (tid == 0u) {
                float m = 0.0f;
#pragma unroll
                for (int wv = 0; wv < PI_RESIDENT_BLOCK / 64; ++wv) m = lds_red[wv] > m ? lds_red[wv] : m;
                lds_red[PI_RESIDENT_BLOCK / 64] = m;
                residual_log[i / check_interval + ((last && i % check_interval != 0) ? 1 : 0)] = m;
            }
            __syncthreads();                              // also: the new table is complete
            if ((double)lds_red[PI_RESIDENT_BLOCK / 64] < theta) break;
        } else {
            __syncthreads();                              // the new table is complete
        }
    }
    if (converge) {
#pragma unroll
        for (int j = 0; j < PI_RESIDENT_K; ++j) {
            const unsigned int s = (unsigned int)j * PI_RESIDENT_BLOCK + tid;
            if (kind[j] != 0u) Va[s] = v_cur[j];
        }
        if (tid == 0u) {
            *sweeps_out = done_sweeps;
            if (delta_out != nullptr) *delta_out = lds_red[PI_RESIDENT_BLOCK / 64];
        }
    } else if (delta_out != nullptr) {
        dmax = pi_wave_max(dmax);
        if ((tid & 63u) == 0u) lds_red[tid >> 6] = dmax;
        __syncthreads();
        if (tid == 0u) {
            float m = 0.0f;
#pragma unroll
            for (int wv = 0; wv < PI_RESIDENT_BLOCK / 64; ++wv) m = lds_red[wv] > m ? lds_red[wv] : m;
            *delta_out = m;
        }
    }
}
// The reference's whole run() (:357-370) for a grid one CU holds, in this ONE launch (pi_policy_iteration): V and the
// policy live in LDS, every round is the evaluation loop above (second mode) followed by the greedy step — argmax_a
// r + gamma E[V] over the LDS table, strict '>' from -1.0e30f in ascending action order (:262-280), terminal states keep
// their entry — until no entry changes or max_pi_iter rounds are done.  A thread improves the states it evaluates, so
// nobody else touches its entries of the policy.  iter_log[4 r ..] = {sweeps, residual bits, entries changed, 0};
// result[0] = rounds done, result[1] = 1 when the policy is stable.  One workgroup: nothing to wait for, nothing
// that can fail.  Arithmetic identical to pi_eval_sweep_kernel's / pi_improve_sweep_kernel's.
extern "C" __global__ void __launch_bounds__(PI_RESIDENT_BLOCK)
pi_run_resident_kernel(float* __restrict__ Va, int* __restrict__ policy, const unsigned char* __restrict__ term,
                       const float* __restrict__ tab, float gamma, int n_sweeps, double theta, int check_interval,
                       int max_pi_iter, int* __restrict__ result, unsigned int* __restrict__ iter_log) {
    __shared__ float lds_tab[PI_GRID.tab_len];
    __shared__ float lv[PI_GRID.n];                       // the value table
    __shared__ int lpol[PI_GRID.n];                       // the policy
    __shared__ float lds_red[PI_RESIDENT_BLOCK / 64 + 1];
    __shared__ unsigned int lds_cnt[PI_RESIDENT_BLOCK / 64 + 1];
    constexpr unsigned int N = (unsigned int)PI_GRID.n;
    const unsigned int tid = threadIdx.x;
    for (unsigned int i = tid; i < N; i += PI_RESIDENT_BLOCK) {
        lv[i] = Va[i];
        lpol[i] = pi_checked_action(policy[i], i);
    }
    pi_stage_table<PI_RESIDENT_BLOCK>(tab, lds_tab);
    __syncthreads();
    unsigned int role[PI_RESIDENT_K];                     // 0 = no state (tail), 1 = terminal, 2 = live
    float v_cur[PI_RESIDENT_K];
#pragma unroll
    for (int j = 0; j < PI_RESIDENT_K; ++j) {
        const unsigned int s = (unsigned int)j * PI_RESIDENT_BLOCK + tid;
        role[j] = s < N ? ((term == nullptr || !term[s]) ? 2u : 1u) : 0u;
        v_cur[j] = s < N ? lv[s] : 0.0f;
    }
    int rounds = 0, stable = 0;
    for (int it = 0; it < max_pi_iter; ++it) {
        // ---- under the current policy, once per state: 1 = keeps its value, 2 = done successor, 3 = interpolates
        unsigned int kind[PI_RESIDENT_K], base[PI_RESIDENT_K];
        float fr[PI_RESIDENT_K][PI_D], reward[PI_RESIDENT_K];
#pragma unroll
        for (int j = 0; j < PI_RESIDENT_K; ++j) {
            const unsigned int s = (unsigned int)j * PI_RESIDENT_BLOCK + tid;
            kind[j] = role[j] != 0u ? 1u : 0u;
            base[j] = 0u;
            reward[j] = 0.0f;
#pragma unroll
            for (int d = 0; d < PI_D; ++d) fr[j][d] = 0.0f;
            if (role[j] == 2u) {
                float x[PI_D], ns[PI_D];
                pi_state_coords(s, lds_tab, x);
                bool done;
                pi_dynamics(x, lds_tab[PI_TAB_ACT + lpol[s]], ns, &reward[j], &done);
                kind[j] = 2u;
                if (!done) {
                    pi_locate(ns, base[j], fr[j]);
                    kind[j] = 3u;
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // ---- policy evaluation (:300-336): the loop of pi_eval_resident_kernel's second mode
        float dmax = 0.0f;
        int sweeps = 0;
        for (int i = 0; i < n_sweeps; ++i) {
            float nv[PI_RESIDENT_K];
#pragma unroll
            for (int j = 0; j < PI_RESIDENT_K; ++j) {
                const float e = pi_interpolate_lds(lv, base[j], fr[j]);
                const float q = reward[j] + gamma * (kind[j] == 3u ? e : 0.0f);
                nv[j] = kind[j] >= 2u ? q : v_cur[j];
                if ((j + 1) % PI_RESIDENT_OVERLAP == 0) __builtin_amdgcn_sched_barrier(0);
            }
            __syncthreads();                              // every lane has read the old table
            const bool look = i == n_sweeps - 1 || i % check_interval == 0;
            if (look) dmax = 0.0f;
#pragma unroll
            for (int j = 0; j < PI_RESIDENT_K; ++j) {
                const unsigned int s = (unsigned int)j * PI_RESIDENT_BLOCK + tid;
                if (kind[j] != 0u) lv[s] = nv[j];
                const float dlt = fabsf(nv[j] - v_cur[j]);
                dmax = (look & (dlt > dmax)) ? dlt : dmax;
                v_cur[j] = nv[j];
            }
            sweeps = i + 1;
            if (look) {
                const float wmax = pi_wave_max(dmax);
                if ((tid & 63u) == 0u) lds_red[tid >> 6] = wmax;
                __syncthreads();
                if (tid == 0u) {
                    float m = 0.0f;
#pragma unroll
                    for (int wv = 0; wv < PI_RESIDENT_BLOCK / 64; ++wv) m = lds_red[wv] > m ? lds_red[wv] : m;
                    lds_red[PI_RESIDENT_BLOCK / 64] = m;
                }
                __syncthreads();                          // also: the new table is complete
                if ((double)lds_red[PI_RESIDENT_BLOCK / 64] < theta) break;
            } else {
                __syncthreads();                          // the new table is complete
            }
        }
        const float residual = lds_red[PI_RESIDENT_BLOCK / 64];
        // ---- policy improvement (:338-355) against the table the evaluation left
        unsigned int n_changed = 0u;
#pragma unroll 1                                          // one copy of the action loop: registers, not speed, matter here
        for (int j = 0; j < PI_RESIDENT_K; ++j) {
            const unsigned int s = (unsigned int)j * PI_RESIDENT_BLOCK + tid;
            if (s < N && (term == nullptr || !term[s])) {
                float x[PI_D];
                pi_state_coords(s, lds_tab, x);
                float best_q = -1.0e30f;
                int best = 0;
                for (int a = 0; a < PI_NA; ++a) {
                    float ns[PI_D], rw;
                    bool done;
                    pi_dynamics(x, lds_tab[PI_TAB_ACT + a], ns, &rw, &done);
                    float e = 0.0f;
                    if (!done) {
                        unsigned int cell;
                        float f[PI_D];
                        pi_locate(ns, cell, f);
                        e = pi_interpolate_lds(lv, cell, f);
                    }
                    const float q = rw + gamma * e;
                    if (q > best_q) { best_q = q; best = a; }
                }
                if (best != lpol[s]) {
                    lpol[s] = best;
                    ++n_changed;
                }
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) n_changed += (unsigned int)__shfl_xor((int)n_changed, o, 64);
        if ((tid & 63u) == 0u) lds_cnt[tid >> 6] = n_changed;
        __syncthreads();
        if (tid == 0u) {
            unsigned int c = 0u;
#pragma unroll
            for (int wv = 0; wv < PI_RESIDENT_BLOCK / 64; ++wv) c += lds_cnt[wv];
            lds_cnt[PI_RESIDENT_BLOCK / 64] = c;
            iter_log[4 * it + 0] = (unsigned int)sweeps;
            iter_log[4 * it + 1] = __float_as_uint(residual);
            iter_log[4 * it + 2] = c;
            iter_log[4 * it + 3] = 0u;
        }
        __syncthreads();
        rounds = it + 1;
        if (lds_cnt[PI_RESIDENT_BLOCK / 64] == 0u) {
            stable = 1;
            break;
        }
    }
#pragma unroll
    for (int j = 0; j < PI_RESIDENT_K; ++j) {
        const unsigned int s = (unsigned int)j * PI_RESIDENT_BLOCK + tid;
        if (role[j] != 0u) {
            Va[s] = v_cur[j];
            policy[s] = lpol[s];
        }
    }
    if (tid == 0u) {
        result[0] = rounds;
        result[1] = stable;
    }
}
#endif

// ---- dataflow evaluation for launch-bound grids (too big for one CU's LDS, too small to fill the chip) -------
// A grid of a few ten thousand states (BASELINE config C2: pendulum 200 x 200) sweeps in ~2 us of kernel time, and a
// policy evaluation is thousands of DEPENDENT sweeps: as launches, each sweep pays a kernel boundary (~1.45 us) plus a
// launch's fill and drain — 3.9 us per sweep in 25-node graphs (profiles/r04/bench_c2.json), whatever the kernel does.
// These kernels run the reference's whole policy_evaluation loop (:300-336) in ONE launch across many workgroups with
// NO barrier of any kind between sweeps.  As in pi_eval_resident_kernel a state's successor cell, fractional offsets
// and reward are computed once and stay in registers (the policy is fixed).  The iterates travel between waves as
// data-tagged granules: version j of V (the iterate after sweep j) lives in ring[j % 16] as one naturally aligned 8-byte
// word per state, {tag = j + 1, value bits}, written by ONE 8-byte store and read by 8-byte loads that bypass the
// reader's L1.  A wave computes sweep j for its 64 states as soon as the 2^D corner granules of each carry tag j: the
// critical path of a sweep is ONE store -> load hop, nothing else.
//   * Buffer reuse: version j overwrites version j - 16, which sweep j - 15 reads, so it may be stored only once EVERY
//     workgroup has completed sweep j - 15.  Workgroups publish "sweeps completed" in per-workgroup progress words; a
//     wave remembers the minimum it last saw and reads the words again only when that no longer covers its store —
//     every ~13 sweeps, in the same round trip as its corner granules.  The rule bounds the skew between waves to 15
//     sweeps, and the slowest wave can always proceed (its inputs cannot have been overwritten, its own store is always
//     allowed): no deadlock while every participating workgroup is resident.
//   * Completion of sweep j - 1 is reported when the poll of sweep j has come back — by then the wave's store has
//     drained, for free — through a counter in LDS: the last of a workgroup's waves to report writes the progress word.
//   * The residual is looked at on sweeps 0, check_interval, 2 check_interval, ... and the last one exactly as the host
//     loop does: on those sweeps a wave drains and reports at once, the workgroup's last reporter folds the
//     workgroup's maximum into checks[look] (atomic max of the bit pattern) BEFORE it writes the progress word, wave 0
//     of every workgroup waits for ALL progress words (the one real barrier, every 25 sweeps), reads the maximum and
//     hands it to the workgroup's other waves through LDS: all stop together.
//   * EVERY wait is bounded (timeout_ticks of the 100 MHz wall clock): a wave that gives up raises the status word
//     (behind the progress words) and leaves; pi_flow_finish_kernel turns a raised status word into *sweeps_out = -1.
// The hand-off is an agent-scope atomic store / load pair (global_store/load_dwordx2 sc1): per-location coherence of an
// 8-byte atomic object is all it relies on — no flag, no fence, no dependence on where a workgroup runs; a hop is a round
// trip through the fabric.  (An XCD-aware form — all workgroups on one XCD, verified at run time from the XCC id
// register, hand-off through that XCD's L2 — was built and is 4x SLOWER at this size: one XCD's L2 cannot serve the
// polling of 625 waves; profiles/r05/negative_results.txt (1).)
// Arithmetic identical to pi_eval_sweep_kernel's, hence the same bits, residuals and sweep counts.
#ifndef PI_FLOW
#define PI_FLOW 0
#endif
#if PI_FLOW
#ifndef PI_FLOW_BLOCK
#define PI_FLOW_BLOCK 256
#endif
#define PI_FLOW_RING 16
#ifndef PI_FLOW_SLEEP
#define PI_FLOW_SLEEP 1                                   // x 64 cycles between two polls of a wave
#endif
#define PI_FLOW_WAVES (PI_FLOW_BLOCK / 64)
#define PI_FLOW_DEAD 0xFFFFFFFFu
typedef unsigned long long PiGranule;                    // tag (high half) | float32 bits (low half)
__device__ __forceinline__ unsigned int pi_flow_load32(const unsigned int* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);        // sc1: never served by this CU's L1
}
__device__ __forceinline__ void pi_flow_store32(unsigned int* p, unsigned int v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// One wave reports that it has completed sweep k (its granules of version k have left: the caller has waited for its
// vector-memory counter).  The last wave of the workgroup to do so publishes the workgroup's progress — after folding
// the workgroup's residual maximum into checks[slot] when sweep k is one the residual is looked at.
__device__ __forceinline__ void pi_flow_report(int k, bool look, int slot, float wave_max, unsigned int wg,
                                               unsigned int* lds_count, unsigned int* lds_max,
                                               unsigned int* __restrict__ progress, unsigned int* __restrict__ checks) {
    if ((threadIdx.x & 63u) != 0u) return;
    if (look && wave_max > 0.0f) (void)atomicMax(lds_max + (slot & 1), __float_as_uint(wave_max));
    const unsigned int before = atomicAdd(lds_count + (k & 31), 1u);
    if (before != PI_FLOW_WAVES - 1u) return;
    lds_count[k & 31] = 0u;                                // next used 32 sweeps on; the skew is at most 15
    if (look) {
        const unsigned int m = atomicExch(lds_max + (slot & 1), 0u);
        if (m != 0u) {
            const unsigned int was = atomicMax(checks + slot, m);          // returning: complete before the word below
            asm volatile("" ::"v"(was));
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    pi_flow_store32(progress + wg, (unsigned int)(k + 1));
}
// All W progress words and the status word behind them, one load per 64 words: the smallest progress, or PI_FLOW_DEAD
// when the status word is raised.  Wave-uniform.
__device__ __forceinline__ unsigned int pi_flow_min_progress(const unsigned int* __restrict__ progress, unsigned int W) {
    const unsigned int lane = threadIdx.x & 63u;
    unsigned int m = 0x7FFFFFFFu;
    bool failed = false;
    for (unsigned int i = lane; i <= W; i += 64u) {
        const unsigned int p = pi_flow_load32(progress + i);
        if (i < W) m = min(m, p);
        else failed = p != 0u;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = min(m, (unsigned int)__shfl_xor((int)m, o, 64));
    return __any(failed) ? PI_FLOW_DEAD : m;
}
extern "C" __global__ void __launch_bounds__(PI_FLOW_BLOCK)
pi_eval_flow_kernel(float* __restrict__ Va, const int* __restrict__ policy, const unsigned char* __restrict__ term,
                    const float* __restrict__ tab, float gamma, int n_sweeps, float* __restrict__ delta_out, double theta,
                    int check_interval, int* __restrict__ sweeps_out, float* __restrict__ residual_log,
                    PiGranule* __restrict__ ring, unsigned int* __restrict__ progress, unsigned int* __restrict__ checks,
                    unsigned long long timeout_ticks) {
    __shared__ float lds_tab[PI_GRID.tab_len];
    __shared__ unsigned int lds_count[32], lds_max[2], lds_seq, lds_bits;
    constexpr unsigned int N = (unsigned int)PI_GRID.n;
    const unsigned int W = gridDim.x, wg = blockIdx.x;
    const unsigned int tid = threadIdx.x, lane = tid & 63u;
    if (tid < 32u) lds_count[tid] = 0u;
    if (tid < 2u) lds_max[tid] = 0u;
    if (tid == 0u) {
        lds_seq = 0u;
        lds_bits = 0u;
    }
    pi_stage_table<PI_FLOW_BLOCK>(tab, lds_tab);
    __syncthreads();                                       // the only workgroup barrier of the kernel
    const unsigned int s = wg * PI_FLOW_BLOCK + tid;

    // per state, once: 0 = no state (tail), 1 = terminal (keeps its value), 2 = done successor (no bootstrap), 3 = interpolates
    unsigned int kind = 0u, base = 0u;
    float fr[PI_D], reward = 0.0f, v_cur = 0.0f;
#pragma unroll
    for (int d = 0; d < PI_D; ++d) fr[d] = 0.0f;
    if (s < N) {
        v_cur = Va[s];
        kind = 1u;
        if (term == nullptr || !term[s]) {
            float x[PI_D], ns[PI_D];
            pi_state_coords(s, lds_tab, x);
            bool done;
            pi_dynamics(x, lds_tab[PI_TAB_ACT + pi_checked_action(policy[s], s)], ns, &reward, &done);
            kind = 2u;
            if (!done) {
                pi_locate(ns, base, fr);
                kind = 3u;
            }
        }
    }

    bool dead = false;                                     // wave-uniform
    int done_sweeps = 0, reported = 0;                     // sweeps this wave has computed / reported as complete
    unsigned int known = 0u;                               // every workgroup has completed at least this many sweeps
    unsigned int looks = 0u;                               // residual looks so far (sequence number of the LDS hand-over)
    float residual = 0.0f;
    for (int j = 0; j < n_sweeps && !dead; ++j) {
        float w[PI_C], v[PI_C];
        pi_corner_weights(fr, w);
#pragma unroll
        for (int c = 0; c < PI_C; ++c) v[c] = 0.0f;
        if (j == 0) {                                      // the caller's V: written before this launch, plain loads
            if (kind == 3u) {
#pragma unroll
                for (int c = 0; c < PI_C; ++c) v[c] = Va[base + (unsigned int)pi_corner_offset(c)];
            }
        } else {
            // one poll = one round trip: the corner granules of version j - 1 and, when the remembered minimum no longer
            // covers this sweep's store, every progress word with the status word
            const PiGranule* src = ring + (size_t)((j - 1) % PI_FLOW_RING) * N;
            const unsigned int want = (unsigned int)j;    // tag of version j - 1
            const unsigned int need = j + 2 > PI_FLOW_RING ? (unsigned int)(j + 2 - PI_FLOW_RING) : 0u;
            unsigned long long t0 = 0ull;
            unsigned int spins = 0u;
            unsigned int missing = kind == 3u ? (unsigned int)((1ull << PI_C) - 1ull) : 0u;   // corners not yet seen at `want`
#if PI_C > 32
#error "the corner mask of the dataflow kernel holds 32 corners (2-D and 4-D grids)"
#endif
            while (true) {
                // a corner that has arrived stays valid until this wave has stored version j (flow control): only the
                // missing ones are asked for again
                PiGranule g[PI_C];
#pragma unroll
                for (int c = 0; c < PI_C; ++c)
                    if (missing & (1u << c))
                        g[c] = __hip_atomic_load(src + base + (unsigned int)pi_corner_offset(c), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (known < need || (spins & 255u) == 255u) {          // also: look at the status word now and then
                    const unsigned int m = pi_flow_min_progress(progress, W);
                    if (m == PI_FLOW_DEAD) { dead = true; break; }
                    known = m;
                }
#pragma unroll
                for (int c = 0; c < PI_C; ++c)
                    if ((missing & (1u << c)) && (unsigned int)(g[c] >> 32) == want) {
                        v[c] = __uint_as_float((unsigned int)g[c]);
                        missing &= ~(1u << c);
                    }
                if (__all(missing == 0u) && known >= need) break;
                if (spins == 0u) t0 = wall_clock64();
                else if ((spins & 15u) == 0u && wall_clock64() - t0 > timeout_ticks) { dead = true; break; }
                ++spins;
                __builtin_amdgcn_s_sleep(PI_FLOW_SLEEP);
            }
            if (dead) break;
            if (reported < j) {                            // sweep j - 1: its store went out before this poll came back
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                pi_flow_report(j - 1, false, 0, 0.0f, wg, lds_count, lds_max, progress, checks);
                reported = j;
            }
        }
        float e = 0.0f;
#pragma unroll
        for (int c = 0; c < PI_C; ++c) e = fmaf(w[pi_corner_mask(c)], v[c], e);
        const float q = reward + gamma * (kind == 3u ? e : 0.0f);
        const float nv = kind >= 2u ? q : v_cur;
        const bool last = j == n_sweeps - 1;
        const bool look = last || j % check_interval == 0;
        const float dlt = fabsf(nv - v_cur);               // 0 for lanes without a state
        v_cur = nv;
        if (kind != 0u)
            __hip_atomic_store(ring + (size_t)(j % PI_FLOW_RING) * N + s,
                               ((PiGranule)(unsigned int)(j + 1) << 32) | (PiGranule)__float_as_uint(nv), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
        done_sweeps = j + 1;
        if (look) {
            // drain and report at once, then the one real barrier: every workgroup's maximum is in when its word says j + 1
            const int slot = j / check_interval + ((last && j % check_interval != 0) ? 1 : 0);
            const float wave_max = pi_wave_max(dlt);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            pi_flow_report(j, true, slot, wave_max, wg, lds_count, lds_max, progress, checks);
            reported = j + 1;
            ++looks;
            unsigned long long t0 = 0ull;
            unsigned int spins = 0u, bits = 0u;
            if (tid < 64u) {                               // wave 0 waits for everyone and hands the verdict over in LDS
                while (true) {
                    const unsigned int m = pi_flow_min_progress(progress, W);
                    if (m == PI_FLOW_DEAD) { dead = true; break; }
                    known = m;
                    if (m >= (unsigned int)(j + 1)) break;
                    if (spins == 0u) t0 = wall_clock64();
                    else if ((spins & 15u) == 0u && wall_clock64() - t0 > timeout_ticks) { dead = true; break; }
                    ++spins;
                    __builtin_amdgcn_s_sleep(1);
                }
                if (!dead) bits = pi_flow_load32(checks + slot);      // issued after every word was seen at j + 1
                if (lane == 0u) {
                    *reinterpret_cast<volatile unsigned int*>(&lds_bits) = bits;
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // the bits are in LDS before the number
                    *reinterpret_cast<volatile unsigned int*>(&lds_seq) = dead ? PI_FLOW_DEAD : looks;
                }
            } else {
                while (true) {
                    const unsigned int seq = *reinterpret_cast<volatile unsigned int*>(&lds_seq);
                    if (seq == PI_FLOW_DEAD) { dead = true; break; }
                    if (seq == looks) break;
                    if (spins == 0u) t0 = wall_clock64();
                    else if ((spins & 15u) == 0u && wall_clock64() - t0 > timeout_ticks + timeout_ticks) { dead = true; break; }
                    ++spins;
                    __builtin_amdgcn_s_sleep(1);
                }
                bits = *reinterpret_cast<volatile unsigned int*>(&lds_bits);
                known = max(known, (unsigned int)(j + 1));
            }
            if (dead) break;
            residual = __uint_as_float(bits);
            if (wg == 0u && tid == 0u) residual_log[slot] = residual;
            if ((double)residual < theta) break;
        }
    }
    if (dead) {
        if (lane == 0u) {
            pi_flow_store32(progress + W, 1u + (unsigned int)done_sweeps);
            *reinterpret_cast<volatile unsigned int*>(&lds_seq) = PI_FLOW_DEAD;
        }
        return;
    }
    // V itself is NOT written here (round 6): a wave elsewhere may still give up, and then the caller's V has to be what
    // it was — pi_flow_finish_kernel copies the last version out of the ring once the status word is known to be clear
    if (wg == 0u && tid == 0u) {
        *sweeps_out = done_sweeps;
        if (delta_out != nullptr) *delta_out = residual;
    }
}
// ---- XCD-local evaluation: a whole policy evaluation on the CUs of ONE XCD, hand-off through that XCD's L2 -------------
// The dataflow kernel above pays one trip through the fabric per sweep (2.8 us on MI355X) because its workgroups may
// run anywhere.  A grid of a few ten thousand states does not need the whole chip: this kernel runs the evaluation on
// the 32 CUs of ONE XCD — one workgroup of 1 024 threads per CU, up to 2 states per thread: 2^16 states —, whose L2
// all of them share:
//   * successor cell, fractional offsets and reward of a thread's states stay in registers (the policy is fixed);
//   * the iterates travel as data-tagged 8-byte granules {tag = sweep + 1, value bits} like the dataflow kernel's, but
//     written with PLAIN stores (a plain store stops in the XCD's L2) and read with 16-byte loads that bypass the
//     reader's L1 (sc1; the L2 answers): a thread asks for its 2^D corner granules of version j - 1, and asks again for
//     the states whose corners do not all carry tag j yet.  No barrier, no flag, no drain between two sweeps: the
//     critical path of a sweep is one store -> L2 -> load, nothing else;
//   * versions live in a ring of 64 (memory is plentiful: 64 x 8 n bytes <= 32 MB); every 32nd sweep — and every sweep
//     the host loop looks at the residual on, i.e. every 25th — ends with a real barrier (below), so a version is
//     overwritten only when every workgroup is at least 32 sweeps past the sweep that read it;
//   * the barrier: every wave drains its stores (s_waitcnt vmcnt(0)), the workgroup meets, thread 0 stores the
//     workgroup's flag granule {sweep + 1, bits of the workgroup's residual maximum} (plain), and wave 0 polls the flag
//     granules of all W <= 64 workgroups — one wave-wide 8-byte sc1 load, four lines — until every one carries this
//     sweep's number.  Two banks of flags (parity of the barrier count).  The residual of a look is the maximum over
//     the flags: all workgroups stop together.
//     What does NOT work here, measured (profiles/r05/negative_results.txt (1)): polling with a non-temporal load — it is
//     served by the CU's L1 once the line is there (37 of 40 workgroups spun on a stale count); a counter of atomic
//     adds polled with returning atomics — correct, but gfx950 performs device-scope atomics on the memory side of the
//     fabric: 3.1 us per sweep; a barrier of this kind after EVERY sweep with untagged values — 2.3 us per sweep, because
//     a value another CU has just stored takes ~1 400 cycles to read even through the shared L2, and the barrier's flag
//     pays that once more.
// HIP promises nothing about where a workgroup runs, so nothing is assumed: the host launches spare workgroups; each reads
// its XCC id from the hardware register; the first W that find themselves on XCD 0 take a ticket and take part, the
// others leave at once.  Too few on XCD 0 or a wait that runs out (timeout_ticks) raise the status word: every
// wave leaves at its next poll, and pi_xcd_finish_kernel — the ONLY writer of V and *sweeps_out — reports
// *sweeps_out < 0 with V untouched; the host then runs the evaluation again with the placement-independent dataflow kernel.
// Arithmetic identical to pi_eval_sweep_kernel's: same bits, residuals, sweep counts.
#ifndef PI_XCD
#define PI_XCD 0
#endif
#if PI_XCD
#define PI_XCD_BLOCK 1024
#ifndef PI_XCD_TIMING
#define PI_XCD_TIMING 0
#endif
#ifndef PI_XCD_RING
#define PI_XCD_RING 64
#endif
#define PI_XCD_SYNC (PI_XCD_RING / 2)                     // a barrier after every 32nd sweep at the latest: RING >= SYNC + 1
#ifndef PI_XCD_FIRST_SLEEP
#define PI_XCD_FIRST_SLEEP 8                              // x 64 cycles between a wave's store and its first look at the next version
#endif
#define PI_XCD_CTL_STATUS 64                              // control words (on lines of their own): 0 tickets
#define PI_XCD_CTL_DONE 80                                // sweeps done, residual bits, iterations, stable (workgroup 0, at the end)
#define PI_XCD_CTL_FLAGS 128                              // 2 banks x 64 flag granules of 8 bytes (256 words)
#define PI_XCD_CTL_WORDS (PI_XCD_CTL_FLAGS + 256)
#ifdef PI_XCD_HOST_CTL_WORDS
static_assert(PI_XCD_CTL_WORDS == PI_XCD_HOST_CTL_WORDS, "the host allocates another control block than this kernel lays out");
#endif
#define PI_XCD_NPAD (((unsigned int)PI_GRID.n + 15u) & ~15u)      // granules per version: whole 128-byte lines
typedef unsigned int PiQuad __attribute__((ext_vector_type(4)));   // two adjacent granules: {bits, tag, bits, tag}
// One workgroup per CU (PI_XCD_PAD floats of LDS that nothing else needs see to it), PI_XCD_S states per workgroup —
// the host's choice: n over the XCD's 32 CUs, rounded up to whole 128-byte lines — i.e. PI_XCD_K = ceil(S / 1024) states
// per thread: every CU carries the same load.
#ifndef PI_XCD_S
#define PI_XCD_S 1024
#endif
#define PI_XCD_K ((PI_XCD_S + PI_XCD_BLOCK - 1) / PI_XCD_BLOCK)
#define PI_XCD_W (((unsigned int)PI_GRID.n + PI_XCD_S - 1u) / PI_XCD_S)
#define PI_XCD_PAD 21504                                  // 84 KB: more than half a CU's LDS
// What every wait of the kernel needs to be bounded.
struct PiXcdWait {
    unsigned int* status;
    unsigned long long timeout_ticks;
};
// The corner values of K cells from version `src` of V, whose granules carry tag `want`: 16-byte loads that bypass the L1
// (two adjacent granules each), asked again for the states some of whose corners are older.  need[k]: state k asks at
// all.  Returns false when the wave gave up (status word raised by somebody, or its own time limit: it raises the word).
template <int K>
__device__ __forceinline__ bool pi_xcd_gather(const PiGranule* src, unsigned int want, bool (&need)[K],
                                              const unsigned int (&base)[K], PiPair (&vp)[K][PI_NPAIR], const PiXcdWait& w,
                                              unsigned int& polls) {
    constexpr int kNear = 1 << (PI_D - 2);
    unsigned long long t0 = 0ull;
    unsigned int spins = 0u;
    while (true) {
        PiQuad q[K][PI_NPAIR];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            if (need[k]) {
#pragma unroll
                for (int m = 0; m < kNear; ++m) {
                    int far = 0;
#pragma unroll
                    for (int d = 0; d < PI_D - 2; ++d) far += ((m >> d) & 1) * PI_GRID.stride[d];
                    const PiGranule* p = src + (base[k] + (unsigned int)far);
                    const PiGranule* p2 = p + PI_GRID.stride[PI_D - 2];
                    asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(q[k][m]) : "v"(p) : "memory");
                    asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(q[k][m | kNear]) : "v"(p2) : "memory");
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        bool any = false;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            if (need[k]) {
                bool fresh = true;
#pragma unroll
                for (int m = 0; m < PI_NPAIR; ++m) {
                    asm volatile("" : "+v"(q[k][m]));                  // nothing reads a quad above the wait
                    fresh = fresh && q[k][m].y == want && q[k][m].w == want;
                }
                if (fresh) {
#pragma unroll
                    for (int m = 0; m < PI_NPAIR; ++m) {
                        vp[k][m].x = __uint_as_float(q[k][m].x);
                        vp[k][m].y = __uint_as_float(q[k][m].z);
                    }
                    need[k] = false;
                }
            }
            any = any || need[k];
        }
        ++polls;
        if (!__any(any)) return true;
        if ((spins & 31u) == 31u && pi_flow_load32(w.status) != 0u) return false;
        if (spins == 0u) t0 = wall_clock64();
        else if (wall_clock64() - t0 > w.timeout_ticks) {
            if ((threadIdx.x & 63u) == 0u) (void)atomicMax(w.status, 1u);
            return false;
        }
        ++spins;
        __builtin_amdgcn_s_sleep(1);
    }
}
// The barrier between the workgroups of the XCD, carrying one word per workgroup: every wave has drained its stores and
// left its word (a residual maximum as float bits, or a count) in lds_part[wave]; wave 0 folds them (maximum of the bit
// patterns, or sum), publishes the workgroup's flag granule {number of this barrier, word} with a plain store and polls
// the flags of all W workgroups until every one carries the number.  Returns false when a wait ran out; `out`: the
// maximum / sum over all workgroups, the same in every thread.
template <bool SUM>
__device__ __forceinline__ bool pi_xcd_barrier(unsigned int& barriers, unsigned int wg, unsigned int* lds_part, unsigned int* lds_word,
                                               unsigned int* lds_ok, PiGranule* flags, const PiXcdWait& w, unsigned int& out) {
    constexpr unsigned int W = PI_XCD_W;
    const unsigned int tid = threadIdx.x, lane = tid & 63u;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's values are in L2 before the workgroup says so
    __syncthreads();
    if (tid < 64u) {
        PiGranule* bank = flags + (size_t)(barriers & 1u) * 64u;
        const unsigned int seq = barriers + 1u;
        if (lane == 0u) {
            unsigned int m = 0u;
#pragma unroll
            for (int wv = 0; wv < PI_XCD_BLOCK / 64; ++wv) m = SUM ? m + lds_part[wv] : (lds_part[wv] > m ? lds_part[wv] : m);
            bank[wg] = ((PiGranule)seq << 32) | (PiGranule)m;                // plain: stays in this XCD's L2
        }
        asm volatile("" ::: "memory");
        unsigned long long t0 = 0ull;
        unsigned int spins = 0u, word = 0u;
        bool ok = true;
        while (true) {
            PiGranule g = (PiGranule)seq << 32;
            if (lane < W) g = __hip_atomic_load(bank + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // sc1: never this CU's L1
            word = (unsigned int)g;
            if (__all((unsigned int)(g >> 32) == seq)) break;
            if ((spins & 31u) == 31u && pi_flow_load32(w.status) != 0u) { ok = false; break; }
            if (spins == 0u) t0 = wall_clock64();
            else if (wall_clock64() - t0 > w.timeout_ticks) { ok = false; break; }
            ++spins;
            __builtin_amdgcn_s_sleep(1);
        }
        if (ok) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const unsigned int t = (unsigned int)__shfl_xor((int)word, o, 64);
                word = SUM ? word + t : (t > word ? t : word);
            }
        }
        if (lane == 0u) {
            if (!ok) (void)atomicMax(w.status, 1u);
            *lds_word = word;
            *lds_ok = ok ? 1u : 0u;
        }
    }
    ++barriers;
    __syncthreads();
    out = *lds_word;
    return *lds_ok != 0u;
}
// max_pi_iter == 0: ONE policy evaluation under the policy at `policy` (pi_policy_evaluation).  max_pi_iter >= 1: the
// reference's whole run() (:357-370) — evaluate, improve, until no entry of the policy changes or max_pi_iter rounds are
// done — in this one launch: the greedy step reads the evaluation's last version (tagged, behind the evaluation's last
// barrier), a thread's actions stay in registers, the number of changed entries travels in the barrier's flags, and
// iter_log[4 it ..] = {sweeps, residual bits, entries changed, 0} for every round.  Results go to the ring / pol_out;
// pi_xcd_finish_kernel copies them into V / policy after a clean run.
extern "C" __global__ void __launch_bounds__(PI_XCD_BLOCK)
pi_xcd_kernel(const float* __restrict__ Va, const int* __restrict__ policy, const unsigned char* __restrict__ term,
              const float* __restrict__ tab, float gamma, int n_sweeps, double theta, int check_interval, int max_pi_iter,
              float* __restrict__ residual_log, unsigned int* __restrict__ iter_log, PiGranule* ring, int* __restrict__ pol_out,
              unsigned int* ctl, unsigned long long timeout_ticks) {
    constexpr unsigned int N = (unsigned int)PI_GRID.n, W = PI_XCD_W;
    static_assert(W <= 64u, "one wave polls the flags of all workgroups");
    static_assert(PI_XCD_RING >= PI_XCD_SYNC + 1, "a version must outlive the sweeps that may still read it");
    __shared__ float lds_tab[PI_GRID.tab_len];
    __shared__ unsigned int lds_part[PI_XCD_BLOCK / 64];
    __shared__ unsigned int lds_wg, lds_ok, lds_word;
    __shared__ float lds_pad[PI_XCD_PAD];
    const unsigned int tid = threadIdx.x, lane = tid & 63u;
    if (n_sweeps < 0) {                                    // never: keeps the allocation
        for (unsigned int i = tid; i < PI_XCD_PAD; i += PI_XCD_BLOCK) lds_pad[i] = gamma;
        __syncthreads();
        residual_log[tid] = lds_pad[(tid * 21u + 1u) % PI_XCD_PAD];
    }
    if (tid == 0u) {
        unsigned int xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        unsigned int wg = PI_FLOW_DEAD;
        if ((xcc & 15u) == 0u) {
            const unsigned int t = atomicAdd(ctl, 1u);
            if (t < W) wg = t;
        }
        lds_wg = wg;
        lds_ok = 1u;
        lds_word = 0u;
    }
    pi_stage_table<PI_XCD_BLOCK>(tab, lds_tab);
    __syncthreads();
    const unsigned int wg = lds_wg;
    if (wg == PI_FLOW_DEAD) return;                        // a spare workgroup, or one on another XCD
    const unsigned int s0 = wg * (unsigned int)PI_XCD_S + tid;
    const unsigned int s_end = min(N, (wg + 1u) * (unsigned int)PI_XCD_S);
    const PiXcdWait wait{ctl + PI_XCD_CTL_STATUS, timeout_ticks};
    PiGranule* flags = reinterpret_cast<PiGranule*>(ctl + PI_XCD_CTL_FLAGS);

    // a thread's states: 0 = no state (tail), 1 = terminal (keeps its value and its entry of the policy), 2 = live
    unsigned int role[PI_XCD_K];
    int act[PI_XCD_K];
    float v_cur[PI_XCD_K];
#pragma unroll
    for (int k = 0; k < PI_XCD_K; ++k) {
        const unsigned int s = s0 + (unsigned int)k * PI_XCD_BLOCK;
        role[k] = 0u;
        act[k] = 0;
        v_cur[k] = 0.0f;
        if (s < s_end) {
            v_cur[k] = Va[s];
            act[k] = pi_checked_action(policy[s], s);
            role[k] = (term == nullptr || !term[s]) ? 2u : 1u;
        }
    }
    unsigned int g = 0u, barriers = 0u, polls = 0u;       // g: sweeps done since the launch = number of the next version
    unsigned int rounds = 0u, stable = 0u;
    float residual = 0.0f;
    bool dead = false;                                     // wave-uniform inside a gather, workgroup-uniform behind a barrier
#if PI_XCD_TIMING
    unsigned long long tacc[5] = {0ull, 0ull, 0ull, 0ull, 0ull}, tp = __builtin_readcyclecounter();
#define PI_XCD_STAMP(k) do { const unsigned long long tn = __builtin_readcyclecounter(); tacc[k] += tn - tp; tp = tn; } while (0)
#else
#define PI_XCD_STAMP(k)
#endif
    const int n_rounds = max_pi_iter > 0 ? max_pi_iter : 1;
    for (int it = 0; it < n_rounds && !dead; ++it) {
        // ---- under the current policy, once per state: 0 = no state, 1 = keeps its value, 2 = done successor (no
        // bootstrap), 3 = interpolates; successor cell, fractional offsets, reward
        unsigned int kind[PI_XCD_K], base[PI_XCD_K];
        float fr[PI_XCD_K][PI_D], reward[PI_XCD_K];
#pragma unroll
        for (int k = 0; k < PI_XCD_K; ++k) {
            const unsigned int s = s0 + (unsigned int)k * PI_XCD_BLOCK;
            kind[k] = role[k] != 0u ? 1u : 0u;
            base[k] = 0u;
            reward[k] = 0.0f;
#pragma unroll
            for (int d = 0; d < PI_D; ++d) fr[k][d] = 0.0f;
            if (role[k] == 2u) {
                float x[PI_D], ns[PI_D];
                pi_state_coords(s, lds_tab, x);
                bool done;
                pi_dynamics(x, lds_tab[PI_TAB_ACT + act[k]], ns, &reward[k], &done);
                kind[k] = 2u;
                if (!done) {
                    pi_locate(ns, base[k], fr[k]);
                    kind[k] = 3u;
                }
            }
            __builtin_amdgcn_sched_barrier(0);            // one state at a time
        }
        // ---- policy evaluation (:300-336)
        int sweeps = 0;
        for (int j = 0; j < n_sweeps; ++j) {
            const bool last = j == n_sweeps - 1;
            const bool look = last || j % check_interval == 0;
            const int slot = j / check_interval + ((last && j % check_interval != 0) ? 1 : 0);
            const bool sync = look || (g + 1u) % PI_XCD_SYNC == 0u;
            PiPair vp[PI_XCD_K][PI_NPAIR];
            if (g == 0u) {
                // the caller's V: plain values, nobody writes them
#pragma unroll
                for (int k = 0; k < PI_XCD_K; ++k) pi_request_corners(Va, base[k], vp[k]);
            } else {
                bool need[PI_XCD_K];
#pragma unroll
                for (int k = 0; k < PI_XCD_K; ++k) need[k] = kind[k] == 3u;
                __builtin_amdgcn_s_sleep(PI_XCD_FIRST_SLEEP);
                if (!pi_xcd_gather<PI_XCD_K>(ring + (size_t)((g - 1u) % PI_XCD_RING) * PI_XCD_NPAD, g, need, base, vp, wait, polls)) {
                    dead = true;
                    break;
                }
            }
            PI_XCD_STAMP(0);
            PiGranule* dst = ring + (size_t)(g % PI_XCD_RING) * PI_XCD_NPAD;
            float dmax = 0.0f;
#pragma unroll
            for (int k = 0; k < PI_XCD_K; ++k) {
                // states that do not interpolate never asked for anything: their pairs are not looked at
                float e = 0.0f;
                if (kind[k] == 3u) e = pi_combine_corners(vp[k], fr[k]);
                const float q = reward[k] + gamma * e;
                const float nv = kind[k] >= 2u ? q : v_cur[k];
                const float dlt = fabsf(nv - v_cur[k]);    // 0 for lanes without a state
                dmax = dlt > dmax ? dlt : dmax;
                v_cur[k] = nv;
                if (kind[k] != 0u)                         // plain: stays in this XCD's L2
                    dst[s0 + (unsigned int)k * PI_XCD_BLOCK] = ((PiGranule)(g + 1u) << 32) | (PiGranule)__float_as_uint(nv);
            }
            ++g;
            sweeps = j + 1;
            PI_XCD_STAMP(1);
            if (!sync) continue;
            if (look) {
                const float wmax = pi_wave_max(dmax);
                if (lane == 0u) lds_part[tid >> 6] = __float_as_uint(wmax);     // non-negative floats order like their bit patterns
            } else if (lane == 0u) {
                lds_part[tid >> 6] = 0u;
            }
            unsigned int bits;
            if (!pi_xcd_barrier<false>(barriers, wg, lds_part, &lds_word, &lds_ok, flags, wait, bits)) {
                dead = true;
                break;
            }
            PI_XCD_STAMP(2);
            if (look) {
                residual = __uint_as_float(bits);
                if (max_pi_iter == 0 && wg == 0u && tid == 0u) residual_log[slot] = residual;
                if ((double)residual < theta) break;
            }
        }
        if (dead || max_pi_iter == 0) break;
        // ---- policy improvement (:338-355) from version g - 1: argmax_a r + gamma E[V], strict '>' from -1.0e30f in
        // ascending action order (:262-280); terminal states keep their entry
        unsigned int n_changed = 0u;
        const PiGranule* vfin = ring + (size_t)((g - 1u) % PI_XCD_RING) * PI_XCD_NPAD;
#pragma unroll
        for (int k = 0; k < PI_XCD_K; ++k) {
            const unsigned int s = s0 + (unsigned int)k * PI_XCD_BLOCK;
            bool gave_up = false;
            if (role[k] == 2u) {
                float x[PI_D];
                pi_state_coords(s, lds_tab, x);
                float best_q = -1.0e30f;
                int best = 0;
                for (int a = 0; a < PI_NA; ++a) {
                    float ns[PI_D], rw;
                    bool done;
                    pi_dynamics(x, lds_tab[PI_TAB_ACT + a], ns, &rw, &done);
                    float e = 0.0f;
                    bool ask[1] = {!done};
                    unsigned int cell[1] = {0u};
                    float f[PI_D];
#pragma unroll
                    for (int d = 0; d < PI_D; ++d) f[d] = 0.0f;
                    if (!done) pi_locate(ns, cell[0], f);
                    PiPair v1[1][PI_NPAIR];
                    if (__any(!done)) {
                        if (!pi_xcd_gather<1>(vfin, g, ask, cell, v1, wait, polls)) { gave_up = true; break; }
                    }
                    if (!done) e = pi_combine_corners(v1[0], f);
                    const float q = rw + gamma * e;
                    if (q > best_q) { best_q = q; best = a; }
                }
                if (!gave_up && best != act[k]) {
                    act[k] = best;
                    ++n_changed;
                }
            }
            if (__any(gave_up)) dead = true;
        }
        // (a wave that gave up still goes through the barrier below: it ends on the status word for everybody)
        {
            unsigned int wsum = n_changed;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) wsum += (unsigned int)__shfl_xor((int)wsum, o, 64);
            if (lane == 0u) lds_part[tid >> 6] = wsum;
        }
        unsigned int changed;
        if (!pi_xcd_barrier<true>(barriers, wg, lds_part, &lds_word, &lds_ok, flags, wait, changed) || dead) {
            dead = true;
            break;
        }
        rounds = (unsigned int)(it + 1);
        if (wg == 0u && tid == 0u) {
            iter_log[4 * it + 0] = (unsigned int)sweeps;
            iter_log[4 * it + 1] = __float_as_uint(residual);
            iter_log[4 * it + 2] = changed;
            iter_log[4 * it + 3] = 0u;
        }
        if (changed == 0u) {
            stable = 1u;
            break;
        }
    }
    if (dead) return;
    if (max_pi_iter > 0) {
#pragma unroll
        for (int k = 0; k < PI_XCD_K; ++k)
            if (role[k] != 0u) pol_out[s0 + (unsigned int)k * PI_XCD_BLOCK] = act[k];
    }
#if PI_XCD_TIMING
    if ((wg == 0u || wg == W - 1u) && tid == 0u) {
        for (int k = 0; k < 3; ++k) ctl[8 + (wg == 0u ? 0 : 8) + k] = (unsigned int)(tacc[k] / (unsigned long long)g);
        ctl[8 + (wg == 0u ? 0 : 8) + 3] = (unsigned int)((unsigned long long)polls * 1000ull / (unsigned long long)g);
    }
#endif
    if (wg == 0u && tid == 0u) {
        ctl[PI_XCD_CTL_DONE] = g;
        ctl[PI_XCD_CTL_DONE + 1] = __float_as_uint(residual);
        ctl[PI_XCD_CTL_DONE + 2] = rounds;
        ctl[PI_XCD_CTL_DONE + 3] = stable;
    }
}
// Launched right behind pi_xcd_kernel, ceil(n / 256) workgroups: the ONLY writer of V, the policy, *sweeps_out and
// *delta_out.  The status word decides (a wait ran out), and so does the ticket count (fewer than W workgroups ever
// found themselves on XCD 0): *sweeps_out = -1 and V and the policy keep what they held before the launch, whatever single
// workgroups went through; otherwise the last iterate (and, after a whole run, the policy) is copied out and *sweeps_out
// = the sweeps done in all (one evaluation), or the rounds done (a whole run; sweeps_out[1] = 1 when the policy is stable).
extern "C" __global__ void __launch_bounds__(256)
pi_xcd_finish_kernel(float* __restrict__ Va, int* __restrict__ policy, const PiGranule* __restrict__ ring,
                     const int* __restrict__ pol_out, const unsigned int* __restrict__ ctl, int whole_run,
                     int* __restrict__ sweeps_out, float* __restrict__ delta_out) {
    constexpr unsigned int N = (unsigned int)PI_GRID.n, W = PI_XCD_W;
    unsigned int st = ctl[PI_XCD_CTL_STATUS];
    const unsigned int done = ctl[PI_XCD_CTL_DONE];
    if (st == 0u && (ctl[0] < W || done == 0u)) st = 1u;
    const bool first = blockIdx.x == 0u && threadIdx.x == 0u;
    if (st != 0u) {
        if (first) *sweeps_out = -(int)st;
        return;
    }
    const unsigned int s = blockIdx.x * 256u + threadIdx.x;
    if (s < N) {
        Va[s] = __uint_as_float((unsigned int)ring[(size_t)((done - 1u) % PI_XCD_RING) * PI_XCD_NPAD + s]);
        if (whole_run) policy[s] = pol_out[s];
    }
    if (first) {
        if (whole_run) {
            sweeps_out[0] = (int)ctl[PI_XCD_CTL_DONE + 2];
            sweeps_out[1] = (int)ctl[PI_XCD_CTL_DONE + 3];
        } else {
            *sweeps_out = (int)done;
        }
        if (delta_out != nullptr) *delta_out = __uint_as_float(ctl[PI_XCD_CTL_DONE + 1]);
    }
}
#endif

// Launched right behind a dataflow kernel, one thread per state: a wave may have given up while the others went through
// their last barrier, so the status word, not workgroup 0, has the final say on whether the evaluation is valid — and this
// kernel is the ONLY writer of the caller's V: status clear -> V[s] = the value of the last version in the ring (every
// workgroup stopped on the same sweep, *sweeps_out); status raised -> *sweeps_out = -1 and V is what it was before the
// launch, so the caller can run the same evaluation sweep by sweep (as pi_xcd_finish_kernel does for the XCD-local kernel).
extern "C" __global__ void __launch_bounds__(256)
pi_flow_finish_kernel(const unsigned int* __restrict__ progress, unsigned int W, int* __restrict__ sweeps_out,
                      const PiGranule* __restrict__ ring, float* __restrict__ Va) {
    constexpr unsigned int N = (unsigned int)PI_GRID.n;
    if (pi_flow_load32(progress + W) != 0u) {
        if (blockIdx.x == 0u && threadIdx.x == 0u) *sweeps_out = -1;
        return;
    }
    const int done = *sweeps_out;                          // written by the kernel in front; nobody writes it on this path
    const unsigned int s = blockIdx.x * 256u + threadIdx.x;
    if (s < N && done >= 1) Va[s] = __uint_as_float((unsigned int)ring[(size_t)((done - 1) % PI_FLOW_RING) * N + s]);
}
#endif
