// pi_push_kernels.hip — the swept-first launch of a sharded evaluation sweep that DELIVERS its rows itself
// (peer-to-peer transport, csrc/pi_p2p.cpp).  Appended to the sweep-kernel translation unit
// (generated defines + pi_math.h + the env plugin + pi_sweep_kernels.hip) and built as a SECOND module of the handle,
// lazily, the first time a plan asks for it — the primary module, its cache key and its code are untouched.
//
// pi_eval_push_kernel is pi_eval_live_kernel (same arithmetic per state, hence the same bits) over the row-exact
// swept-first list of a shard, plus: every listed state carries a mask of the peers that read its row
// (dest[k], bit j = peers[j]), and the lane that stores V'(s) into this rank's buffer stores it into the SAME
// offset of those peers' buffers as well (plain stores to HIP IPC mappings: they travel over xGMI while the rest of
// the launch computes).  No flags in here: the kernel boundary is the release — the host enqueues a one-wave kernel
// that raises the peers' data counters right behind this launch (pi_p2p_sigwait_kernel) — so the hot loop carries no
// fence.  Replaces, per sweep, the copy kernel of the unfused exchange, its stream hop and one pass over the rows.
// No reference counterpart (src/cuda_policy_iteration.py:616-649 is the single-device sweep it restates).

extern "C" __global__ void __launch_bounds__(PI_BLOCK_EVAL) __attribute__((amdgpu_num_sgpr(80)))
pi_eval_push_kernel(const float* __restrict__ V, float* __restrict__ Vn, const int* __restrict__ policy,
                    const int* __restrict__ live, const unsigned char* __restrict__ dest,
                    float* const* __restrict__ peers, int n_peers, const float* __restrict__ tab, long long n_live,
                    float gamma, unsigned int* __restrict__ delta_bits, int cpw) {
    __shared__ float lds_tab[PI_GRID.tab_len];
    long long chunk0, n_chunks;
    if (!pi_first_chunk<PI_BLOCK_EVAL>(n_live, cpw, chunk0, n_chunks)) return;
    const int n_here = (int)(min(chunk0 + cpw, n_chunks) - chunk0);
    const unsigned int tid = threadIdx.x;
    const long long kb0 = chunk0 * PI_BLOCK_EVAL;                       // first list entry of the workgroup
    const bool need_old = delta_bits != nullptr;                        // launch-uniform
    auto lane_of = [&](int k) {
        return min(tid, (unsigned int)(min(n_live - (kb0 + (long long)k * PI_BLOCK_EVAL), (long long)PI_BLOCK_EVAL) - 1));
    };
    auto entry = [&](int k, unsigned int lane) {
        return (unsigned int)__builtin_nontemporal_load(pi_lane_ptr(live + kb0 + (long long)k * PI_BLOCK_EVAL, lane));
    };
    auto mask_of = [&](int k, unsigned int lane) {
        return (unsigned int)__builtin_nontemporal_load(pi_lane_ptr(dest + kb0 + (long long)k * PI_BLOCK_EVAL, lane));
    };
    unsigned int lane_cur = lane_of(0);
    unsigned int s_cur = entry(0, lane_cur);
    unsigned int m_cur = mask_of(0, lane_cur);
    unsigned int lane_nxt = lane_cur, s_nxt = s_cur, m_nxt = m_cur;
    if (n_here > 1) {
        lane_nxt = lane_of(1);
        s_nxt = entry(1, lane_nxt);
        m_nxt = mask_of(1, lane_nxt);
    }
    int a_cur = __builtin_nontemporal_load(policy + s_cur);
    float v_cur = 0.0f;
    if (need_old) v_cur = V[s_cur];
    pi_stage_table<PI_BLOCK_EVAL>(tab, lds_tab);
    __syncthreads();

    float dmax = 0.0f;
    for (int k = 0; k < n_here; ++k) {
        const unsigned int s = s_cur, lane_c = lane_cur, m = m_cur;
        const int action = a_cur;
        const float v_old = v_cur;
        if (k + 1 < n_here) {                                           // inputs of the next chunk; index of the one after
            s_cur = s_nxt;
            lane_cur = lane_nxt;
            m_cur = m_nxt;
            a_cur = __builtin_nontemporal_load(policy + s_cur);
            if (need_old) v_cur = V[s_cur];
            if (k + 2 < n_here) {
                lane_nxt = lane_of(k + 2);
                s_nxt = entry(k + 2, lane_nxt);
                m_nxt = mask_of(k + 2, lane_nxt);
            }
        }
        float x[PI_D], ns[PI_D], reward;
        pi_state_coords(s, lds_tab, x);
        const float a = lds_tab[PI_TAB_ACT + pi_checked_action(action, s)];
        bool done;
        pi_dynamics(x, a, ns, &reward, &done);
        float e = 0.0f;
        if (!done) {
            unsigned int base;
            float fr[PI_D];
            pi_locate(ns, base, fr);
            e = pi_interpolate(V, base, fr);
        }
        const float nv = reward + gamma * e;
        if (tid == lane_c) {
            Vn[s] = nv;
            for (int j = 0; j < n_peers; ++j)                           // wave-uniform loop, stores under the lanes' masks
                if ((m >> j) & 1u) peers[j][s] = nv;
            const float dlt = fabsf(nv - v_old);
            dmax = dlt > dmax ? dlt : dmax;
        }
    }
    if (delta_bits != nullptr) pi_wave_max_to<PI_BLOCK_EVAL>(dmax, delta_bits);
}

// ---- reach of a state range at (dimension 0, coupled velocity) granularity, in ANY memory order ---------------------
// The halo of a shard along dimension 0 is a triangle in (i_0, i_v) — v the velocity that moves coordinate 0
// (x' = x + dt x_dot: the USER's dimension 1 in every reference env) — because only states that are fast enough reach
// the neighbouring planes.  pi_reach_units_kernel measures it in rows (i_0, i_1) of the MEMORY order, which are the
// same sets only while the velocity is the second-slowest dimension — the env's own order.  With the velocity anywhere
// else (the fast single-GPU orders put it along the lanes: DESIGN.md section 3) those rows are all reachable and the
// halo doubles.  The fused exchange does not need contiguous rows — every state carries its own destination mask —
// so this probe measures the pairs (i_0, i_v) themselves, v = PI_MEM_OF[1]: bit i_0 * g_v + i_v of `bitmap` is set
// when some state of [s_begin, s_end) under some action reads a corner with those two indices.
constexpr int PI_PAIR_DIM = PI_D >= 2 ? PI_MEM_OF[1] : 0;
constexpr int PI_PAIR_UNITS = PI_GRID.g[0] * (PI_D >= 2 ? PI_GRID.g[PI_PAIR_DIM] : 1);
#define PI_PAIR_WORDS ((PI_PAIR_UNITS + 31) / 32)
#if PI_D >= 3
extern "C" __global__ void __launch_bounds__(PI_BLOCK)
pi_reach_pairs_kernel(const unsigned char* __restrict__ term, const float* __restrict__ tab, long long s_begin,
                      long long s_end, unsigned int* __restrict__ bitmap, int cpw) {
    // the host launches this only when g_0 * g_v <= 2^17 (16 KB of bits); bigger grids keep the row-level plan
    constexpr bool kFits = PI_PAIR_UNITS <= (1 << 17);
    __shared__ float lds_tab[PI_GRID.tab_len];
    __shared__ unsigned int lds_bits[kFits ? PI_PAIR_WORDS : 1];
    if (!kFits) return;
    long long chunk0, n_chunks;
    if (!pi_first_chunk<PI_BLOCK>(s_end - s_begin, cpw, chunk0, n_chunks)) return;
    const long long chunk_end = min(chunk0 + cpw, n_chunks);
    pi_stage_table<PI_BLOCK>(tab, lds_tab);
    for (int i = threadIdx.x; i < PI_PAIR_WORDS; i += PI_BLOCK) lds_bits[i] = 0u;
    __syncthreads();
    constexpr unsigned int gv = PI_GRID.g[PI_PAIR_DIM];
    constexpr unsigned int st0 = PI_GRID.stride[0], stv = PI_GRID.stride[PI_PAIR_DIM];
    for (long long chunk = chunk0; chunk < chunk_end; ++chunk) {
        const long long s = s_begin + chunk * PI_BLOCK + threadIdx.x;
        if (s >= s_end || (term != nullptr && term[s])) continue;
        float x[PI_D];
        pi_state_coords((unsigned int)s, lds_tab, x);
        int last = -1;
        for (int a = 0; a < PI_NA; ++a) {
            float ns[PI_D], reward, fr[PI_D];
            bool done;
            pi_dynamics(x, lds_tab[PI_TAB_ACT + a], ns, &reward, &done);
            if (done) continue;
            unsigned int base;
            pi_locate(ns, base, fr);
            const int u = (int)((base / st0) * gv + (base / stv) % gv);
            if (u == last) continue;
            last = u;
            atomicOr(&lds_bits[u >> 5], 1u << (u & 31));                      // the cell's four corners in (i_0, i_v)
            atomicOr(&lds_bits[(u + 1) >> 5], 1u << ((u + 1) & 31));
            atomicOr(&lds_bits[(u + (int)gv) >> 5], 1u << ((u + (int)gv) & 31));
            atomicOr(&lds_bits[(u + (int)gv + 1) >> 5], 1u << ((u + (int)gv + 1) & 31));
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < PI_PAIR_WORDS; i += PI_BLOCK)
        if (lds_bits[i] != 0u) atomicOr(&bitmap[i], lds_bits[i]);
}
#endif
