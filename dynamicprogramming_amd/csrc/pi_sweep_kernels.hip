// pi_sweep_kernels.hip — Bellman-backup sweep kernels for gfx950 (MI355X, CDNA4).
//
// This is a device-code TEMPLATE, never compiled on its own.  libpi_mi355.so
// (pi_api.cpp) builds one translation unit per (grid shape, action count, env):
//
//     <generated #defines: PI_D, PI_NA, PI_GRID_INIT, PI_SPT, PI_SCHED, PI_TILE_INIT, ...>
//     <include/pi_math.h>  + #define sinf/cosf/fmodf -> pi_*   (deterministic math)
//     <the user's step_dynamics C string>                       (env plugin)
//     <this file>  +  pi_tile_kernels.hip (opt-in tile-staged family, reach probe)
//
// and compiles it with hipRTC (--offload-arch=gfx950 -O3 -ffp-contract=off).
// __graft_entry__.build() runs the same assembly through `hipcc --genco` for the
// built-in envs so the code objects are checked and cached ahead of time.
//
// Semantics restated from the reference (src/cuda_policy_iteration.py, NVRTC strings):
//   interpolation      get_barycentric_2d :183-210 / _4d :580-614 / _6d :1007-1042
//   evaluation sweep   policy_eval_kernel :212-242 / _4d :616-649 / _6d :1044-1079
//   improvement sweep  policy_improve_kernel :244-283 / _4d :651-691 / _6d :1081-1123
//   max|V'-V|          cp.ReductionKernel :164-172   (fused here: no second pass)
//   policy-stable test old.copy() / all(==) :340,:354 (fused here: changed counter)
// One thread owns one state, as in the reference; what is different is everything
// around it: state coordinates come from per-dimension bin tables in LDS instead of
// an (n, D) float array in HBM (16-24 B/state of traffic removed), grid shape and
// strides are compile-time constants, the 2^D corner weights share their partial
// products, the residual and the changed-count are reduced with wave shuffles and
// one atomic per workgroup, and workgroups walk the state range in an XCD-aware
// order so that each XCD's private L2 sees one contiguous slab of V.
//
// Arithmetic contract (bit-exact against oracle/pi_oracle.cpp): fp32 throughout,
// no contraction, IEEE division, the fmaf chain over corners in ascending corner
// order from 0.0f, `reward + gamma * E` as mul then add, strict `>` argmax from
// -1.0e30f (lowest index wins ties, NaN never wins).

#define PI_C (1 << PI_D)
#define PI_BLOCK 256
// Optional occupancy targets (waves per SIMD) for the two hot kernels; 0 = let the compiler
// decide.  Second argument of __launch_bounds__ on AMD = minimum waves per SIMD (EU).
#ifndef PI_EVAL_MIN_WAVES
#define PI_EVAL_MIN_WAVES 0
#endif
#ifndef PI_IMPROVE_MIN_WAVES
#define PI_IMPROVE_MIN_WAVES 0
#endif
#if PI_EVAL_MIN_WAVES > 0
#define PI_LB_EVAL __launch_bounds__(PI_BLOCK, PI_EVAL_MIN_WAVES)
#else
#define PI_LB_EVAL __launch_bounds__(PI_BLOCK)
#endif
#if PI_IMPROVE_MIN_WAVES > 0
#define PI_LB_IMPROVE __launch_bounds__(PI_BLOCK, PI_IMPROVE_MIN_WAVES)
#else
#define PI_LB_IMPROVE __launch_bounds__(PI_BLOCK)
#endif
#define PI_NXCD 8

// ---- compile-time grid geometry ------------------------------------------------
struct PiGrid {
    int g[PI_D];
    int stride[PI_D];
    int bins_off[PI_D];   // offset of dimension d's bin table inside the float table
    int tab_len;
};
__host__ __device__ constexpr PiGrid pi_make_grid() {
    PiGrid r = {};
    const int g[PI_D] = PI_GRID_INIT;
    for (int d = 0; d < PI_D; ++d) r.g[d] = g[d];
    r.stride[PI_D - 1] = 1;
    for (int d = PI_D - 2; d >= 0; --d) r.stride[d] = r.stride[d + 1] * r.g[d + 1];
    int off = 2 * PI_D + PI_NA;
    for (int d = 0; d < PI_D; ++d) { r.bins_off[d] = off; off += r.g[d]; }
    r.tab_len = off;
    return r;
}
constexpr PiGrid PI_GRID = pi_make_grid();
// Float table layout (device buffer `tab`, built by pi_create):
//   [0, D) bounds_low | [D, 2D) bounds_high | [2D, 2D+NA) actions | bins_0 | bins_1 | ...
#define PI_TAB_LO 0
#define PI_TAB_HI PI_D
#define PI_TAB_ACT (2 * PI_D)

// Which dimension-bit of the partial-product index a corner number selects.
// 4D/6D: bit d of corner c <-> dimension d (:607, :1035).  2D is written out with
// dimension 1 toggling fastest (:201-209), i.e. the two bits are swapped.
__device__ __forceinline__ constexpr int pi_corner_mask(int c) {
#if PI_D == 2
    return ((c & 1) << 1) | ((c >> 1) & 1);
#else
    return c;
#endif
}
__device__ __forceinline__ constexpr int pi_corner_offset(int c) {
    int m = pi_corner_mask(c), off = 0;
    for (int d = 0; d < PI_D; ++d) off += ((m >> d) & 1) * PI_GRID.stride[d];
    return off;
}

// Call the plugin with the arity the reference documents for each D (:11-15, :456-460, :869-874).
__device__ __forceinline__ void pi_dynamics(const float (&s)[PI_D], float a, float (&ns)[PI_D],
                                            float* reward, bool* done) {
#if PI_D == 2
    step_dynamics(s[0], s[1], a, &ns[0], &ns[1], reward, done);
#elif PI_D == 4
    step_dynamics(s[0], s[1], s[2], s[3], a, &ns[0], &ns[1], &ns[2], &ns[3], reward, done);
#elif PI_D == 6
    step_dynamics(s[0], s[1], s[2], s[3], s[4], s[5], a,
                  &ns[0], &ns[1], &ns[2], &ns[3], &ns[4], &ns[5], reward, done);
#else
#error "PI_D must be 2, 4 or 6"
#endif
}

// Cell of a continuous point: flat index of its lowest corner and the D fractional offsets
// (get_barycentric_*: normalise, clamp to the border, truncate, `frac = n - i`).
__device__ __forceinline__ void pi_locate(const float (&ns)[PI_D], const float* __restrict__ tab,
                                          int& base, float (&fr)[PI_D]) {
    base = 0;
#pragma unroll
    for (int d = 0; d < PI_D; ++d) {
        const float lo = tab[PI_TAB_LO + d];       // wave-uniform: scalar loads
        const float hi = tab[PI_TAB_HI + d];
        const float top = (float)(PI_GRID.g[d] - 1);
        float n = (ns[d] - lo) / (hi - lo) * top;
        n = fmaxf(0.0f, fminf(n, top));            // clamp-to-border; NaN lands on `top`
        int i = min((int)n, PI_GRID.g[d] - 2);
        fr[d] = n - (float)i;
        base += i * PI_GRID.stride[d];
    }
}

// The 2^D corner weights from the fractional offsets.  The reference multiplies
// 1.0f * a_0 * a_1 * ... left to right for every corner; sharing the common prefixes is the
// same sequence of roundings.  w[] is indexed by the partial-product mask (bit d <-> dim d).
__device__ __forceinline__ void pi_corner_weights(const float (&fr)[PI_D], float (&w)[PI_C]) {
    w[0] = 1.0f - fr[0];
    w[1] = fr[0];
#pragma unroll
    for (int k = 1; k < PI_D; ++k) {
        const float om = 1.0f - fr[k];
#pragma unroll
        for (int m = (1 << k) - 1; m >= 0; --m) {
            w[m + (1 << k)] = w[m] * fr[k];
            w[m] = w[m] * om;
        }
    }
}

// Multilinear interpolation of V over the cell: all 2^D loads are issued first (corner pairs
// along the last dimension are adjacent in memory and fuse into 8-byte loads), then the
// fmaf chain runs in ascending corner order from 0.0f like the reference's.
__device__ __forceinline__ float pi_interpolate(const float* __restrict__ V, int base,
                                                const float (&fr)[PI_D]) {
    float w[PI_C];
    pi_corner_weights(fr, w);
    const float* __restrict__ Vb = V + base;
    float v[PI_C];
#pragma unroll
    for (int c = 0; c < PI_C; ++c) v[c] = Vb[pi_corner_offset(c)];
    float e = 0.0f;
#pragma unroll
    for (int c = 0; c < PI_C; ++c) e = fmaf(w[pi_corner_mask(c)], v[c], e);
    return e;
}

__device__ __forceinline__ float pi_backup(const float (&s)[PI_D], float a,
                                           const float* __restrict__ V,
                                           const float* __restrict__ tab, float gamma) {
    float ns[PI_D], reward;
    bool done;
    pi_dynamics(s, a, ns, &reward, &done);
    float e = 0.0f;
    if (!done) {
        int base;
        float fr[PI_D];
        pi_locate(ns, tab, base, fr);
        e = pi_interpolate(V, base, fr);
    }
    return reward + gamma * e;
}

// Flat state index -> coordinates, through the LDS copy of the bin tables.
__device__ __forceinline__ void pi_state_coords(unsigned int s, const float* lds_tab,
                                                float (&x)[PI_D]) {
    unsigned int r = s;
#pragma unroll
    for (int d = PI_D - 1; d > 0; --d) {
        unsigned int q = r / (unsigned int)PI_GRID.g[d];
        x[d] = lds_tab[PI_GRID.bins_off[d] + (int)(r - q * (unsigned int)PI_GRID.g[d])];
        r = q;
    }
    x[0] = lds_tab[PI_GRID.bins_off[0] + (int)r];
}

// Workgroup -> 256-state chunk schedule.  Workgroups are dealt round-robin over the
// 8 XCDs (blockIdx % 8 shares an L2), so XCD x walks chunks [x*span, (x+1)*span) in
// order: every L2 sees one contiguous slab of V.  Placement only affects speed.
// The host launches one workgroup per chunk by default (the loops below then run once): the
// dispatcher starts workgroups in index order, so the chunks in flight on an XCD form one
// compact advancing window — measured 5x less traffic past L2 than a grid-stride launch with
// 8 resident workgroups per CU.  Fewer workgroups still work (grid-stride).
struct PiChunks {
    long long n_chunks, span, j, step, x;
};
#ifndef PI_SCHED
#define PI_SCHED 0      // 0: one contiguous slab per XCD (default); 1: plain grid-stride (tuning)
#endif
__device__ __forceinline__ PiChunks pi_chunks_of(long long n_chunks) {
    PiChunks c;
    c.n_chunks = n_chunks;
#if PI_SCHED == 0
    c.span = (c.n_chunks + PI_NXCD - 1) / PI_NXCD;
    c.x = blockIdx.x % PI_NXCD;
    c.j = blockIdx.x / PI_NXCD;
    c.step = gridDim.x / PI_NXCD;      // host launches a multiple of 8 workgroups
#else
    c.span = c.n_chunks;
    c.x = 0;
    c.j = blockIdx.x;
    c.step = gridDim.x;
#endif
    return c;
}
__device__ __forceinline__ PiChunks pi_chunks(long long count) {
    return pi_chunks_of((count + PI_BLOCK - 1) / PI_BLOCK);
}

__device__ __forceinline__ float pi_wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        float t = __shfl_xor(v, o, 64);
        v = t > v ? t : v;
    }
    return v;
}

// ---- transition records ------------------------------------------------------------
// Under a FIXED policy the transition of every state (reward, cell, fractional offsets) is
// the same in every evaluation sweep; only V changes.  The first sweep of a policy
// evaluation can therefore record it (PI_BUILD) and the remaining sweeps — thousands of them
// at gamma = 0.999 — replay the records instead of re-running the dynamics: the same fp32
// operations on the same operands, hence bit-identical V, at a few loads per state instead of
// ~450 VALU instructions.  MI355X's 288 GB make this a non-issue in size: (2 + D) * 4 B per
// state, 0.98 GB for the 80^4 grid, 7.8 GB for 25^6.
// Layout (struct of arrays over k = s - s_base, s_base = s_begin rounded down to 4;
// `cap` entries per array, a multiple of 4):  reward[cap] | base[cap] | frac_0[cap] | ...
//   base >= 0 : flat index of the cell's lowest corner
//   base = -1 : the transition terminates (E = 0, V' = reward + gamma * 0)
//   base = -2 : terminal grid node (V' = V)
#ifndef PI_STREAM_NT
#define PI_STREAM_NT 1
#endif
#if PI_STREAM_NT
#define PI_STREAM_LOAD(p) __builtin_nontemporal_load(p)
#else
#define PI_STREAM_LOAD(p) (*(p))
#endif
#define PI_REC_DONE (-1)
#define PI_REC_TERMINAL (-2)
#ifndef PI_SPT
#define PI_SPT 2              // states per thread in the replay kernel (1, 2 or 4)
#endif

// Residual / changed-count are accumulated into PI_NSLOT slots (one word saturates at ~90
// atomics per microsecond on MI355X: with one counter an exact-grid improvement sweep of 80^4
// spent 4 ms of its 7 ms in 640 k same-address atomics) and folded by pi_finalize_kernel.
#define PI_NSLOT 256
__device__ __forceinline__ void pi_block_max_to(float dmax, float* lds_red,
                                                unsigned int* __restrict__ delta_bits) {
    dmax = pi_wave_max(dmax);
    if ((threadIdx.x & 63) == 0) lds_red[threadIdx.x >> 6] = dmax;
    __syncthreads();
    if (threadIdx.x == 0) {
        float m = lds_red[0];
#pragma unroll
        for (int w = 1; w < PI_BLOCK / 64; ++w) m = lds_red[w] > m ? lds_red[w] : m;
        if (m > 0.0f) atomicMax(delta_bits + (blockIdx.x & (PI_NSLOT - 1)), __float_as_uint(m));
    }
}

// ---- policy evaluation sweep ---------------------------------------------------
// Vn[s] = r(s, pi(s)) + gamma * E[V](s')   for s in [s_begin, s_end); terminal: copy.
// delta_bits (nullable): atomic max of the bit pattern of max|Vn - V| (>= 0, so the
// unsigned order is the float order); the host zeroes it before the launch.
// BUILD: additionally write the transition records (rec, cap) described above.
template <bool BUILD>
__device__ __forceinline__ void pi_eval_body(const float* __restrict__ V, float* __restrict__ Vn,
                                             const int* __restrict__ policy,
                                             const unsigned char* __restrict__ term,
                                             const float* __restrict__ tab, long long s_begin,
                                             long long s_end, float gamma,
                                             unsigned int* __restrict__ delta_bits,
                                             float* __restrict__ rec, long long cap) {
    __shared__ float lds_tab[PI_GRID.tab_len];
    __shared__ float lds_red[PI_BLOCK / 64];
    for (int i = threadIdx.x; i < PI_GRID.tab_len; i += PI_BLOCK) lds_tab[i] = tab[i];
    __syncthreads();

    const long long s_base = s_begin & ~3LL;
    const PiChunks ck = pi_chunks(s_end - s_begin);
    float dmax = 0.0f;
    for (long long cl = ck.j; cl < ck.span; cl += ck.step) {
        const long long chunk = ck.x * ck.span + cl;
        if (chunk >= ck.n_chunks) break;
        const long long s = s_begin + chunk * PI_BLOCK + threadIdx.x;
        if (s >= s_end) continue;
        const float v_old = V[s];
        float nv = v_old;
        int base = PI_REC_TERMINAL;
        float reward = 0.0f, fr[PI_D];
#pragma unroll
        for (int d = 0; d < PI_D; ++d) fr[d] = 0.0f;
        // policy and mask are read exactly once per sweep: stream them (nt) so they do not
        // displace V lines, which every neighbouring state re-reads, from L2 / Infinity Cache.
        if (!PI_STREAM_LOAD(&term[s])) {
            float x[PI_D], ns[PI_D];
            pi_state_coords((unsigned int)s, lds_tab, x);
            const float a = lds_tab[PI_TAB_ACT + PI_STREAM_LOAD(&policy[s])];
            bool done;
            pi_dynamics(x, a, ns, &reward, &done);
            float e = 0.0f;
            base = PI_REC_DONE;
            if (!done) {
                pi_locate(ns, tab, base, fr);
                e = pi_interpolate(V, base, fr);
            }
            nv = reward + gamma * e;
        }
        Vn[s] = nv;
        if (BUILD) {
            const long long k = s - s_base;
            rec[k] = reward;
            reinterpret_cast<int*>(rec)[cap + k] = base;
#pragma unroll
            for (int d = 0; d < PI_D; ++d) rec[(2 + d) * cap + k] = fr[d];
        }
        const float dlt = fabsf(nv - v_old);
        dmax = dlt > dmax ? dlt : dmax;
    }
    if (delta_bits != nullptr) pi_block_max_to(dmax, lds_red, delta_bits);
}

extern "C" __global__ void PI_LB_EVAL
pi_eval_sweep_kernel(const float* __restrict__ V, float* __restrict__ Vn,
                     const int* __restrict__ policy, const unsigned char* __restrict__ term,
                     const float* __restrict__ tab, long long s_begin, long long s_end,
                     float gamma, unsigned int* __restrict__ delta_bits) {
    pi_eval_body<false>(V, Vn, policy, term, tab, s_begin, s_end, gamma, delta_bits, nullptr, 0);
}

extern "C" __global__ void __launch_bounds__(PI_BLOCK)
pi_eval_build_kernel(const float* __restrict__ V, float* __restrict__ Vn,
                     const int* __restrict__ policy, const unsigned char* __restrict__ term,
                     const float* __restrict__ tab, long long s_begin, long long s_end,
                     float gamma, unsigned int* __restrict__ delta_bits,
                     float* __restrict__ rec, long long cap) {
    pi_eval_body<true>(V, Vn, policy, term, tab, s_begin, s_end, gamma, delta_bits, rec, cap);
}

// ---- evaluation sweep from the transition records (the steady-state hot loop) --------
// HBM-bound: per state (2 + D) * 4 B of records in, 4 B of V' out, the 2^D corner reads of V
// served by L2 / Infinity Cache.  Each thread owns PI_SPT consecutive states so the record
// loads and the V' store are 8- or 16-byte accesses; chunks are aligned to s_base so those
// vectors are naturally aligned whatever the shard boundaries are.
template <int N> struct PiVec {
    typedef float f __attribute__((ext_vector_type(N)));
    typedef int i __attribute__((ext_vector_type(N)));
};
template <> struct PiVec<1> { typedef float f; typedef int i; };

#ifndef PI_REPLAY_NT
#define PI_REPLAY_NT 1        // records are read once per sweep: stream them past L2 / MALL
#endif
#ifndef PI_REPLAY_PREFETCH
#define PI_REPLAY_PREFETCH 1  // fetch the next chunk's records while gathering this chunk's V
#endif

template <typename T>
__device__ __forceinline__ T pi_stream_load(const T* p) {
#if PI_REPLAY_NT
    return __builtin_nontemporal_load(p);
#else
    return *p;
#endif
}

struct PiRecords {
    typename PiVec<PI_SPT>::f r, f[PI_D];
    typename PiVec<PI_SPT>::i b;
};
__device__ __forceinline__ void pi_load_records(PiRecords& q, const float* __restrict__ rec,
                                                long long cap, long long k0) {
    typedef PiVec<PI_SPT>::f vf;
    typedef PiVec<PI_SPT>::i vi;
    q.r = pi_stream_load(reinterpret_cast<const vf*>(rec + k0));
    q.b = pi_stream_load(reinterpret_cast<const vi*>(reinterpret_cast<const int*>(rec) + cap + k0));
#pragma unroll
    for (int d = 0; d < PI_D; ++d)
        q.f[d] = pi_stream_load(reinterpret_cast<const vf*>(rec + (2 + d) * cap + k0));
}
#if PI_SPT == 1
#define PI_LANE(v, j) (v)
#else
#define PI_LANE(v, j) ((v)[j])
#endif

extern "C" __global__ void __launch_bounds__(PI_BLOCK)
pi_eval_replay_kernel(const float* __restrict__ V, float* __restrict__ Vn,
                      const float* __restrict__ rec, long long cap, long long s_begin,
                      long long s_end, float gamma, unsigned int* __restrict__ delta_bits) {
    typedef PiVec<PI_SPT>::f vf;
    __shared__ float lds_red[PI_BLOCK / 64];
    const long long s_base = s_begin & ~3LL;
    const long long count = s_end - s_base;
    const PiChunks ck = pi_chunks_of((count + PI_BLOCK * PI_SPT - 1) / (PI_BLOCK * PI_SPT));
    float dmax = 0.0f;

    // The chunk list of this workgroup is known up front, so the loop is software-pipelined:
    // the records of chunk t+1 are requested before chunk t's dependent V gathers are issued.
    long long cl = ck.j;
    long long chunk = ck.x * ck.span + cl;
    bool live = cl < ck.span && chunk < ck.n_chunks;
    PiRecords cur, nxt;
    if (live) pi_load_records(cur, rec, cap, (chunk * PI_BLOCK + threadIdx.x) * PI_SPT);
    while (live) {
        const long long k0 = (chunk * PI_BLOCK + threadIdx.x) * PI_SPT;
        const long long s0 = s_base + k0;
        const long long cl_n = cl + ck.step;
        const long long chunk_n = ck.x * ck.span + cl_n;
        const bool live_n = cl_n < ck.span && chunk_n < ck.n_chunks;
#if PI_REPLAY_PREFETCH
        if (live_n) pi_load_records(nxt, rec, cap, (chunk_n * PI_BLOCK + threadIdx.x) * PI_SPT);
#endif
        // Straight-line body: lanes outside [s_begin, s_end) and records without a cell
        // (terminal node / terminating transition) gather from cell 0 and discard the result,
        // so every load of the chunk is in flight before the first fmaf.
        const bool full = (s0 >= s_begin) && (s0 + PI_SPT <= s_end);
        vf out;
        float old[PI_SPT];
        bool ok[PI_SPT];
#pragma unroll
        for (int j = 0; j < PI_SPT; ++j) {
            ok[j] = full || (s0 + j >= s_begin && s0 + j < s_end);
            old[j] = V[ok[j] ? s0 + j : s_begin];
        }
#pragma unroll
        for (int j = 0; j < PI_SPT; ++j) {
            const int b = PI_LANE(cur.b, j);
            float fr[PI_D];
#pragma unroll
            for (int d = 0; d < PI_D; ++d) fr[d] = PI_LANE(cur.f[d], j);
            const float e = pi_interpolate(V, (ok[j] && b >= 0) ? b : 0, fr);
            float nv = PI_LANE(cur.r, j) + gamma * (b >= 0 ? e : 0.0f);
            nv = (b == PI_REC_TERMINAL) ? old[j] : nv;
            const float dlt = ok[j] ? fabsf(nv - old[j]) : 0.0f;
            dmax = dlt > dmax ? dlt : dmax;
            PI_LANE(out, j) = nv;
        }
        if (full) *reinterpret_cast<vf*>(Vn + s0) = out;
        else {
#pragma unroll
            for (int j = 0; j < PI_SPT; ++j) if (ok[j]) Vn[s0 + j] = PI_LANE(out, j);
        }
#if PI_REPLAY_PREFETCH
        cur = nxt;
#else
        if (live_n) pi_load_records(cur, rec, cap, (chunk_n * PI_BLOCK + threadIdx.x) * PI_SPT);
#endif
        cl = cl_n;
        chunk = chunk_n;
        live = live_n;
    }
    if (delta_bits != nullptr) pi_block_max_to(dmax, lds_red, delta_bits);
}

// ---- greedy policy improvement sweep -------------------------------------------
// policy[s] = argmax_a [ r(s,a) + gamma * E[V](s'_a) ], first maximum wins; terminal
// states keep their entry.  changed (nullable): number of entries that changed.
// WRITE_V (value-iteration sweep, the fused form the reference's README sketches at :790-799
// but does not implement): also Vn[s] = max_a Q(s,a) (terminal: copy) and the residual.
template <bool WRITE_V>
__device__ __forceinline__ void pi_improve_body(const float* __restrict__ V, float* __restrict__ Vn,
                                                int* __restrict__ policy,
                                                const unsigned char* __restrict__ term,
                                                const float* __restrict__ tab, long long s_begin,
                                                long long s_end, float gamma,
                                                unsigned int* __restrict__ delta_bits,
                                                unsigned int* __restrict__ changed) {
    __shared__ float lds_tab[PI_GRID.tab_len];
    __shared__ float lds_red[PI_BLOCK / 64];
    for (int i = threadIdx.x; i < PI_GRID.tab_len; i += PI_BLOCK) lds_tab[i] = tab[i];
    __syncthreads();

    const PiChunks ck = pi_chunks(s_end - s_begin);
    unsigned int n_changed = 0;
    float dmax = 0.0f;
    for (long long cl = ck.j; cl < ck.span; cl += ck.step) {
        const long long chunk = ck.x * ck.span + cl;
        if (chunk >= ck.n_chunks) break;
        const long long s = s_begin + chunk * PI_BLOCK + threadIdx.x;
        if (s >= s_end) continue;
        if (term[s]) {
            if (WRITE_V) Vn[s] = V[s];
            continue;
        }
        float x[PI_D];
        pi_state_coords((unsigned int)s, lds_tab, x);
        float best_q = -1.0e30f;
        int best = 0;
        for (int a = 0; a < PI_NA; ++a) {
            const float q = pi_backup(x, tab[PI_TAB_ACT + a], V, tab, gamma);
            if (q > best_q) { best_q = q; best = a; }
        }
        const int old = policy[s];
        policy[s] = best;
        n_changed += (old != best) ? 1u : 0u;
        if (WRITE_V) {
            Vn[s] = best_q;
            const float dlt = fabsf(best_q - V[s]);
            dmax = dlt > dmax ? dlt : dmax;
        }
    }
    if (changed != nullptr) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) n_changed += __shfl_xor(n_changed, o, 64);
        if ((threadIdx.x & 63) == 0 && n_changed != 0u)
            atomicAdd(changed + ((blockIdx.x * (PI_BLOCK / 64) + (threadIdx.x >> 6)) & (PI_NSLOT - 1)),
                      n_changed);
    }
    if (WRITE_V && delta_bits != nullptr) pi_block_max_to(dmax, lds_red, delta_bits);
}

extern "C" __global__ void PI_LB_IMPROVE
pi_improve_sweep_kernel(const float* __restrict__ V, int* __restrict__ policy,
                        const unsigned char* __restrict__ term, const float* __restrict__ tab,
                        long long s_begin, long long s_end, float gamma,
                        unsigned int* __restrict__ changed) {
    pi_improve_body<false>(V, nullptr, policy, term, tab, s_begin, s_end, gamma, nullptr, changed);
}

extern "C" __global__ void PI_LB_IMPROVE
pi_value_sweep_kernel(const float* __restrict__ V, float* __restrict__ Vn, int* __restrict__ policy,
                      const unsigned char* __restrict__ term, const float* __restrict__ tab,
                      long long s_begin, long long s_end, float gamma,
                      unsigned int* __restrict__ delta_bits, unsigned int* __restrict__ changed) {
    pi_improve_body<true>(V, Vn, policy, term, tab, s_begin, s_end, gamma, delta_bits, changed);
}

// Fold the PI_NSLOT accumulator slots into the caller's scalars and clear them for the next
// launch.  One wave; launched by the host right after a sweep that asked for a residual and/or a
// changed-count (either pointer pair may be null).
extern "C" __global__ void __launch_bounds__(64)
pi_finalize_kernel(unsigned int* __restrict__ delta_slots, float* __restrict__ delta_out,
                   unsigned int* __restrict__ changed_slots, unsigned int* __restrict__ changed_out) {
    const int lane = threadIdx.x;
    if (delta_slots != nullptr) {
        unsigned int m = 0u;
        for (int i = lane; i < PI_NSLOT; i += 64) { m = max(m, delta_slots[i]); delta_slots[i] = 0u; }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned int)__shfl_xor((int)m, o, 64));
        if (lane == 0) *delta_out = __uint_as_float(m);
    }
    if (changed_slots != nullptr) {
        unsigned int c = 0u;
        for (int i = lane; i < PI_NSLOT; i += 64) { c += changed_slots[i]; changed_slots[i] = 0u; }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) c += (unsigned int)__shfl_xor((int)c, o, 64);
        if (lane == 0) *changed_out = c;
    }
}

// ---- which dim-0 planes of V can the states of a range read? -----------------------------
// For every state in [s_begin, s_end) and EVERY action: the plane (dimension-0 index) of the
// successor's cell and the one above it are marked in a bitmap of g_0 bits.  Policy-independent,
// so it is computed once; the multi-GPU host uses it to exchange only the planes a rank's
// shard can reach (halo exchange) instead of all-gathering the whole V after every sweep.
#define PI_PLANE_WORDS ((PI_GRID.g[0] + 31) / 32)
extern "C" __global__ void __launch_bounds__(PI_BLOCK)
pi_reach_planes_kernel(const unsigned char* __restrict__ term, const float* __restrict__ tab,
                       long long s_begin, long long s_end, unsigned int* __restrict__ bitmap) {
    __shared__ float lds_tab[PI_GRID.tab_len];
    __shared__ unsigned int lds_bits[PI_PLANE_WORDS];
    for (int i = threadIdx.x; i < PI_GRID.tab_len; i += PI_BLOCK) lds_tab[i] = tab[i];
    for (int i = threadIdx.x; i < PI_PLANE_WORDS; i += PI_BLOCK) lds_bits[i] = 0u;
    __syncthreads();
    const PiChunks ck = pi_chunks(s_end - s_begin);
    for (long long cl = ck.j; cl < ck.span; cl += ck.step) {
        const long long chunk = ck.x * ck.span + cl;
        if (chunk >= ck.n_chunks) break;
        const long long s = s_begin + chunk * PI_BLOCK + threadIdx.x;
        if (s >= s_end) continue;
        if (term[s]) continue;
        float x[PI_D];
        pi_state_coords((unsigned int)s, lds_tab, x);
        int last = -1;
        for (int a = 0; a < PI_NA; ++a) {
            float ns[PI_D], reward, fr[PI_D];
            bool done;
            pi_dynamics(x, tab[PI_TAB_ACT + a], ns, &reward, &done);
            if (done) continue;
            int base;
            pi_locate(ns, tab, base, fr);
            const int p = base / PI_GRID.stride[0];
            if (p != last) {
                atomicOr(&lds_bits[p >> 5], 1u << (p & 31));
                atomicOr(&lds_bits[(p + 1) >> 5], 1u << ((p + 1) & 31));
                last = p;
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < PI_PLANE_WORDS; i += PI_BLOCK)
        if (lds_bits[i] != 0u) atomicOr(&bitmap[i], lds_bits[i]);
}

// ---- plugin probe (parity tests for the env dynamics and the interpolation) ------
// One thread per query point: runs step_dynamics on (state, action) and, when `idxs`
// is given, the interpolation of an arbitrary point.  Not on the hot path.
extern "C" __global__ void __launch_bounds__(PI_BLOCK)
pi_probe_step_kernel(const float* __restrict__ states, const float* __restrict__ acts,
                     float* __restrict__ next, float* __restrict__ reward,
                     unsigned char* __restrict__ done, long long m) {
    const long long k = (long long)blockIdx.x * PI_BLOCK + threadIdx.x;
    if (k >= m) return;
    float s[PI_D], ns[PI_D], r;
    bool t;
#pragma unroll
    for (int d = 0; d < PI_D; ++d) s[d] = states[k * PI_D + d];
    pi_dynamics(s, acts[k], ns, &r, &t);
#pragma unroll
    for (int d = 0; d < PI_D; ++d) next[k * PI_D + d] = ns[d];
    reward[k] = r;
    done[k] = t ? 1 : 0;
}

extern "C" __global__ void __launch_bounds__(PI_BLOCK)
pi_probe_interp_kernel(const float* __restrict__ pts, const float* __restrict__ tab,
                       int* __restrict__ idxs, float* __restrict__ wgts, long long m) {
    const long long k = (long long)blockIdx.x * PI_BLOCK + threadIdx.x;
    if (k >= m) return;
    float p[PI_D], fr[PI_D], w[PI_C];
#pragma unroll
    for (int d = 0; d < PI_D; ++d) p[d] = pts[k * PI_D + d];
    int base;
    pi_locate(p, tab, base, fr);
    pi_corner_weights(fr, w);
#pragma unroll
    for (int c = 0; c < PI_C; ++c) {
        idxs[k * PI_C + c] = base + pi_corner_offset(c);
        wgts[k * PI_C + c] = w[pi_corner_mask(c)];
    }
}
