// pi_sweep_kernels.hip — Bellman-backup sweep kernels for gfx950 (MI355X, CDNA4).
//
// This is a device-code TEMPLATE, never compiled on its own.  libpi_mi355.so
// (pi_api.cpp) builds one translation unit per (grid shape, action count, env):
//
//     <generated #defines: PI_D, PI_NA, PI_GRID_INIT, PI_MAXG>
//     <include/pi_math.h>  + #define sinf/cosf/fmodf -> pi_*   (deterministic math)
//     <the user's step_dynamics C string>                       (env plugin)
//     <this file>
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

// Expected next value: multilinear interpolation of V at ns over the 2^D cell corners.
__device__ __forceinline__ float pi_expected_value(const float (&ns)[PI_D],
                                                   const float* __restrict__ V,
                                                   const float* __restrict__ tab) {
    int base = 0;
    float fr[PI_D], om[PI_D];
#pragma unroll
    for (int d = 0; d < PI_D; ++d) {
        const float lo = tab[PI_TAB_LO + d];       // wave-uniform: scalar loads
        const float hi = tab[PI_TAB_HI + d];
        const float top = (float)(PI_GRID.g[d] - 1);
        float n = (ns[d] - lo) / (hi - lo) * top;
        n = fmaxf(0.0f, fminf(n, top));            // clamp-to-border; NaN lands on `top`
        int i = min((int)n, PI_GRID.g[d] - 2);
        fr[d] = n - (float)i;
        om[d] = 1.0f - fr[d];
        base += i * PI_GRID.stride[d];
    }
    // Corner weights.  The reference multiplies 1.0f * a_0 * a_1 * ... left to right for
    // every corner; sharing the common prefixes is the same sequence of roundings.
    float w[PI_C];
    w[0] = om[0];
    w[1] = fr[0];
#pragma unroll
    for (int k = 1; k < PI_D; ++k) {
#pragma unroll
        for (int m = (1 << k) - 1; m >= 0; --m) {
            w[m + (1 << k)] = w[m] * fr[k];
            w[m] = w[m] * om[k];
        }
    }
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
    if (!done) e = pi_expected_value(ns, V, tab);
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
struct PiChunks {
    long long n_chunks, span, j, step, x;
};
__device__ __forceinline__ PiChunks pi_chunks(long long count) {
    PiChunks c;
    c.n_chunks = (count + PI_BLOCK - 1) / PI_BLOCK;
    c.span = (c.n_chunks + PI_NXCD - 1) / PI_NXCD;
    c.x = blockIdx.x % PI_NXCD;
    c.j = blockIdx.x / PI_NXCD;
    c.step = gridDim.x / PI_NXCD;      // host launches a multiple of 8 workgroups
    return c;
}

__device__ __forceinline__ float pi_wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        float t = __shfl_xor(v, o, 64);
        v = t > v ? t : v;
    }
    return v;
}

// ---- policy evaluation sweep ---------------------------------------------------
// Vn[s] = r(s, pi(s)) + gamma * E[V](s')   for s in [s_begin, s_end); terminal: copy.
// delta_bits (nullable): atomic max of the bit pattern of max|Vn - V| (>= 0, so the
// unsigned order is the float order); the host zeroes it before the launch.
extern "C" __global__ void __launch_bounds__(PI_BLOCK)
pi_eval_sweep_kernel(const float* __restrict__ V, float* __restrict__ Vn,
                     const int* __restrict__ policy, const unsigned char* __restrict__ term,
                     const float* __restrict__ tab, long long s_begin, long long s_end,
                     float gamma, unsigned int* __restrict__ delta_bits) {
    __shared__ float lds_tab[PI_GRID.tab_len];
    __shared__ float lds_red[PI_BLOCK / 64];
    for (int i = threadIdx.x; i < PI_GRID.tab_len; i += PI_BLOCK) lds_tab[i] = tab[i];
    __syncthreads();

    const PiChunks ck = pi_chunks(s_end - s_begin);
    float dmax = 0.0f;
    for (long long cl = ck.j; cl < ck.span; cl += ck.step) {
        const long long chunk = ck.x * ck.span + cl;
        if (chunk >= ck.n_chunks) break;
        const long long s = s_begin + chunk * PI_BLOCK + threadIdx.x;
        if (s >= s_end) continue;
        const float v_old = V[s];
        float nv = v_old;
        if (!term[s]) {
            float x[PI_D];
            pi_state_coords((unsigned int)s, lds_tab, x);
            const float a = lds_tab[PI_TAB_ACT + policy[s]];
            nv = pi_backup(x, a, V, tab, gamma);
        }
        Vn[s] = nv;
        const float d = fabsf(nv - v_old);
        dmax = d > dmax ? d : dmax;
    }
    if (delta_bits != nullptr) {
        dmax = pi_wave_max(dmax);
        if ((threadIdx.x & 63) == 0) lds_red[threadIdx.x >> 6] = dmax;
        __syncthreads();
        if (threadIdx.x == 0) {
            float m = lds_red[0];
#pragma unroll
            for (int w = 1; w < PI_BLOCK / 64; ++w) m = lds_red[w] > m ? lds_red[w] : m;
            if (m > 0.0f) atomicMax(delta_bits, __float_as_uint(m));
        }
    }
}

// ---- greedy policy improvement sweep -------------------------------------------
// policy[s] = argmax_a [ r(s,a) + gamma * E[V](s'_a) ], first maximum wins; terminal
// states keep their entry.  changed (nullable): number of entries that changed.
extern "C" __global__ void __launch_bounds__(PI_BLOCK)
pi_improve_sweep_kernel(const float* __restrict__ V, int* __restrict__ policy,
                        const unsigned char* __restrict__ term, const float* __restrict__ tab,
                        long long s_begin, long long s_end, float gamma,
                        unsigned int* __restrict__ changed) {
    __shared__ float lds_tab[PI_GRID.tab_len];
    for (int i = threadIdx.x; i < PI_GRID.tab_len; i += PI_BLOCK) lds_tab[i] = tab[i];
    __syncthreads();

    const PiChunks ck = pi_chunks(s_end - s_begin);
    unsigned int n_changed = 0;
    for (long long cl = ck.j; cl < ck.span; cl += ck.step) {
        const long long chunk = ck.x * ck.span + cl;
        if (chunk >= ck.n_chunks) break;
        const long long s = s_begin + chunk * PI_BLOCK + threadIdx.x;
        if (s >= s_end) continue;
        if (term[s]) continue;
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
    }
    if (changed != nullptr) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) n_changed += __shfl_xor(n_changed, o, 64);
        if ((threadIdx.x & 63) == 0 && n_changed != 0u) atomicAdd(changed, n_changed);
    }
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
    int base = 0;
    float fr[PI_D], om[PI_D];
#pragma unroll
    for (int d = 0; d < PI_D; ++d) {
        const float lo = tab[PI_TAB_LO + d], hi = tab[PI_TAB_HI + d];
        const float top = (float)(PI_GRID.g[d] - 1);
        float n = (pts[k * PI_D + d] - lo) / (hi - lo) * top;
        n = fmaxf(0.0f, fminf(n, top));
        int i = min((int)n, PI_GRID.g[d] - 2);
        fr[d] = n - (float)i;
        om[d] = 1.0f - fr[d];
        base += i * PI_GRID.stride[d];
    }
    float w[PI_C];
    w[0] = om[0];
    w[1] = fr[0];
#pragma unroll
    for (int kk = 1; kk < PI_D; ++kk) {
#pragma unroll
        for (int mm = (1 << kk) - 1; mm >= 0; --mm) {
            w[mm + (1 << kk)] = w[mm] * fr[kk];
            w[mm] = w[mm] * om[kk];
        }
    }
#pragma unroll
    for (int c = 0; c < PI_C; ++c) {
        idxs[k * PI_C + c] = base + pi_corner_offset(c);
        wgts[k * PI_C + c] = w[pi_corner_mask(c)];
    }
}
