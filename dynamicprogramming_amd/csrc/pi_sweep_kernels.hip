// pi_sweep_kernels.hip — Bellman-backup sweep kernels for gfx950 (MI355X, CDNA4).
//
// This is a device-code TEMPLATE, never compiled on its own.  libpi_mi355.so
// (pi_api.cpp) builds one translation unit per (grid, action count, env):
//
//     <generated #defines: PI_D, PI_NA, PI_GRID_INIT, PI_LO_INIT, PI_SPAN_INIT, PI_RCP_INIT,
//                          PI_FASTDIV_INIT>
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
// One thread owns one state, as in the reference.  Beyond the reference: a fused value-iteration
// sweep, an LDS-resident kernel that runs whole batches of sweeps (or a whole policy evaluation)
// of a small grid in one launch, and the reach probes the multi-GPU exchange is planned from.
//
// What bounds these kernels on MI355X, and what the code does about it (DESIGN.md sections 4 and 5, profiles/r04):
// no single unit.  On the 80^4 evaluation sweep the most utilised one is VALU issue at ~0.5 of the guide's 1 228.8 G
// wave64 instructions/s (0.67 of the 950 G/s the chip sustains); its vector loads need ~0.4 of the launch at the
// 16-cycle-per-load floor of the vector-memory path, HBM-side traffic is ~0.25 of 8 TB/s; in 6-D the load floor leads
// (0.56).  Why the units' times ADD instead of overlapping (per-wave timeline, tools/phase_timeline.py,
// profiles/r04/phase_timeline_c4.txt): a wave's phases are one dependent chain — inputs, ~400 VALU instructions of
// dynamics and cell search that issue one per ~6 cycles (latency-, not throughput-bound: a lone wave cannot issue
// faster than one per ~5), 2^(D-1) loads that queue at the TA, a ~1 200-cycle round trip, the fmaf chain — and only 5.6
// of a SIMD's 8 wave slots are occupied on average (the arbiter serves a workgroup's waves oldest-first, so they finish
// over a 6 000-cycle window and a 1 024-thread workgroup frees its slots only as a whole), of which ~2 are in an
// arithmetic phase at any time: 2 waves x 1/6 instruction per cycle is the VALU utilisation measured.  More waves per
// SIMD do not exist, fewer are slower (profiles/r03/negative_results.txt (8)); persistent workgroups, prefetching the
// next workgroup's inputs, dropping the LDS table and its barrier, and per-wave priorities were all built and are
// slower (profiles/r04/negative_results.txt).  VALU issue is ~2.2 cycles per wave64 instruction per SIMD for ANY
// mix that is at least half fp32 fma/mul/add (tools/valu_issue_bench.hip, profiles/r03/valu_issue.txt).  So the
// code keeps the instruction count low — in the SCALAR prologue too: a wave requests nothing before it — keeps every
// load in flight early, and touches as few lines per wave as the dynamics allow:
//   * the interpolation's IEEE divisions (s - lo) / (hi - lo) have a per-dimension constant
//     divisor: they run as a * rcp, one residual fma and one correction fma — bit-identical to
//     the IEEE quotient for every dividend in [2^-40, 2^40], which the host PROVES per divisor
//     by exhaustive enumeration before it enables the path (pi_api.cpp, validate_fast_division);
//     lanes outside that range take the IEEE division; on the proven path the border clamp is the
//     fma's free output modifier instead of a min and a max;
//   * every address is a 32-bit byte offset from a scalar base (global_load ... v, s[..]), so
//     there is no 64-bit vector address arithmetic;
//   * all loads a state needs up front (mask, policy, old value) are issued before the first
//     wait, and a workgroup sweeps several consecutive chunks, prefetching the next chunk's
//     while it computes the current one;
//   * the residual and the changed-count are reduced inside a wave (shuffles) and leave through
//     one atomic per wave into one of 256 slots: no workgroup barrier after the prologue.
//
// Arithmetic contract (bit-exact against oracle/pi_oracle.cpp): fp32 throughout, no
// contraction, the fmaf chain over corners in ascending corner order from 0.0f,
// `reward + gamma * E` as mul then add, strict `>` argmax from -1.0e30f (lowest index wins
// ties, NaN never wins).

#define PI_C (1 << PI_D)
// Threads per workgroup = states per chunk, per kernel family (generated: 256, 512 or 1024).
// Big workgroups put the gathers of 4-16 adjacent waves behind ONE vector L1: their corner rows
// overlap, which is worth 8 % on the 80^4 evaluation sweep (profiles/r02/block_cpw_sweep.txt).
#ifndef PI_BLOCK_EVAL
#define PI_BLOCK_EVAL 256
#endif
#ifndef PI_BLOCK_IMPROVE
#define PI_BLOCK_IMPROVE 256
#endif
#define PI_BLOCK 256        // probes and the reach kernel
#define PI_NXCD 8
#define PI_NSLOT 256       // accumulator slots for the residual / the changed-count

// ---- compile-time grid geometry ------------------------------------------------
struct PiGrid {
    int g[PI_D];
    int stride[PI_D];
    int bins_off[PI_D];   // offset of dimension d's bin table inside the float table
    int tab_len;
    long long n;
};
__host__ __device__ constexpr PiGrid pi_make_grid() {
    PiGrid r = {};
    const int g[PI_D] = PI_GRID_INIT;
    for (int d = 0; d < PI_D; ++d) r.g[d] = g[d];
    r.stride[PI_D - 1] = 1;
    for (int d = PI_D - 2; d >= 0; --d) r.stride[d] = r.stride[d + 1] * r.g[d + 1];
    int off = PI_NA;
    for (int d = 0; d < PI_D; ++d) { r.bins_off[d] = off; off += r.g[d]; }
    r.tab_len = off;
    r.n = 1;
    for (int d = 0; d < PI_D; ++d) r.n *= r.g[d];
    return r;
}
constexpr PiGrid PI_GRID = pi_make_grid();
// Float table (device buffer `tab`, built by pi_create):  actions[NA] | bins_0 | bins_1 | ...
#define PI_TAB_ACT 0
// Interpolation constants, exact float32 values printed as hex-float literals by the host:
// bounds_low, hi - lo (rounded once, as the reference's kernels compute it), and the correctly
// rounded reciprocal of that span.  PI_FASTDIV[d] = 1 when the host has proven the
// reciprocal-multiply division exact for dimension d's divisor.
constexpr float PI_LO[PI_D] = PI_LO_INIT;
constexpr float PI_SPAN[PI_D] = PI_SPAN_INIT;
constexpr float PI_RCP[PI_D] = PI_RCP_INIT;
constexpr int PI_FASTDIV[PI_D] = PI_FASTDIV_INIT;
// V offsets as 32-bit BYTE offsets (4 n < 2^32) — otherwise 64-bit element indexing.
constexpr bool PI_OFF32 = PI_GRID.n * 4 < (1LL << 32);

// Memory order of the dimensions (pi_set_option 4).  Everything above — PI_GRID, PI_LO, ... and every flat state
// index — is in MEMORY order: memory dimension 0 is the slowest, PI_D - 1 the one lanes run along.  The USER's
// dimension d (the d-th argument of step_dynamics, the d-th key of bins_space) is memory dimension PI_MEM_OF[d];
// the identity unless the host chose another order for this grid.  Where a state lives is all the order changes:
// the arithmetic that has an order of its own — corner weights as products over the dimensions, the fmaf chain
// over the corners — stays in the USER's order, so results do not depend on it bit for bit.
#ifndef PI_MEM_OF_INIT
#if PI_D == 2
#define PI_MEM_OF_INIT {0, 1}
#elif PI_D == 4
#define PI_MEM_OF_INIT {0, 1, 2, 3}
#else
#define PI_MEM_OF_INIT {0, 1, 2, 3, 4, 5}
#endif
#endif
constexpr int PI_MEM_OF[PI_D] = PI_MEM_OF_INIT;

// Which dimension-bit of the partial-product index a corner number selects, in USER dimensions.
// 4D/6D: bit d of corner c <-> dimension d (:607, :1035).  2D is written out with
// dimension 1 toggling fastest (:201-209), i.e. the two bits are swapped.
__device__ __forceinline__ constexpr int pi_corner_mask(int c) {
#if PI_D == 2
    return ((c & 1) << 1) | ((c >> 1) & 1);
#else
    return c;
#endif
}
// The same corner as a mask over MEMORY dimensions (bit k <-> memory dimension k).
__device__ __forceinline__ constexpr int pi_mem_mask(int user_mask) {
    int m = 0;
    for (int d = 0; d < PI_D; ++d) m |= ((user_mask >> d) & 1) << PI_MEM_OF[d];
    return m;
}
__device__ __forceinline__ constexpr int pi_corner_offset(int c) {
    int m = pi_mem_mask(pi_corner_mask(c)), off = 0;
    for (int k = 0; k < PI_D; ++k) off += ((m >> k) & 1) * PI_GRID.stride[k];
    return off;
}

// Call the plugin with the arity the reference documents for each D (:11-15, :456-460, :869-874).  `sm` / `nm` hold
// the state and its successor in MEMORY order; the plugin sees its own (the user's) order — compile-time indices, i.e.
// register renaming.
#define PI_U(v, d) v[PI_MEM_OF[d]]
__device__ __forceinline__ void pi_dynamics(const float (&sm)[PI_D], float a, float (&nm)[PI_D],
                                            float* reward, bool* done) {
#if PI_D == 2
    step_dynamics(PI_U(sm, 0), PI_U(sm, 1), a, &PI_U(nm, 0), &PI_U(nm, 1), reward, done);
#elif PI_D == 4
    step_dynamics(PI_U(sm, 0), PI_U(sm, 1), PI_U(sm, 2), PI_U(sm, 3), a,
                  &PI_U(nm, 0), &PI_U(nm, 1), &PI_U(nm, 2), &PI_U(nm, 3), reward, done);
#elif PI_D == 6
    step_dynamics(PI_U(sm, 0), PI_U(sm, 1), PI_U(sm, 2), PI_U(sm, 3), PI_U(sm, 4), PI_U(sm, 5), a,
                  &PI_U(nm, 0), &PI_U(nm, 1), &PI_U(nm, 2), &PI_U(nm, 3), &PI_U(nm, 4), &PI_U(nm, 5), reward, done);
#else
#error "PI_D must be 2, 4 or 6"
#endif
}

// ---- checked build (PI_DEBUG_BOUNDS) ----------------------------------------------------
// PI_MI355_DEBUG=1 when the handle compiles its kernels: every index the sweeps derive from DATA — the
// action a state's policy entry names, the cell a successor falls in — is checked before it is used,
// a bad one is counted, remembered (the first: kind, flat state or cell, offending value) and replaced by
// index 0, so that a corrupted policy array or a plugin bug shows up as a report (pi_debug_report) instead of
// a wild read.  The GPU has no address sanitizer on this platform; the two-buffer Jacobi sweep is race-free
// by construction (reference :321-323), which leaves indices as the thing to check.  Off (0): no code at all.
#ifndef PI_DEBUG_BOUNDS
#define PI_DEBUG_BOUNDS 0
#endif
#if PI_DEBUG_BOUNDS
extern "C" __device__ unsigned int pi_debug_words[4] = {0u, 0u, 0u, 0u};   // faults | kind | where | value
__device__ __forceinline__ void pi_debug_fault(unsigned int kind, unsigned int where, unsigned int value) {
    if (atomicAdd(&pi_debug_words[0], 1u) == 0u) {
        pi_debug_words[1] = kind;
        pi_debug_words[2] = where;
        pi_debug_words[3] = value;
    }
}
#endif
// kind 1: policy[s] is not an action index (where = s, value = the entry)
__device__ __forceinline__ int pi_checked_action(int action, unsigned int s) {
#if PI_DEBUG_BOUNDS
    if ((unsigned int)action >= (unsigned int)PI_NA) {
        pi_debug_fault(1u, s, (unsigned int)action);
        return 0;
    }
#endif
    (void)s;
    return action;
}
// kind 2: a cell whose far corner lies outside the grid (where = the cell's flat index, value = 0)
__device__ __forceinline__ unsigned int pi_checked_cell(unsigned int base) {
#if PI_DEBUG_BOUNDS
    if ((unsigned long long)base + (unsigned long long)pi_corner_offset(PI_C - 1) >= (unsigned long long)PI_GRID.n) {
        pi_debug_fault(2u, base, 0u);
        return 0u;
    }
#endif
    return base;
}

// ---- interpolation ---------------------------------------------------------------
// (s - lo) / (hi - lo) for every dimension.  Fast path: q = a * rcp is within an ulp of the
// quotient, r = fma(-q, span, a) is the exact residual, fma(r, rcp, q) rounds to the IEEE
// quotient (Markstein); proven per divisor on the host for every float32 significand, so the
// only run-time condition is that no intermediate leaves the normal range: 2^-40 <= |a| < 2^40
// (the host requires 2^-30 <= span <= 2^30).  NaN dividends may take either path (NaN both ways).
constexpr bool pi_any_fastdiv() {
    for (int d = 0; d < PI_D; ++d) if (PI_FASTDIV[d]) return true;
    return false;
}
constexpr bool pi_all_fastdiv() {
    for (int d = 0; d < PI_D; ++d) if (!PI_FASTDIV[d]) return false;
    return true;
}

// One dimension of get_barycentric_*: clamp the grid coordinate n to [0, g-1] (NaN lands on the
// top border: fminf/fmaxf return the non-NaN operand), truncate, `frac = n - i`.
__device__ __forceinline__ void pi_cell_1d(float n, int d, unsigned int& base, float& fr) {
    const float top = (float)(PI_GRID.g[d] - 1);
    n = fmaxf(0.0f, fminf(n, top));
    const int i = min((int)n, PI_GRID.g[d] - 2);
    fr = n - (float)i;
    base += (unsigned int)i * (unsigned int)PI_GRID.stride[d];
}

// Cell of a continuous point: flat index of its lowest corner and the D fractional offsets.
// n_d = (s_d - lo_d) / (hi_d - lo_d) * (g_d - 1), divide then multiply, as the reference.
//
// Fast path (all lanes of practical interest): the divisor is a per-dimension constant, so
// q = a * rcp is within an ulp of the quotient, r = fma(-q, span, a) is the exact residual and
// fma(r, rcp, q) rounds to the IEEE quotient (Markstein) — proven per divisor on the host for
// every float32 significand; the run-time condition is only that nothing leaves the normal range:
// (2^-40 <= |a_d| or a_d == 0) and sum |a_d| < 2^40 (the host requires 2^-30 <= span <= 2^30).  A NaN or Inf
// coordinate fails the sum test, so the fast path never sees one and may clamp the QUOTIENT to
// [0, 1] with the fma's free output modifier instead of clamping n with a min and a max:
// for a finite q both give the same n (RN(q * top) is monotone in q and exact at q = 0 and 1).
// Every other lane takes the IEEE division and the min/max clamp.
__device__ __forceinline__ void pi_locate(const float (&ns)[PI_D], unsigned int& base,
                                          float (&fr)[PI_D]) {
    float a[PI_D];
#pragma unroll
    for (int d = 0; d < PI_D; ++d) a[d] = ns[d] - PI_LO[d];
    base = 0u;
    bool fast = false;
    if (pi_any_fastdiv()) {
        // |a_d| >= 2^-40 OR a_d == 0 in every proven dimension: a successor clamped exactly onto a
        // lower bound (envs that clip a position or a velocity) has a_d == 0, for which the fast
        // path is trivially exact (t = r = q = 0).  As one unsigned comparison per dimension:
        // 2 bits(|a|) - 2 wraps to 0xFFFFFFFE for +-0 and stays below the threshold for every other
        // value under 2^-40 (denormals included); huge values, Inf and NaN fail the sum test.
        float asum = 0.0f;
        unsigned int umin = 0xFFFFFFFFu;
#pragma unroll
        for (int d = 0; d < PI_D; ++d)
            if (PI_FASTDIV[d]) {
                asum += fabsf(a[d]);
                const unsigned int u = __float_as_uint(a[d]);
                umin = min(umin, (u + u) - 2u);
            }
        fast = (asum < 0x1p40f) & (umin >= 2u * 0x2B800000u - 2u);     // 0x2B800000 = bits(2^-40)
    }
    if (__builtin_expect(fast, 1)) {
#pragma unroll
        for (int d = 0; d < PI_D; ++d) {
            if (PI_FASTDIV[d]) {
                const float t = a[d] * PI_RCP[d];
                const float r = fmaf(-t, PI_SPAN[d], a[d]);
                const float q = __builtin_amdgcn_fmed3f(fmaf(r, PI_RCP[d], t), 0.0f, 1.0f);
                const float n = q * (float)(PI_GRID.g[d] - 1);
                const int i = min((int)n, PI_GRID.g[d] - 2);
                fr[d] = n - (float)i;
                base += (unsigned int)i * (unsigned int)PI_GRID.stride[d];
            } else {
                pi_cell_1d(a[d] / PI_SPAN[d] * (float)(PI_GRID.g[d] - 1), d, base, fr[d]);
            }
        }
    } else {
#pragma unroll
        for (int d = 0; d < PI_D; ++d)
            pi_cell_1d(a[d] / PI_SPAN[d] * (float)(PI_GRID.g[d] - 1), d, base, fr[d]);
    }
    base = pi_checked_cell(base);
}

// The 2^D corner weights from the fractional offsets.  The reference multiplies
// 1.0f * a_0 * a_1 * ... left to right for every corner; sharing the common prefixes is the
// same sequence of roundings.  w[] is indexed by the partial-product mask (bit d <-> dim d).
// `fr` is in MEMORY order; the products run over the USER's dimensions in ascending order.
__device__ __forceinline__ void pi_corner_weights(const float (&fr)[PI_D], float (&w)[PI_C]) {
    w[0] = 1.0f - PI_U(fr, 0);
    w[1] = PI_U(fr, 0);
#pragma unroll
    for (int k = 1; k < PI_D; ++k) {
        const float om = 1.0f - PI_U(fr, k);
#pragma unroll
        for (int m = (1 << k) - 1; m >= 0; --m) {
            w[m + (1 << k)] = w[m] * PI_U(fr, k);
            w[m] = w[m] * om;
        }
    }
}

// Two adjacent floats with 4-byte alignment: one global_load_dwordx2.
typedef float PiPair __attribute__((ext_vector_type(2)));
typedef PiPair PiPairU __attribute__((aligned(4)));
#define PI_NPAIR (PI_C / 2)

// Multilinear interpolation of V over the cell.  All 2^D values are requested first — the two
// corners along the last dimension are adjacent in memory and come as one 8-byte load — then
// the fmaf chain runs in ascending corner order from 0.0f like the reference's.
// vp[mb] holds the corners with slow-dimension mask mb (bits 0 .. D-2): .x = last-dim bit 0, .y = 1.
// Addressing: a 32-bit BYTE offset from the scalar table pointer (global_load ... v, s[..]) where
// 4 n < 2^32; corners that differ in the slow dimensions cost one 32-bit add each, the
// second-to-last dimension rides in the instruction's immediate offset.
__device__ __forceinline__ void pi_request_corners(const float* __restrict__ V, unsigned int base,
                                                   PiPair (&vp)[PI_NPAIR]) {
    constexpr int kNear = 1 << (PI_D - 2);
#pragma unroll
    for (int m = 0; m < kNear; ++m) {                // m: corner mask over dimensions 0 .. D-3
        int far = 0;
#pragma unroll
        for (int d = 0; d < PI_D - 2; ++d) far += ((m >> d) & 1) * PI_GRID.stride[d];
        const char* p;
        if (PI_OFF32) p = reinterpret_cast<const char*>(V) + (base * 4u + (unsigned int)far * 4u);
        else p = reinterpret_cast<const char*>(V + ((unsigned long long)base + (unsigned long long)far));
        vp[m] = *reinterpret_cast<const PiPairU*>(p);
        vp[m | kNear] = *reinterpret_cast<const PiPairU*>(p + (long)PI_GRID.stride[PI_D - 2] * 4);
    }
}
__device__ __forceinline__ float pi_combine_corners(const PiPair (&vp)[PI_NPAIR], const float (&fr)[PI_D]) {
    constexpr int kLast = 1 << (PI_D - 1);
    float w[PI_C];
    pi_corner_weights(fr, w);
    float e = 0.0f;
#pragma unroll
    for (int c = 0; c < PI_C; ++c) {
        const int mask = pi_corner_mask(c);               // user dimensions: the weight and the order of the chain
        const int mm = pi_mem_mask(mask);                 // memory dimensions: where the value was loaded to
        const float v = (mm & kLast) ? vp[mm & (kLast - 1)].y : vp[mm & (kLast - 1)].x;
        e = fmaf(w[mask], v, e);
    }
    return e;
}
// Between requesting its corner values and having consumed them a wave runs at raised issue priority
// (s_setprio 1; everything else at 0): a wave whose gather has come back is the one whose instructions
// free load-return slots and whose next chunk's loads keep the vector L1 fed, so it should not queue
// behind waves that are still in their arithmetic.  Pure scheduling, results unchanged; measured on one
// box: 25^6 evaluation 4.65 -> 4.20 ms, improvement 9.25 -> 9.03 ms (swing-up 8.06 -> 7.90 / 37.6 -> 36.4),
// 80^4 and 50^4 unchanged (profiles/r03/negative_results.txt (12) lists the variants that lose).
__device__ __forceinline__ float pi_interpolate(const float* __restrict__ V, unsigned int base,
                                                const float (&fr)[PI_D]) {
    PiPair vp[PI_NPAIR];
    pi_request_corners(V, base, vp);
    __builtin_amdgcn_s_setprio(1);
    const float e = pi_combine_corners(vp, fr);
    __builtin_amdgcn_s_setprio(0);
    return e;
}

__device__ __forceinline__ float pi_backup(const float (&s)[PI_D], float a,
                                           const float* __restrict__ V, float gamma) {
    float ns[PI_D], reward;
    bool done;
    pi_dynamics(s, a, ns, &reward, &done);
    float e = 0.0f;
    if (!done) {
        unsigned int base;
        float fr[PI_D];
        pi_locate(ns, base, fr);
        e = pi_interpolate(V, base, fr);
    }
    return reward + gamma * e;
}

// ---- state coordinates -------------------------------------------------------------
// Flat state index -> coordinates, through the LDS copy of the bin tables (the divisions are by
// compile-time constants: a multiply-high, a shift and a multiply-add each; on gfx950 a 32-bit
// integer multiply issues at the same rate as a compare or a shift, tools/valu_issue_bench.hip).
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

// ---- workgroup -> chunk schedule ---------------------------------------------------
// A workgroup sweeps `cpw` consecutive 256-state chunks (a "group").  Workgroups are dealt
// round-robin over the 8 XCDs (blockIdx % 8 shares an L2: tools/xcc_probe.hip), so XCD x is given
// the contiguous run of groups [x * span, (x + 1) * span): every private L2 sees one slab of V,
// and because the dispatcher starts workgroups in index order the states in flight on an XCD
// form one compact, advancing window (measured 5x less traffic past L2 than a grid-stride launch,
// profiles/r01).  Placement only affects speed.  Returns false when the workgroup has no group.
// No division by a run-time value here: the host launches exactly PI_NXCD * span workgroups (launch_blocks in
// pi_api.cpp), so span is gridDim.x / PI_NXCD, and "group g exists" is g * cpw < n_chunks.  The first version
// computed groups = ceil(n_chunks / cpw) and span = ceil(groups / 8) in 64-bit arithmetic: ~190 scalar instructions
// per wave in front of its first load, i.e. ~3 000 issue slots of the CU's ONE scalar unit per 1 024-thread workgroup
// — a fifth of a wave's life on the 80^4 evaluation sweep went by before it had requested anything
// (tools/phase_timeline.py, profiles/r04/phase_timeline_c4.txt).
template <int BLOCK>
__device__ __forceinline__ bool pi_first_chunk(long long count, int cpw, long long& chunk0,
                                               long long& n_chunks) {
    static_assert((BLOCK & (BLOCK - 1)) == 0, "chunk size must be a power of two (shift, not divide)");
    n_chunks = (count + BLOCK - 1) / BLOCK;
    const unsigned int span = gridDim.x / PI_NXCD;
    const unsigned int x = blockIdx.x % PI_NXCD, j = blockIdx.x / PI_NXCD;
    const unsigned int g = x * span + j;
    chunk0 = (long long)g * (long long)cpw;
    return chunk0 < n_chunks;
}

// The STRIP schedule (round 6).  Under the slab schedule above an XCD walks its run of groups along memory dimension 0:
// the planes of that dimension a successor cell spans (i0 + k, i0 + k + 1) are read a whole plane's sweep apart, and on a
// grid whose planes are megabytes (80^4: 2 MB each) the second read finds the line gone from a 4 MiB L2 that the V' stream
// shares — V was fetched 2.9x per sweep (profiles/r05/counters_bench_c4.json).  Here the groups are cut into PERIODS of
// `period` groups (the host picks a period = one plane of a slow memory dimension) and XCD x takes the x-th eighth — a
// strip — of EVERY period, period after period: what it touches between the two reads of a line is an eighth of a plane.
// The launch is two-dimensional: blockIdx.y is the period, blockIdx.x = 8 r + x the r-th workgroup of XCD x in it (the
// dispatcher deals workgroups to the XCDs round-robin in x-then-y order and gridDim.x is a multiple of 8, so x is still the
// XCD: tools/xcc_probe.hip) — no division.  Strip boundaries are floor((x * period + rot) / 8) with rot = 3 * period_number
// mod 8, so that a period that is not a multiple of 8 groups gives every XCD the same work on average (250 groups per plane
// on 80^4: strips of 31 or 32 groups, each XCD 31.25 on average) while a boundary moves by at most one group between periods;
// gridDim.x / 8 = ceil(period / 8) workgroups per XCD and period are launched, those beyond the strip's length leave at
// once.  `phase`: the launch's first group lies that many groups inside its period (ranges that do not start on a period
// boundary).  Placement only: every group is taken by exactly one workgroup (tests: pi_probe_coords walks the same schedule).
struct PiSched {
    int cpw;                    // chunks per workgroup
    unsigned int period;        // groups per period; 0 = the slab schedule (gridDim.y == 1)
    unsigned int phase;
};
template <int BLOCK>
__device__ __forceinline__ bool pi_first_chunk(long long count, const PiSched& sc, long long& chunk0,
                                               long long& n_chunks) {
    if (sc.period == 0u) return pi_first_chunk<BLOCK>(count, sc.cpw, chunk0, n_chunks);
    n_chunks = (count + BLOCK - 1) / BLOCK;
    const unsigned int x = blockIdx.x % PI_NXCD, r = blockIdx.x / PI_NXCD, p = blockIdx.y;
    const unsigned int lo8 = x * sc.period + ((p * 3u) & 7u);
    const unsigned int b0 = lo8 >> 3, b1 = (lo8 + sc.period) >> 3;
    if (r >= b1 - b0) return false;
    const int g = (int)(p * sc.period + b0 + r) - (int)sc.phase;             // groups < 2^31 (n < 2^31)
    chunk0 = (long long)g * (long long)sc.cpw;
    return g >= 0 && chunk0 < n_chunks;
}

// Bin tables and actions -> LDS.  All loads are issued before the first store.
template <int BLOCK>
__device__ __forceinline__ void pi_stage_table(const float* __restrict__ tab, float* lds_tab) {
    constexpr int kPer = (PI_GRID.tab_len + BLOCK - 1) / BLOCK;
    if (kPer <= 8) {
        float t[kPer];
#pragma unroll
        for (int j = 0; j < kPer; ++j) {
            const int i = j * BLOCK + (int)threadIdx.x;
            t[j] = tab[min(i, PI_GRID.tab_len - 1)];
        }
#pragma unroll
        for (int j = 0; j < kPer; ++j) {
            const int i = j * BLOCK + (int)threadIdx.x;
            if (i < PI_GRID.tab_len) lds_tab[i] = t[j];
        }
    } else {
        for (int i = threadIdx.x; i < PI_GRID.tab_len; i += BLOCK) lds_tab[i] = tab[i];
    }
}

__device__ __forceinline__ float pi_wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        float t = __shfl_xor(v, o, 64);
        v = t > v ? t : v;
    }
    return v;
}
// Residual of a wave -> one of the PI_NSLOT accumulator words (bit pattern of a float >= 0
// orders like an unsigned int; one word saturates at ~90 atomics per microsecond on MI355X,
// hence the slots; pi_finalize_kernel folds them).
template <int BLOCK>
__device__ __forceinline__ void pi_wave_max_to(float dmax, unsigned int* __restrict__ delta_bits) {
    dmax = pi_wave_max(dmax);
    if ((threadIdx.x & 63) == 0 && dmax > 0.0f)
        atomicMax(delta_bits + ((blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6)) & (PI_NSLOT - 1)),
                  __float_as_uint(dmax));
}
template <int BLOCK>
__device__ __forceinline__ void pi_wave_sum_to(unsigned int c, unsigned int* __restrict__ slots) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
    if ((threadIdx.x & 63) == 0 && c != 0u)
        atomicAdd(slots + ((blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6)) & (PI_NSLOT - 1)), c);
}

// Per-state inputs of a sweep, requested together so that one wait covers them.  Addresses are
// (scalar chunk base) + (32-bit lane offset): no vector 64-bit arithmetic.
struct PiStateIn {
    float v_old;
    int action;
    unsigned char term;
};
template <typename T>
__device__ __forceinline__ const T* pi_lane_ptr(const T* chunk_base, unsigned int lane) {
    return reinterpret_cast<const T*>(reinterpret_cast<const char*>(chunk_base) +
                                      lane * (unsigned int)sizeof(T));
}
// term == nullptr: the grid has no terminal states (the caller's promise; the solver passes it for envs
// whose mask is empty) — no mask stream at all.  need_old == false: nobody will look at the state's old
// value (no terminal state to copy, no residual asked of this launch), so it is not read either: on a
// grid without terminal states 24 of 25 evaluation sweeps stream 4 B per state (the action) instead of 9.
__device__ __forceinline__ PiStateIn pi_load_state(const float* __restrict__ V,
                                                   const int* __restrict__ policy,
                                                   const unsigned char* __restrict__ term,
                                                   long long sb, unsigned int lane, bool need_old) {
    PiStateIn in;
    // policy and mask are read exactly once per sweep: stream them (nt) so they do not displace
    // V lines, which neighbouring states re-read, from L2 / Infinity Cache.
    in.term = 0;
    if (term != nullptr) in.term = __builtin_nontemporal_load(pi_lane_ptr(term + sb, lane));
    in.action = __builtin_nontemporal_load(pi_lane_ptr(policy + sb, lane));
    in.v_old = 0.0f;
    if (need_old) in.v_old = *pi_lane_ptr(V + sb, lane);
    return in;
}
template <typename T>
__device__ __forceinline__ void pi_store_lane(T* chunk_base, unsigned int lane, T value) {
    *reinterpret_cast<T*>(reinterpret_cast<char*>(chunk_base) + lane * (unsigned int)sizeof(T)) = value;
}

// ---- policy evaluation sweep ---------------------------------------------------
// Vn[s] = r(s, pi(s)) + gamma * E[V](s')   for s in [s_begin, s_end); terminal: copy (keep_terminals != 0:
// Vn holds the terminal values already — not stored, and their old values not read).
// delta_bits (nullable): slots receiving the atomic max of the bit pattern of |Vn - V|.
// At most 80 SGPRs: measured on MI355X, one SGPR allocation granule more (TotalSGPRs 84 .. 96) and only
// seven waves fit a SIMD — the compiler still reports eight — so that ONE 1 024-thread workgroup is resident
// per CU instead of two and the 80^4 sweep goes from 0.395 to 0.458 ms (profiles/r03/negative_results.txt (15)).
extern "C" __global__ void __launch_bounds__(PI_BLOCK_EVAL) __attribute__((amdgpu_num_sgpr(80)))
pi_eval_sweep_kernel(const float* __restrict__ V, float* __restrict__ Vn,
                     const int* __restrict__ policy, const unsigned char* __restrict__ term,
                     const float* __restrict__ tab, long long s_begin, long long s_end,
                     float gamma, unsigned int* __restrict__ delta_bits, PiSched sc, int keep_terminals) {
    __shared__ float lds_tab[PI_GRID.tab_len];
    long long chunk0, n_chunks;
    if (!pi_first_chunk<PI_BLOCK_EVAL>(s_end - s_begin, sc, chunk0, n_chunks)) return;
    const int n_here = (int)(min(chunk0 + sc.cpw, n_chunks) - chunk0);

    const unsigned int tid = threadIdx.x;
    long long sb = s_begin + chunk0 * PI_BLOCK_EVAL;                     // first state of the chunk
    // lanes past s_end (tail of the last chunk) shadow the last valid state and store nothing
    unsigned int lane = min(tid, (unsigned int)(min(s_end - sb, (long long)PI_BLOCK_EVAL) - 1));
    // Old values are streamed when this launch reports a residual or has to copy the terminal states'
    // values into Vn.  keep_terminals != 0 (every sweep of a ping-pong batch but the first): Vn already
    // holds them — the first sweep copied them into one buffer out of the other — so terminal states are
    // simply not stored and no old value is needed (launch-uniform).
    const bool need_old = delta_bits != nullptr || (term != nullptr && keep_terminals == 0);
    PiStateIn nxt = pi_load_state(V, policy, term, sb, lane, need_old);
    pi_stage_table<PI_BLOCK_EVAL>(tab, lds_tab);
    __syncthreads();

    float dmax = 0.0f;
    for (int k = 0; k < n_here; ++k) {
        const PiStateIn cur = nxt;
        const long long sb_c = sb;
        const unsigned int lane_c = lane;
        if (k + 1 < n_here) {                                        // prefetch the next chunk's inputs
            sb += PI_BLOCK_EVAL;
            lane = min(tid, (unsigned int)(min(s_end - sb, (long long)PI_BLOCK_EVAL) - 1));
            nxt = pi_load_state(V, policy, term, sb, lane, need_old);
        }
        float nv = cur.v_old;
        if (!cur.term) {
            float x[PI_D], ns[PI_D], reward;
            pi_state_coords((unsigned int)sb_c + lane_c, lds_tab, x);
            const float a = lds_tab[PI_TAB_ACT + pi_checked_action(cur.action, (unsigned int)sb_c + lane_c)];
            bool done;
            pi_dynamics(x, a, ns, &reward, &done);
            float e = 0.0f;
            if (!done) {
                unsigned int base;
                float fr[PI_D];
                pi_locate(ns, base, fr);
                e = pi_interpolate(V, base, fr);
            }
            nv = reward + gamma * e;
        }
        if (tid == lane_c && (need_old || !cur.term)) {
            pi_store_lane(Vn + sb_c, lane_c, nv);
            const float dlt = fabsf(nv - cur.v_old);
            dmax = dlt > dmax ? dlt : dmax;
        }
    }
    if (delta_bits != nullptr) pi_wave_max_to<PI_BLOCK_EVAL>(dmax, delta_bits);
}

// ---- policy evaluation sweep over the LIVE states only -----------------------------------
// Grids with many terminal states (double cartpole 25^6: 35 %) leave lanes idle in every wave that straddles the
// border of a terminal region (16 % of that grid's waves), and an idle lane costs its wave's gather as much as a
// busy one (the vector L1 charges per instruction and quad).  `live` lists the non-terminal states in ascending
// order (built once per mask by the host: pi_prepare_mask); lane k of the launch takes state live[k], so every
// wave is full.  Only for sweeps that do not have to copy terminal values (keep_terminals of
// pi_eval_sweep_kernel: every sweep of a batch but the first) — terminal states are simply not visited; their
// residual contribution is 0 by definition.  Same arithmetic per state, hence the same bits.  The list index is
// fetched two chunks ahead and the state's inputs one chunk ahead, so the dependent load is off the critical path.
extern "C" __global__ void __launch_bounds__(PI_BLOCK_EVAL) __attribute__((amdgpu_num_sgpr(80)))
pi_eval_live_kernel(const float* __restrict__ V, float* __restrict__ Vn, const int* __restrict__ policy,
                    const int* __restrict__ live, const float* __restrict__ tab, long long n_live, float gamma,
                    unsigned int* __restrict__ delta_bits, PiSched sc) {
    __shared__ float lds_tab[PI_GRID.tab_len];
    long long chunk0, n_chunks;
    if (!pi_first_chunk<PI_BLOCK_EVAL>(n_live, sc, chunk0, n_chunks)) return;
    const int n_here = (int)(min(chunk0 + sc.cpw, n_chunks) - chunk0);
    const unsigned int tid = threadIdx.x;
    const long long kb0 = chunk0 * PI_BLOCK_EVAL;                       // first list entry of the workgroup
    const bool need_old = delta_bits != nullptr;                        // launch-uniform
    // lanes past the end of the list (tail of the last chunk) shadow its last entry and store nothing
    auto lane_of = [&](int k) {
        return min(tid, (unsigned int)(min(n_live - (kb0 + (long long)k * PI_BLOCK_EVAL), (long long)PI_BLOCK_EVAL) - 1));
    };
    auto entry = [&](int k, unsigned int lane) {
        return (unsigned int)__builtin_nontemporal_load(pi_lane_ptr(live + kb0 + (long long)k * PI_BLOCK_EVAL, lane));
    };
    unsigned int lane_cur = lane_of(0);
    unsigned int s_cur = entry(0, lane_cur);
    unsigned int lane_nxt = lane_cur, s_nxt = s_cur;
    if (n_here > 1) {
        lane_nxt = lane_of(1);
        s_nxt = entry(1, lane_nxt);
    }
    int a_cur = __builtin_nontemporal_load(policy + s_cur);
    float v_cur = 0.0f;
    if (need_old) v_cur = V[s_cur];
    pi_stage_table<PI_BLOCK_EVAL>(tab, lds_tab);
    __syncthreads();

    float dmax = 0.0f;
    for (int k = 0; k < n_here; ++k) {
        const unsigned int s = s_cur, lane_c = lane_cur;
        const int action = a_cur;
        const float v_old = v_cur;
        if (k + 1 < n_here) {                                           // inputs of the next chunk; index of the one after
            s_cur = s_nxt;
            lane_cur = lane_nxt;
            a_cur = __builtin_nontemporal_load(policy + s_cur);
            if (need_old) v_cur = V[s_cur];
            if (k + 2 < n_here) {
                lane_nxt = lane_of(k + 2);
                s_nxt = entry(k + 2, lane_nxt);
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
            const float dlt = fabsf(nv - v_old);
            dmax = dlt > dmax ? dlt : dmax;
        }
    }
    if (delta_bits != nullptr) pi_wave_max_to<PI_BLOCK_EVAL>(dmax, delta_bits);
}

// ---- per-evaluation list: live states whose successor is not terminal ----------------------
// Under a FIXED policy a live state whose successor (s, pi(s)) is terminal has V'(s) = reward + gamma * 0 in every
// sweep of the evaluation, whatever V is: once both Jacobi buffers hold that value (after the evaluation's first two
// sweeps) the state need not be visited again until the policy changes — its residual contribution is 0.  This
// kernel filters the live list down to the states that DO bootstrap (double cartpole 25^6: ~85-90 % of the live
// ones), keeping the ascending order the sweeps' XCD-aware chunk schedule relies on (an unordered append — blocks in
// completion order — made the sweeps 30 % SLOWER): pass 0 counts the survivors of every 256-entry block, a scan
// turns the counts into offsets, pass 1 repeats the test and writes the survivors at their block's offset.
extern "C" __global__ void __launch_bounds__(PI_BLOCK)
pi_policy_list_kernel(const int* __restrict__ live, long long n_live, const int* __restrict__ policy,
                      const float* __restrict__ tab, unsigned long long* __restrict__ block_slots,
                      int* __restrict__ out, int pass) {
    __shared__ float lds_tab[PI_GRID.tab_len];
    __shared__ unsigned int wave_count[PI_BLOCK / 64];
    pi_stage_table<PI_BLOCK>(tab, lds_tab);
    __syncthreads();
    const long long k = (long long)blockIdx.x * PI_BLOCK + threadIdx.x;
    bool keep = false;
    int s = 0;
    if (k < n_live) {
        s = live[k];
        float x[PI_D], ns[PI_D], reward;
        pi_state_coords((unsigned int)s, lds_tab, x);
        bool done;
        pi_dynamics(x, lds_tab[PI_TAB_ACT + pi_checked_action(policy[s], (unsigned int)s)], ns, &reward, &done);
        keep = !done;
    }
    const unsigned long long votes = __ballot(keep);
    const unsigned int lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    if (lane == 0) wave_count[wave] = (unsigned int)__popcll(votes);
    __syncthreads();
    if (pass == 0) {
        if (threadIdx.x == 0) {
            unsigned int total = 0;
            for (int w = 0; w < PI_BLOCK / 64; ++w) total += wave_count[w];
            block_slots[blockIdx.x] = total;
        }
    } else if (keep) {
        unsigned int before = 0;
        for (unsigned int w = 0; w < wave; ++w) before += wave_count[w];
        before += (unsigned int)__popcll(votes & ((1ull << lane) - 1ull));
        out[block_slots[blockIdx.x] + before] = s;
    }
}
// In-place exclusive scan of `count` block counts (one workgroup of 1 024 threads walks them 1 024 at a time with a
// running carry); slots[count] receives the total.
extern "C" __global__ void __launch_bounds__(1024)
pi_scan_slots_kernel(unsigned long long* __restrict__ slots, long long count) {
    __shared__ unsigned long long part[1024];
    __shared__ unsigned long long carry;
    if (threadIdx.x == 0) carry = 0ull;
    __syncthreads();
    for (long long base = 0; base < count; base += 1024) {
        const long long i = base + threadIdx.x;
        const unsigned long long mine = i < count ? slots[i] : 0ull;
        part[threadIdx.x] = mine;
        __syncthreads();
        for (unsigned int step = 1; step < 1024u; step <<= 1) {              // inclusive Hillis-Steele scan
            const unsigned long long add = threadIdx.x >= step ? part[threadIdx.x - step] : 0ull;
            __syncthreads();
            part[threadIdx.x] += add;
            __syncthreads();
        }
        if (i < count) slots[i] = carry + part[threadIdx.x] - mine;
        __syncthreads();
        if (threadIdx.x == 1023) carry += part[1023];
        __syncthreads();
    }
    if (threadIdx.x == 0) slots[count] = carry;
}

// ---- the live-state list of pi_prepare_mask, built on the device ----------------------------------------------------
// The non-terminal states of [s_begin, s_end) in ascending order, without the mask ever leaving the device (round 4 copied
// it to the host and walked it there: 244 MB and a 244 M-iteration loop at 25^6).  Pass 0: every wave's ballot of "live"
// is one word of the bitmap (word k covers states [w0 + 64 k, w0 + 64 k + 64), w0 = s_begin rounded down to 64; the host
// keeps the bitmap — 1 bit per state — for the positions of arbitrary sub-ranges), and every 256-state block leaves
// its count in a slot, packed with the number of its waves that have a live lane at all (high half: what the "is the
// list worth it" test needs).  pi_scan_slots_kernel turns the counts into offsets; pass 1 repeats the ballots and writes
// every live state at its block's offset + the live lanes before it.
extern "C" __global__ void __launch_bounds__(PI_BLOCK)
pi_mask_list_kernel(const unsigned char* __restrict__ term, long long s_begin, long long s_end, long long w0,
                    unsigned long long* __restrict__ bits, unsigned long long* __restrict__ block_slots,
                    int* __restrict__ out, int pass) {
    __shared__ unsigned int wave_count[PI_BLOCK / 64];
    const long long s = w0 + (long long)blockIdx.x * PI_BLOCK + threadIdx.x;
    const bool live = s >= s_begin && s < s_end && !term[s];
    const unsigned long long votes = __ballot(live);
    const unsigned int lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    if (lane == 0) wave_count[wave] = (unsigned int)__popcll(votes);
    if (pass == 0 && lane == 0 && s < s_end) bits[(s - w0) >> 6] = votes;
    __syncthreads();
    if (pass == 0) {
        if (threadIdx.x == 0) {
            unsigned long long total = 0ull;
            for (int w = 0; w < PI_BLOCK / 64; ++w)
                total += (unsigned long long)wave_count[w] + ((unsigned long long)(wave_count[w] != 0u) << 32);
            block_slots[blockIdx.x] = total;
        }
    } else if (live) {
        unsigned int before = (unsigned int)(block_slots[blockIdx.x] & 0xFFFFFFFFull);
        for (unsigned int w = 0; w < wave; ++w) before += wave_count[w];
        before += (unsigned int)__popcll(votes & ((1ull << lane) - 1ull));
        out[before] = (int)s;
    }
}

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

// ---- greedy policy improvement sweep -------------------------------------------
// policy[s] = argmax_a [ r(s,a) + gamma * E[V](s'_a) ], first maximum wins; terminal
// states keep their entry.  changed (nullable): slots counting the entries that changed.
// WRITE_V (value-iteration sweep, the fused form the reference's README sketches at :790-799
// but does not implement): also Vn[s] = max_a Q(s,a) (terminal: copy) and the residual.
// The argmax is a serial loop per lane on purpose: everything in step_dynamics that does not
// depend on the action (in the double pendulum: all seven sin/cos and both angle wraps) is
// hoisted out of it by the compiler; spreading the actions over lanes would redo that work
// n_actions times.  The action values come from LDS.
// Corner reuse across the action loop (below) pays where the action set is dense enough for neighbouring
// actions to share cells, and costs registers: measured on MI355X 80^4 x 11 actions -4.7 %, 25^6 x 9 -3.5 %,
// 50^4 x 5 and 200^2 x 21 unchanged, 25^6 x 3 +15 % (177 instead of 146 VGPRs: two waves per SIMD instead of three).
#ifndef PI_IMPROVE_REUSE
#define PI_IMPROVE_REUSE (PI_D >= 4 && PI_NA >= 8)
#endif
// The greedy action of one state: argmax_a r(s, a) + gamma E[V](s'), strict '>' from -1.0e30f in ascending
// action order (reference :262-280).  Shared by the state-order sweeps and the live-list improvement sweep.
__device__ __forceinline__ void pi_best_action(const float (&x)[PI_D], const float* lds_tab,
                                               const float* __restrict__ V, float gamma, int& best_out,
                                               float& best_q_out) {
    float best_q = -1.0e30f;
    int best = 0;
#if PI_IMPROVE_REUSE
    // Neighbouring actions often land in the same cell (80^4: 41 % of consecutive pairs, 25^6 swing-up
    // 43 %): the corner values of the cell the lane looked at last stay in registers and only lanes
    // whose cell changed issue loads — the vector L1 charges per distinct line among the ACTIVE lanes
    // of each lane quad (profiles/r03/negative_results.txt (16)).
    PiPair vp[PI_NPAIR];
#pragma unroll
    for (int p = 0; p < PI_NPAIR; ++p) vp[p] = PiPair{0.0f, 0.0f};
    unsigned int held = 0xffffffffu;
    for (int a = 0; a < PI_NA; ++a) {
        float ns[PI_D], reward;
        bool done;
        pi_dynamics(x, lds_tab[PI_TAB_ACT + a], ns, &reward, &done);
        float e = 0.0f;
        if (!done) {
            unsigned int base;
            float fr[PI_D];
            pi_locate(ns, base, fr);
            if (base != held) {
                pi_request_corners(V, base, vp);
                held = base;
            }
            __builtin_amdgcn_s_setprio(1);
            e = pi_combine_corners(vp, fr);
            __builtin_amdgcn_s_setprio(0);
        }
        const float q = reward + gamma * e;
        if (q > best_q) { best_q = q; best = a; }
    }
#else
    for (int a = 0; a < PI_NA; ++a) {
        const float q = pi_backup(x, lds_tab[PI_TAB_ACT + a], V, gamma);
        if (q > best_q) { best_q = q; best = a; }
    }
#endif
    best_out = best;
    best_q_out = best_q;
}

template <bool WRITE_V>
__device__ __forceinline__ void pi_improve_body(const float* __restrict__ V, float* __restrict__ Vn,
                                                int* __restrict__ policy,
                                                const unsigned char* __restrict__ term,
                                                const float* __restrict__ tab, long long s_begin,
                                                long long s_end, float gamma,
                                                unsigned int* __restrict__ delta_bits,
                                                unsigned int* __restrict__ changed, PiSched sc) {
    __shared__ float lds_tab[PI_GRID.tab_len];
    long long chunk0, n_chunks;
    if (!pi_first_chunk<PI_BLOCK_IMPROVE>(s_end - s_begin, sc, chunk0, n_chunks)) return;
    const int n_here = (int)(min(chunk0 + sc.cpw, n_chunks) - chunk0);

    const unsigned int tid = threadIdx.x;
    long long sb = s_begin + chunk0 * PI_BLOCK_IMPROVE;
    unsigned int lane = min(tid, (unsigned int)(min(s_end - sb, (long long)PI_BLOCK_IMPROVE) - 1));
    // the old value is only needed by the value sweep (residual, terminal copy)
    const bool need_old = WRITE_V && (delta_bits != nullptr || term != nullptr);
    PiStateIn nxt = pi_load_state(V, policy, term, sb, lane, need_old);
    pi_stage_table<PI_BLOCK_IMPROVE>(tab, lds_tab);
    __syncthreads();

    unsigned int n_changed = 0;
    float dmax = 0.0f;
    for (int k = 0; k < n_here; ++k) {
        const PiStateIn cur = nxt;
        const long long sb_c = sb;
        const unsigned int lane_c = lane;
        if (k + 1 < n_here) {
            sb += PI_BLOCK_IMPROVE;
            lane = min(tid, (unsigned int)(min(s_end - sb, (long long)PI_BLOCK_IMPROVE) - 1));
            nxt = pi_load_state(V, policy, term, sb, lane, need_old);
        }
        const bool live = tid == lane_c;
        if (!cur.term) {
            float x[PI_D];
            pi_state_coords((unsigned int)sb_c + lane_c, lds_tab, x);
            float best_q;
            int best;
            pi_best_action(x, lds_tab, V, gamma, best, best_q);
            if (live) {
                pi_store_lane(policy + sb_c, lane_c, best);
                n_changed += (cur.action != best) ? 1u : 0u;
                if (WRITE_V) {
                    pi_store_lane(Vn + sb_c, lane_c, best_q);
                    const float dlt = fabsf(best_q - cur.v_old);
                    dmax = dlt > dmax ? dlt : dmax;
                }
            }
        } else if (WRITE_V && live) {
            pi_store_lane(Vn + sb_c, lane_c, cur.v_old);
        }
    }
    if (changed != nullptr) pi_wave_sum_to<PI_BLOCK_IMPROVE>(n_changed, changed);
    if (WRITE_V && delta_bits != nullptr) pi_wave_max_to<PI_BLOCK_IMPROVE>(dmax, delta_bits);
}

extern "C" __global__ void __launch_bounds__(PI_BLOCK_IMPROVE)
pi_improve_sweep_kernel(const float* __restrict__ V, int* __restrict__ policy,
                        const unsigned char* __restrict__ term, const float* __restrict__ tab,
                        long long s_begin, long long s_end, float gamma,
                        unsigned int* __restrict__ changed, PiSched sc) {
    pi_improve_body<false>(V, nullptr, policy, term, tab, s_begin, s_end, gamma, nullptr, changed, sc);
}

// The improvement sweep over the LIVE states only (the list of pi_prepare_mask; see pi_eval_live_kernel): an
// improvement sweep never touches terminal states (:253), so every whole-grid launch may take this form.
extern "C" __global__ void __launch_bounds__(PI_BLOCK_IMPROVE)
pi_improve_live_kernel(const float* __restrict__ V, int* __restrict__ policy, const int* __restrict__ live,
                       const float* __restrict__ tab, long long n_live, float gamma,
                       unsigned int* __restrict__ changed, PiSched sc) {
    __shared__ float lds_tab[PI_GRID.tab_len];
    long long chunk0, n_chunks;
    if (!pi_first_chunk<PI_BLOCK_IMPROVE>(n_live, sc, chunk0, n_chunks)) return;
    const int n_here = (int)(min(chunk0 + sc.cpw, n_chunks) - chunk0);
    const unsigned int tid = threadIdx.x;
    const long long kb0 = chunk0 * PI_BLOCK_IMPROVE;
    auto lane_of = [&](int k) {
        return min(tid, (unsigned int)(min(n_live - (kb0 + (long long)k * PI_BLOCK_IMPROVE), (long long)PI_BLOCK_IMPROVE) - 1));
    };
    auto entry = [&](int k, unsigned int lane) {
        return (unsigned int)__builtin_nontemporal_load(pi_lane_ptr(live + kb0 + (long long)k * PI_BLOCK_IMPROVE, lane));
    };
    unsigned int lane_cur = lane_of(0);
    unsigned int s_cur = entry(0, lane_cur);
    pi_stage_table<PI_BLOCK_IMPROVE>(tab, lds_tab);
    __syncthreads();
    unsigned int n_changed = 0;
    for (int k = 0; k < n_here; ++k) {
        const unsigned int s = s_cur, lane_c = lane_cur;
        if (k + 1 < n_here) {
            lane_cur = lane_of(k + 1);
            s_cur = entry(k + 1, lane_cur);
        }
        const int old_action = __builtin_nontemporal_load(policy + s);    // needed only after the action loop
        float x[PI_D];
        pi_state_coords(s, lds_tab, x);
        float best_q;
        int best;
        pi_best_action(x, lds_tab, V, gamma, best, best_q);
        if (tid == lane_c) {
            policy[s] = best;
            n_changed += (old_action != best) ? 1u : 0u;
        }
    }
    if (changed != nullptr) pi_wave_sum_to<PI_BLOCK_IMPROVE>(n_changed, changed);
}

extern "C" __global__ void __launch_bounds__(PI_BLOCK_IMPROVE)
pi_value_sweep_kernel(const float* __restrict__ V, float* __restrict__ Vn, int* __restrict__ policy,
                      const unsigned char* __restrict__ term, const float* __restrict__ tab,
                      long long s_begin, long long s_end, float gamma,
                      unsigned int* __restrict__ delta_bits, unsigned int* __restrict__ changed,
                      PiSched sc) {
    pi_improve_body<true>(V, Vn, policy, term, tab, s_begin, s_end, gamma, delta_bits, changed, sc);
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

// ---- which planes of V can the states of a range read? ---------------------------------
// For every state in [s_begin, s_end) and EVERY action: along dimension PI_REACH_DIM (a kernel
// argument `dim`), the index of the successor's cell and the one above it are marked in a bitmap
// of g[dim] bits.  Policy-independent, so it is computed once; the multi-GPU host uses it to
// exchange only the slabs a rank's shard can reach (halo exchange) instead of all-gathering the
// whole V after every sweep, and to pick the dimension with the narrowest reach.
constexpr int pi_max_extent() {
    int m = 0;
    for (int d = 0; d < PI_D; ++d) m = PI_GRID.g[d] > m ? PI_GRID.g[d] : m;
    return m;
}
#define PI_PLANE_WORDS ((pi_max_extent() + 31) / 32)
extern "C" __global__ void __launch_bounds__(PI_BLOCK)
pi_reach_planes_kernel(const unsigned char* __restrict__ term, const float* __restrict__ tab,
                       long long s_begin, long long s_end, unsigned int* __restrict__ bitmap,
                       int dim, int cpw) {
    __shared__ float lds_tab[PI_GRID.tab_len];
    __shared__ unsigned int lds_bits[PI_PLANE_WORDS];
    long long chunk0, n_chunks;
    if (!pi_first_chunk<PI_BLOCK>(s_end - s_begin, cpw, chunk0, n_chunks)) return;
    const long long chunk_end = min(chunk0 + cpw, n_chunks);
    pi_stage_table<PI_BLOCK>(tab, lds_tab);
    for (int i = threadIdx.x; i < PI_PLANE_WORDS; i += PI_BLOCK) lds_bits[i] = 0u;
    __syncthreads();
    unsigned int stride_dim = 1u, g_dim = 1u;
#pragma unroll
    for (int d = 0; d < PI_D; ++d)
        if (d == dim) { stride_dim = (unsigned int)PI_GRID.stride[d]; g_dim = (unsigned int)PI_GRID.g[d]; }
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
            const int p = (int)((base / stride_dim) % g_dim);
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

// The same question at a finer grain: "units" = the leading `depth` dimensions flattened (depth 1:
// the planes of dimension 0 again; depth 2: the rows (i0, i1), each stride[1] contiguous states).
// Every env moves a position by dt * velocity, so which neighbouring planes a state reads depends
// on the sign and size of its velocity coordinate: the (plane, row) bitmap of a shard is a
// triangle, 2-4x smaller than the band of whole planes (profiles/r02/halo_granularity.txt), and the
// rows are still contiguous runs of V, so the exchange plan is the same planner on finer units.
constexpr int pi_unit_count(int depth) {
    int u = 1;
    for (int d = 0; d < depth; ++d) u *= PI_GRID.g[d];
    return u;
}
constexpr int pi_reach_depth_max() {
    return (PI_D >= 3 && (long long)PI_GRID.g[0] * PI_GRID.g[1] <= (1LL << 17)) ? 2 : 1;
}
#define PI_UNIT_WORDS ((pi_unit_count(pi_reach_depth_max()) + 31) / 32)
extern "C" __global__ void __launch_bounds__(PI_BLOCK)
pi_reach_units_kernel(const unsigned char* __restrict__ term, const float* __restrict__ tab,
                      long long s_begin, long long s_end, unsigned int* __restrict__ bitmap,
                      int depth, int cpw) {
    __shared__ float lds_tab[PI_GRID.tab_len];
    __shared__ unsigned int lds_bits[PI_UNIT_WORDS];
    long long chunk0, n_chunks;
    if (!pi_first_chunk<PI_BLOCK>(s_end - s_begin, cpw, chunk0, n_chunks)) return;
    const long long chunk_end = min(chunk0 + cpw, n_chunks);
    pi_stage_table<PI_BLOCK>(tab, lds_tab);
    for (int i = threadIdx.x; i < PI_UNIT_WORDS; i += PI_BLOCK) lds_bits[i] = 0u;
    __syncthreads();
    constexpr unsigned int g1 = PI_D >= 2 ? PI_GRID.g[1] : 1;
    constexpr unsigned int st0 = PI_GRID.stride[0], st1 = PI_D >= 2 ? PI_GRID.stride[1] : 1;
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
            const unsigned int c0 = base / st0;
            const int u = depth >= 2 ? (int)(c0 * g1 + (base / st1) % g1) : (int)c0;
            if (u == last) continue;
            last = u;
            // the cell's corners along the leading dimensions: +1 in each of them
            atomicOr(&lds_bits[u >> 5], 1u << (u & 31));
            if (depth >= 2) {
                atomicOr(&lds_bits[(u + 1) >> 5], 1u << ((u + 1) & 31));
                atomicOr(&lds_bits[(u + (int)g1) >> 5], 1u << ((u + (int)g1) & 31));
                atomicOr(&lds_bits[(u + (int)g1 + 1) >> 5], 1u << ((u + (int)g1 + 1) & 31));
            } else {
                atomicOr(&lds_bits[(u + 1) >> 5], 1u << ((u + 1) & 31));
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < PI_UNIT_WORDS; i += PI_BLOCK)
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
    float s[PI_D], ns[PI_D], r;                           // memory order inside, the caller's (user) order outside
    bool t;
#pragma unroll
    for (int d = 0; d < PI_D; ++d) PI_U(s, d) = states[k * PI_D + d];
    pi_dynamics(s, acts[k], ns, &r, &t);
#pragma unroll
    for (int d = 0; d < PI_D; ++d) next[k * PI_D + d] = PI_U(ns, d);
    reward[k] = r;
    done[k] = t ? 1 : 0;
}

extern "C" __global__ void __launch_bounds__(PI_BLOCK)
pi_probe_interp_kernel(const float* __restrict__ pts, int* __restrict__ idxs,
                       float* __restrict__ wgts, long long m) {
    const long long k = (long long)blockIdx.x * PI_BLOCK + threadIdx.x;
    if (k >= m) return;
    float p[PI_D], fr[PI_D], w[PI_C];                     // points in the user's order; indices are MEMORY-order flat indices
#pragma unroll
    for (int d = 0; d < PI_D; ++d) PI_U(p, d) = pts[k * PI_D + d];
    unsigned int base;
    pi_locate(p, base, fr);
    pi_corner_weights(fr, w);
#pragma unroll
    for (int c = 0; c < PI_C; ++c) {
        idxs[k * PI_C + c] = (int)base + pi_corner_offset(c);
        wgts[k * PI_C + c] = w[pi_corner_mask(c)];
    }
}

// State coordinates of arbitrary flat indices (test probe for pi_state_coords and the chunk walk):
// out[(s - s_begin) * D + d] for s in [s_begin, s_end), walked exactly like the sweeps walk it.
extern "C" __global__ void __launch_bounds__(PI_BLOCK)
pi_probe_coords_kernel(const float* __restrict__ tab, long long s_begin, long long s_end,
                       float* __restrict__ out, PiSched sc) {
    __shared__ float lds_tab[PI_GRID.tab_len];
    long long chunk0, n_chunks;
    if (!pi_first_chunk<PI_BLOCK>(s_end - s_begin, sc, chunk0, n_chunks)) return;
    const long long chunk_end = min(chunk0 + sc.cpw, n_chunks);
    pi_stage_table<PI_BLOCK>(tab, lds_tab);
    __syncthreads();
    for (long long chunk = chunk0; chunk < chunk_end; ++chunk) {
        const long long s = s_begin + chunk * PI_BLOCK + threadIdx.x;
        if (s >= s_end) continue;
        float x[PI_D];
        pi_state_coords((unsigned int)s, lds_tab, x);
#pragma unroll
        for (int d = 0; d < PI_D; ++d) out[(s - s_begin) * PI_D + d] = PI_U(x, d);      // columns in the user's order
    }
}
