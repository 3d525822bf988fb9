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

// ---- one-launch kernels of launch-bound grids --------------------------------------------------------------------------
// pi_eval_resident_kernel / pi_run_resident_kernel (grids one CU's LDS holds), pi_eval_flow_kernel (dataflow evaluation) and
// pi_xcd_kernel (XCD-local evaluation / whole run) live in csrc/pi_onelaunch_kernels.hip, appended to this unit only
// for handles whose grid qualifies (round 6).

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
