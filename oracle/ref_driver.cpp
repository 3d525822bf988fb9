/*
 * oracle/ref_driver.cpp — host driver around the REFERENCE's own kernel text.
 *
 * TEST INFRASTRUCTURE ONLY, and only in the build container: oracle/build_ref.py reads
 * the CUDA-C strings out of /root/reference (the generic kernels of
 * src/cuda_policy_iteration.py:180-286 / :577-694 / :1004-1126 and one runner's
 * `_dynamics_cuda_src()` text), writes them to a temporary directory and compiles THIS
 * file with them included, producing oracle/_ref/libref_<env>.so.  No reference text is
 * stored in the repository; the shared objects are git-ignored.
 *
 * Nothing below computes anything: the qualifiers CUDA adds to C++ are defined away so
 * the text compiles as host C++ (`__device__` -> static inline, `__global__` -> static,
 * blockIdx/blockDim/threadIdx -> plain structs the driver sets before each call), and
 * all arithmetic is the reference's own source plus glibc's libm.  The driver loops
 * "one thread per state" and exports the same C ABI as pi_oracle.cpp, so the Python
 * wrapper (oracle.OracleLib) can load either.
 *
 *   g++ -O2 -mfma -msse4.1 -ffp-contract=off -shared -fPIC -DPI_D=<D>
 *       -DREF_DYN_FILE="..." -DREF_GENERIC_FILE="..." ref_driver.cpp
 */
#include <cmath>
#include <cstdint>
#include <cstring>
#include <algorithm>

#ifndef PI_D
#error "build with -DPI_D=2, 4 or 6"
#endif
#define PI_C (1 << PI_D)

using std::min;
using std::max;
#define __device__ static inline
#define __global__ static
struct ref_dim3 { int x, y, z; };
static ref_dim3 blockIdx, blockDim, threadIdx;

#include REF_DYN_FILE
#include REF_GENERIC_FILE

#undef __device__
#undef __global__

namespace {

#if PI_D == 2
#define REF_EVAL policy_eval_kernel
#define REF_IMPROVE policy_improve_kernel
inline void ref_interp(const float* s, const float* lo, const float* hi, const int* g,
                       const int* st, int* idxs, float* wgts) {
    get_barycentric_2d(s[0], s[1], lo, hi, g, st, idxs, wgts);
}
inline void ref_dyn(const float* s, float a, float* ns, float* r, bool* t) {
    step_dynamics(s[0], s[1], a, &ns[0], &ns[1], r, t);
}
#elif PI_D == 4
#define REF_EVAL policy_eval_kernel_4d
#define REF_IMPROVE policy_improve_kernel_4d
inline void ref_interp(const float* s, const float* lo, const float* hi, const int* g,
                       const int* st, int* idxs, float* wgts) {
    get_barycentric_4d(s[0], s[1], s[2], s[3], lo, hi, g, st, idxs, wgts);
}
inline void ref_dyn(const float* s, float a, float* ns, float* r, bool* t) {
    step_dynamics(s[0], s[1], s[2], s[3], a, &ns[0], &ns[1], &ns[2], &ns[3], r, t);
}
#else
#define REF_EVAL policy_eval_kernel_6d
#define REF_IMPROVE policy_improve_kernel_6d
inline void ref_interp(const float* s, const float* lo, const float* hi, const int* g,
                       const int* st, int* idxs, float* wgts) {
    get_barycentric_6d(s[0], s[1], s[2], s[3], s[4], s[5], lo, hi, g, st, idxs, wgts);
}
inline void ref_dyn(const float* s, float a, float* ns, float* r, bool* t) {
    step_dynamics(s[0], s[1], s[2], s[3], s[4], s[5], a, &ns[0], &ns[1], &ns[2], &ns[3],
                  &ns[4], &ns[5], r, t);
}
#endif

inline void set_thread(int64_t s) {
    blockDim.x = 256;                 /* threads_per_block of the reference (:293) */
    blockIdx.x = (int)(s / 256);
    threadIdx.x = (int)(s % 256);
}

}  // namespace

extern "C" {

int oracle_dim(void) { return PI_D; }
int oracle_uses_libm(void) { return 1; }

void oracle_interp(int64_t m, const float* pts, const float* lo, const float* hi,
                   const int32_t* g, const int32_t* st, int32_t* idxs, float* wgts) {
    for (int64_t k = 0; k < m; ++k)
        ref_interp(pts + k * PI_D, lo, hi, g, st, idxs + k * PI_C, wgts + k * PI_C);
}

void oracle_step(int64_t m, const float* states, const float* act, float* next, float* reward,
                 uint8_t* term) {
    for (int64_t k = 0; k < m; ++k) {
        bool t;
        ref_dyn(states + k * PI_D, act[k], next + k * PI_D, &reward[k], &t);
        term[k] = t ? 1 : 0;
    }
}

float oracle_eval_sweep(const float* states, const float* actions, const int32_t* policy,
                        const float* V, float* newV, const uint8_t* is_term, const float* lo,
                        const float* hi, const int32_t* g, const int32_t* st, int64_t s0,
                        int64_t s1, float gamma) {
    const bool* term = reinterpret_cast<const bool*>(is_term);
    const int n_states = (int)s1;        /* threads >= n_states return at once */
    float delta = 0.0f;
    for (int64_t s = s0; s < s1; ++s) {
        set_thread(s);
        REF_EVAL(states, actions, policy, V, newV, term, lo, hi, g, st, n_states, gamma);
        float d = fabsf(newV[s] - V[s]);    /* max_abs_diff ReductionKernel (:164-172) */
        if (d > delta) delta = d;
    }
    return delta;
}

int64_t oracle_improve_sweep(const float* states, const float* actions, int32_t n_actions,
                             int32_t* policy, const float* V, const uint8_t* is_term,
                             const float* lo, const float* hi, const int32_t* g,
                             const int32_t* st, int64_t s0, int64_t s1, float gamma,
                             float* q_best, float* q_second) {
    (void)q_best;
    (void)q_second;                      /* the reference kernel does not expose Q values */
    const bool* term = reinterpret_cast<const bool*>(is_term);
    const int n_states = (int)s1;
    int64_t changed = 0;
    for (int64_t s = s0; s < s1; ++s) {
        int old = policy[s];
        set_thread(s);
        REF_IMPROVE(states, actions, policy, V, term, lo, hi, g, st, n_states, n_actions, gamma);
        if (policy[s] != old) ++changed;  /* all(policy == old) (:340, :354) */
    }
    return changed;
}

/* run() loop of the reference (:300-370) around its own kernels. */
void oracle_run(const float* states, const float* actions, int32_t n_actions, int32_t* policy,
                float* V, float* Vtmp, const uint8_t* is_term, const float* lo, const float* hi,
                const int32_t* g, const int32_t* st, int64_t n, float gamma, double theta,
                int32_t max_eval_iter, int32_t max_pi_iter, float* V_out, int64_t* stats,
                int32_t* sweeps_per_iter) {
    float* cur = V;
    float* nxt = Vtmp;
    int64_t total = 0;
    int32_t outer = 0;
    int stable = 0;
    for (int32_t it = 0; it < max_pi_iter; ++it) {
        int32_t sweeps = 0;
        for (int32_t i = 0; i < max_eval_iter; ++i) {
            float delta = oracle_eval_sweep(states, actions, policy, cur, nxt, is_term, lo, hi, g,
                                            st, 0, n, gamma);
            std::swap(cur, nxt);
            ++sweeps;
            if (i % 25 == 0 || i == max_eval_iter - 1)
                if ((double)delta < theta) break;      // :329 compares float(delta) with the config's Python float
        }
        total += sweeps;
        if (sweeps_per_iter) sweeps_per_iter[it] = sweeps;
        ++outer;
        int64_t changed = oracle_improve_sweep(states, actions, n_actions, policy, cur, is_term, lo,
                                               hi, g, st, 0, n, gamma, nullptr, nullptr);
        if (changed == 0) { stable = 1; break; }
    }
    std::memcpy(V_out, cur, sizeof(float) * (size_t)n);
    stats[0] = outer;
    stats[1] = total;
    stats[2] = stable;
}

}  // extern "C"
