/*
 * oracle/pi_oracle.cpp — CPU restatement of the reference's Bellman-backup path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under dynamicprogramming_amd/, src/, runners/
 * or utils/ may import, link or execute this file.  Allowed users: tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg — as the checker,
 * never as the thing shipped.
 *
 * What it restates (all citations are /root/reference/src/cuda_policy_iteration.py):
 *   oracle_interp        get_barycentric_2d  :183-210, _4d :580-614, _6d :1007-1042
 *   oracle_eval_sweep    policy_eval_kernel  :212-242, _4d :616-649, _6d :1044-1079
 *                        + max_abs_diff ReductionKernel :164-172
 *   oracle_improve_sweep policy_improve_kernel :244-283, _4d :651-691, _6d :1081-1123
 *                        + the `all(policy == old)` test :340,:354
 *   oracle_run           policy_evaluation :300-336, policy_improvement :338-355,
 *                        run :357-370  (delta looked at on sweeps 0,25,50,.. and the last)
 *   oracle_step          thin batch driver over the plugged `step_dynamics`.
 *   oracle_eval_points / oracle_improve_points   the same backups for a list of states (coordinates).
 *
 * Built once per (D, dynamics string) by oracle/__init__.py:
 *   g++ -O2 -mfma -msse4.1 -ffp-contract=off -fopenmp -shared -fPIC
 *       -DPI_D=<2|4|6> -DPI_DYN_FILE="<file holding the user's step_dynamics text>"
 *       [-DPI_ORACLE_LIBM]
 * Two arithmetic modes:
 *   default          sinf/cosf/fmodf -> include/pi_math.h (the product's definition;
 *                    HIP kernel == this oracle bit for bit)
 *   PI_ORACLE_LIBM   glibc sinf/cosf/fmodf (what the reference's own kernel text
 *                    computes when built with g++; used to pin this restatement
 *                    against oracle/_ref and the reference's committed .npz results)
 * No floating-point contraction anywhere; explicit fmaf only where the reference
 * has fmaf (the expected-value chain).
 */
#include <cmath>
#include <cstdint>
#include <cstring>
#include <algorithm>
#ifdef _OPENMP
#include <omp.h>
#endif

#ifndef PI_D
#error "build with -DPI_D=2, 4 or 6"
#endif
#define PI_C (1 << PI_D)

/* ---- the env plugin: CUDA-C text compiled as host C++ ------------------------- */
#define __device__ static inline
#define __host__
#define __forceinline__
using std::min;
using std::max;
#ifndef PI_ORACLE_LIBM
#include "pi_math.h"
#define sinf pi_sinf
#define cosf pi_cosf
#define fmodf pi_fmodf
#endif
#include PI_DYN_FILE
#undef __device__

namespace {

/* Call the plugin with the arity the reference documents for each D
 * (:11-15, :456-460, :869-874). */
inline void dyn(const float* s, float a, float* ns, float* r, bool* t) {
#if PI_D == 2
    step_dynamics(s[0], s[1], a, &ns[0], &ns[1], r, t);
#elif PI_D == 4
    step_dynamics(s[0], s[1], s[2], s[3], a, &ns[0], &ns[1], &ns[2], &ns[3], r, t);
#elif PI_D == 6
    step_dynamics(s[0], s[1], s[2], s[3], s[4], s[5], a,
                  &ns[0], &ns[1], &ns[2], &ns[3], &ns[4], &ns[5], r, t);
#else
#error "PI_D must be 2, 4 or 6"
#endif
}

/* Corner c selects i[d] or i[d]+1 per dimension.  2D is written out in the
 * reference with dim 1 toggling fastest (:201-209); 4D/6D use bit d of c for
 * dimension d, dim 0 = LSB (:607, :1035). */
inline int corner_bit(int c, int d) {
#if PI_D == 2
    return (c >> (1 - d)) & 1;
#else
    return (c >> d) & 1;
#endif
}

inline void interp(const float* s, const float* lo, const float* hi,
                   const int32_t* g, const int32_t* st, int32_t* idxs, float* wgts) {
    int i[PI_D];
    float frac[PI_D];
    for (int d = 0; d < PI_D; ++d) {
        float n = (s[d] - lo[d]) / (hi[d] - lo[d]) * (float)(g[d] - 1);
        n = fmaxf(0.0f, fminf(n, (float)(g[d] - 1)));
        i[d] = min((int)n, g[d] - 2);
        frac[d] = n - (float)i[d];
    }
    for (int c = 0; c < PI_C; ++c) {
        int idx = 0;
        float w = 1.0f;
        for (int d = 0; d < PI_D; ++d) {
            int bit = corner_bit(c, d);
            idx += (i[d] + bit) * st[d];
            w *= bit ? frac[d] : (1.0f - frac[d]);
        }
        idxs[c] = idx;
        wgts[c] = w;
    }
}

/* One state-action backup: r + gamma * sum_c w_c V[idx_c]  (fmaf chain, c ascending). */
inline float backup(const float* s, float a, const float* V, const float* lo, const float* hi,
                    const int32_t* g, const int32_t* st, float gamma) {
    float ns[PI_D], reward;
    bool term;
    dyn(s, a, ns, &reward, &term);
    float e = 0.0f;
    if (!term) {
        int32_t idxs[PI_C];
        float wgts[PI_C];
        interp(ns, lo, hi, g, st, idxs, wgts);
        for (int c = 0; c < PI_C; ++c) e = fmaf(wgts[c], V[idxs[c]], e);
    }
    return reward + gamma * e;
}

}  // namespace

extern "C" {

int oracle_dim(void) { return PI_D; }
void oracle_set_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}
int oracle_uses_libm(void) {
#ifdef PI_ORACLE_LIBM
    return 1;
#else
    return 0;
#endif
}

void oracle_interp(int64_t m, const float* pts, const float* lo, const float* hi,
                   const int32_t* g, const int32_t* st, int32_t* idxs, float* wgts) {
    for (int64_t k = 0; k < m; ++k)
        interp(pts + k * PI_D, lo, hi, g, st, idxs + k * PI_C, wgts + k * PI_C);
}

void oracle_step(int64_t m, const float* states, const float* act, float* next,
                 float* reward, uint8_t* term) {
    for (int64_t k = 0; k < m; ++k) {
        bool t;
        dyn(states + k * PI_D, act[k], next + k * PI_D, &reward[k], &t);
        term[k] = t ? 1 : 0;
    }
}

/* Jacobi evaluation sweep over states [s0, s1).  Returns max|newV - V| over that
 * range (the reference's separate max_abs_diff pass; identity 0, NaNs never win). */
float oracle_eval_sweep(const float* states, const float* actions, const int32_t* policy,
                        const float* V, float* newV, const uint8_t* is_term,
                        const float* lo, const float* hi, const int32_t* g, const int32_t* st,
                        int64_t s0, int64_t s1, float gamma) {
    float delta = 0.0f;
#pragma omp parallel for schedule(static) reduction(max : delta)
    for (int64_t s = s0; s < s1; ++s) {
        float nv;
        if (is_term[s]) nv = V[s];
        else nv = backup(states + s * PI_D, actions[policy[s]], V, lo, hi, g, st, gamma);
        newV[s] = nv;
        float d = fabsf(nv - V[s]);
        if (d > delta) delta = d;
    }
    return delta;
}

/* Greedy improvement over [s0, s1).  Strict '>' from -1e30: lowest index wins ties,
 * NaN never wins.  Terminal states keep their policy entry.  Returns how many
 * entries changed; q_best/q_second (optional) receive the top-2 action values so
 * tests can mask tie-fragile states. */
int64_t oracle_improve_sweep(const float* states, const float* actions, int32_t n_actions,
                             int32_t* policy, const float* V, const uint8_t* is_term,
                             const float* lo, const float* hi, const int32_t* g,
                             const int32_t* st, int64_t s0, int64_t s1, float gamma,
                             float* q_best, float* q_second) {
    int64_t changed = 0;
#pragma omp parallel for schedule(static) reduction(+ : changed)
    for (int64_t s = s0; s < s1; ++s) {
        if (is_term[s]) {
            if (q_best) q_best[s] = 0.0f;
            if (q_second) q_second[s] = -INFINITY;
            continue;
        }
        float max_q = -1.0e30f, second = -INFINITY;
        int best = 0;
        for (int a = 0; a < n_actions; ++a) {
            float q = backup(states + s * PI_D, actions[a], V, lo, hi, g, st, gamma);
            if (q > max_q) { second = max_q; max_q = q; best = a; }
            else if (q > second) second = q;
        }
        if (policy[s] != best) ++changed;
        policy[s] = best;
        if (q_best) q_best[s] = max_q;
        if (q_second) q_second[s] = second;
    }
    return changed;
}

/* Value-iteration sweep (the fused form of the reference's README :790-799): newV = max_a Q,
 * policy = argmax (strict '>' from -1e30), terminal states copy V.  Returns max|newV - V|;
 * *changed_out receives the number of policy entries that changed. */
float oracle_value_sweep(const float* states, const float* actions, int32_t n_actions,
                         int32_t* policy, const float* V, float* newV, const uint8_t* is_term,
                         const float* lo, const float* hi, const int32_t* g, const int32_t* st,
                         int64_t s0, int64_t s1, float gamma, int64_t* changed_out) {
    float delta = 0.0f;
    int64_t changed = 0;
#pragma omp parallel for schedule(static) reduction(max : delta) reduction(+ : changed)
    for (int64_t s = s0; s < s1; ++s) {
        if (is_term[s]) { newV[s] = V[s]; continue; }
        float max_q = -1.0e30f;
        int best = 0;
        for (int a = 0; a < n_actions; ++a) {
            float q = backup(states + s * PI_D, actions[a], V, lo, hi, g, st, gamma);
            if (q > max_q) { max_q = q; best = a; }
        }
        if (policy[s] != best) ++changed;
        policy[s] = best;
        newV[s] = max_q;
        float d = fabsf(max_q - V[s]);
        if (d > delta) delta = d;
    }
    if (changed_out) *changed_out = changed;
    return delta;
}

/* The same backups for a LIST of states given by their coordinates instead of a range of the
 * reference's flat order: tests of grids that the product holds in another memory order check windows of
 * MEMORY-order indices, which are scattered in the reference's order.  V is the whole table in the
 * reference's order.  eval: out[k] = backup(coords[k], action_values[k]) (:616-649 for one non-terminal
 * state; the caller handles terminal states).  improve: best[k] / best_q[k] of the strict-'>' argmax
 * from -1e30 (:651-691). */
void oracle_eval_points(int64_t m, const float* coords, const float* action_values, const float* V,
                        const float* lo, const float* hi, const int32_t* g, const int32_t* st,
                        float gamma, float* out) {
#pragma omp parallel for schedule(static)
    for (int64_t k = 0; k < m; ++k)
        out[k] = backup(coords + k * PI_D, action_values[k], V, lo, hi, g, st, gamma);
}

void oracle_improve_points(int64_t m, const float* coords, const float* actions, int32_t n_actions,
                           const float* V, const float* lo, const float* hi, const int32_t* g,
                           const int32_t* st, float gamma, int32_t* best, float* best_q) {
#pragma omp parallel for schedule(static)
    for (int64_t k = 0; k < m; ++k) {
        float max_q = -1.0e30f;
        int b = 0;
        for (int a = 0; a < n_actions; ++a) {
            float q = backup(coords + k * PI_D, actions[a], V, lo, hi, g, st, gamma);
            if (q > max_q) { max_q = q; b = a; }
        }
        best[k] = b;
        if (best_q) best_q[k] = max_q;
    }
}

/*
 * The whole run() loop.  V and Vtmp are the two Jacobi buffers (both seeded by the
 * caller exactly as _allocate_tensors_and_compile :152-161 does); on return `V_out`
 * receives the final iterate.  stats[0]=outer iterations done, stats[1]=total eval
 * sweeps, stats[2]=1 if the policy became stable; sweeps_per_iter (len max_pi_iter)
 * gets the eval sweeps of each outer iteration.
 */
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
