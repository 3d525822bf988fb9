/*
 * pi_mi355.h — C ABI of libpi_mi355.so, the MI355X (gfx950) Bellman-backup engine.
 *
 * This is the drop-in boundary for the reference's device path: every entry point
 * replaces one piece of /root/reference/src/cuda_policy_iteration.py that the
 * reference reaches through cupy (citations are to that file unless noted).  The
 * reference has no FFI of its own — its "binding" is cupy.RawModule/ReductionKernel —
 * so the ABI below is what a ctypes stub in that file would bind instead
 * (INTEGRATION.md shows the stub).
 *
 * Conventions
 *   - plain C types only; device buffers are raw device pointers borrowed from the
 *     caller (the Python host uses torch-ROCm tensors' data_ptr()); `stream` is a
 *     hipStream_t passed as void* (NULL = the legacy default stream).
 *   - every int-returning call returns 0 on success, non-zero on failure;
 *     pi_last_error() then describes the failure (thread-local string).
 *   - one handle per (device, grid, action set); a handle is not thread-safe.
 *   - sweep calls never allocate, never synchronise, never copy to the host: they
 *     only enqueue kernels (and one memset for a counter) on `stream`.
 *   - flat state indices and V offsets are int32 on the device like the reference's
 *     (`int s_idx`, `int idxs[]`): n_states must be < 2^31.
 */
#ifndef PI_MI355_H_
#define PI_MI355_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pi_handle pi_handle;

/* ABI version of this header (bumped on any signature change). */
#define PI_MI355_ABI_VERSION 1
int pi_abi_version(void);

/* Last error message of the calling thread ("" if none). */
const char* pi_last_error(void);

/*
 * Create an engine for one grid + action set.
 * Replaces the device-array uploads of _allocate_tensors_and_compile (:145-150,
 * :545-550, :969-974): bounds, grid shape, strides and actions.  The (n, D) `d_states`
 * array is NOT uploaded — `bins[d]` (grid_shape[d] floats each; states_space[:, d]
 * takes exactly these values) replaces it.
 *   device      HIP device ordinal, or -1 for a host-only handle that can compile
 *               (pi_compile) but not launch — used by the CPU-side build check.
 *   D           2, 4 or 6          (CudaPolicyIteration2D/4D/6D)
 *   lo, hi      bounds_low/high, D floats (:96-97)
 *   actions     action_space, n_actions floats (:78)
 */
pi_handle* pi_create(int device, int D, const int32_t* grid_shape, const float* lo,
                     const float* hi, const float* const* bins, const float* actions,
                     int n_actions);
void pi_destroy(pi_handle* h);

/*
 * Compile the sweep kernels with the user's `step_dynamics` inlined.
 * Replaces _compile_cuda_module (:177-296, :576-704, :1000-1136): `source = user
 * string + generic kernels` -> NVRTC there, -> hipRTC for gfx950 here.  The string is
 * the reference plugin format unchanged (`__device__ void step_dynamics(...)` with
 * the arity for D; helper __device__ functions and #defines allowed).
 *   cache_dir   directory for compiled code objects (NULL = no cache).  A hit skips
 *               hipRTC entirely.
 *   log/log_len receives the compiler log (may be NULL/0).
 * On a host-only handle the code object is produced (and cached) but not loaded.
 */
int pi_compile(pi_handle* h, const char* dynamics_src, const char* cache_dir, char* log,
               size_t log_len);

/* The full translation unit pi_compile would build (for inspection / AOT builds).
 * Returns the length needed (excluding NUL); copies at most buf_len-1 bytes. */
size_t pi_kernel_source(pi_handle* h, const char* dynamics_src, char* buf, size_t buf_len);

/*
 * One Jacobi policy-evaluation sweep over states [s_begin, s_end):
 *   Vnew[s] = r(s, pi(s)) + gamma * sum_c w_c V[idx_c],  terminal states copy V[s].
 * Replaces the eval_kernel launch (:306-316, :714-724, :1146-1156) AND the
 * max_abs_diff reduction launched after it (:318-320): when d_delta != NULL it is
 * zeroed on `stream` and receives max|Vnew - V| over the range (float, device).
 * V is read over the whole grid; Vnew/policy/term only over the range.
 */
int pi_eval_sweep(pi_handle* h, const float* V, float* Vnew, const int32_t* policy,
                  const uint8_t* term, int64_t s_begin, int64_t s_end, float gamma,
                  float* d_delta, void* stream);

/*
 * Pick the number of workgroups per CU for the evaluation sweeps by timing them on the caller's
 * own V / policy (candidates 2..8; the gather-bound sweeps prefer few states in flight per XCD,
 * the arithmetic-bound ones prefer occupancy — it depends on the env and on the policy).
 * Vscratch is overwritten.  BLOCKS on the stream (event synchronise): call it outside any timed
 * or captured region.  No-op when PI_MI355_EVAL_BLOCKS_PER_CU pins the value or the range is small.
 */
int pi_autotune_eval(pi_handle* h, const float* V, float* Vscratch, const int32_t* policy,
                     const uint8_t* term, int64_t s_begin, int64_t s_end, float gamma, void* stream);

/*
 * n_sweeps evaluation sweeps ping-ponging between Va and Vb (sweep 0 reads Va and
 * writes Vb, sweep 1 reads Vb, ...) — the body of policy_evaluation's loop between
 * two host checks (:305-331, SYNC_INTERVAL = 25).  d_delta (nullable) receives the
 * residual of the LAST sweep only, which is the only one the reference looks at.
 * The newest iterate is in Vb when n_sweeps is odd, in Va when even.
 */
int pi_eval_sweeps(pi_handle* h, float* Va, float* Vb, const int32_t* policy,
                   const uint8_t* term, int64_t s_begin, int64_t s_end, float gamma,
                   int n_sweeps, float* d_delta, void* stream);

/*
 * Policy evaluation with TRANSITION RECORDS (MI355X-first; no counterpart in the reference,
 * same results bit for bit).  Between two policy improvements the policy is fixed, so the
 * transition of every state — reward, interpolation cell, fractional offsets — is the same in
 * every sweep of policy_evaluation's loop (:305-331).  With rebuild != 0 the first sweep runs
 * step_dynamics as usual and also writes (2 + D) * 4 bytes per state into `cache`; every
 * later sweep (and later calls with rebuild = 0) replays those records and only gathers V.
 *   cache, cache_bytes : caller-owned device workspace, 16-byte aligned, at least
 *                        pi_transition_cache_bytes(h, s_begin, s_end) bytes.
 *   rebuild            : must be non-zero on the first call after `policy`, `term`, the
 *                        range or the cache buffer changed (the library checks the last two).
 * Ping-pong and d_delta semantics as pi_eval_sweeps.  Va / Vb must be 16-byte aligned.
 */
size_t pi_transition_cache_bytes(pi_handle* h, int64_t s_begin, int64_t s_end);
int pi_eval_sweeps_cached(pi_handle* h, float* Va, float* Vb, const int32_t* policy,
                          const uint8_t* term, int64_t s_begin, int64_t s_end, float gamma,
                          int n_sweeps, int rebuild, void* cache, size_t cache_bytes,
                          float* d_delta, void* stream);

/*
 * Greedy improvement over [s_begin, s_end): policy[s] = argmax_a Q(s, a), first
 * maximum wins, terminal states untouched.  Replaces the improve_kernel launch
 * (:342-352, :750-760, :1182-1192) AND `old = policy.copy(); all(policy == old)`
 * (:340, :354): when d_changed != NULL it is zeroed on `stream` and receives the
 * number of entries that changed (uint32, device); stable <=> 0.
 */
int pi_improve_sweep(pi_handle* h, const float* V, int32_t* policy, const uint8_t* term,
                     int64_t s_begin, int64_t s_end, float gamma, uint32_t* d_changed,
                     void* stream);

/*
 * Value-iteration sweep over [s_begin, s_end): Vnew[s] = max_a Q(s, a) and policy[s] = argmax
 * in one pass (terminal states copy V and keep their policy entry).  This is the fused form the
 * reference's README sketches (README.md:790-799, "V_new <- best_Q") but does not implement —
 * its code alternates policy_eval_kernel / policy_improve_kernel.  d_delta (nullable, zeroed
 * here) receives max|Vnew - V|, d_changed (nullable, zeroed here) the number of policy entries
 * that changed.
 */
int pi_value_sweep(pi_handle* h, const float* V, float* Vnew, int32_t* policy, const uint8_t* term,
                   int64_t s_begin, int64_t s_end, float gamma, float* d_delta, uint32_t* d_changed,
                   void* stream);

/*
 * Which dimension-0 planes of V can the states of [s_begin, s_end) read, under ANY action?
 * d_bitmap (device, ceil(grid_shape[0] / 32) uint32 words, zeroed here) receives one bit per
 * plane: the plane of every successor cell and the plane above it.  No counterpart in the
 * reference (it has no multi-GPU path); the multi-GPU host uses it to exchange only reachable
 * planes between ranks instead of all-gathering V after every sweep.
 */
int pi_reach_planes(pi_handle* h, const uint8_t* term, int64_t s_begin, int64_t s_end,
                    uint32_t* d_bitmap, void* stream);

/*
 * Probes used by the parity tests (not on the hot path): run the compiled plugin /
 * interpolation on m arbitrary points.  All pointers are device pointers.
 *   pi_probe_step   : states (m,D), acts (m) -> next (m,D), reward (m), done (m)
 *   pi_probe_interp : pts (m,D) -> idxs (m,2^D) int32, wgts (m,2^D) float
 *                     (get_barycentric_{2,4,6}d, reference corner order)
 */
int pi_probe_step(pi_handle* h, const float* states, const float* acts, float* next,
                  float* reward, uint8_t* done, int64_t m, void* stream);
int pi_probe_interp(pi_handle* h, const float* pts, int32_t* idxs, float* wgts, int64_t m,
                    void* stream);

/* Introspection: 0 n_states, 1 n_actions, 2 D, 3 workgroups per launch,
 * 4 VGPRs of the eval kernel, 5 VGPRs of the improve kernel, 6 compute units,
 * 7 = 1 if the last pi_compile was served from the cache, 8 VGPRs of the replay kernel,
 * 9 states per thread of the replay kernel, 10 tiled kernels loaded, 12 / 13 workgroups per CU
 * of the evaluation / improvement sweeps, 20+d box extent, 30+d tile extent, 40+d reach. */
int64_t pi_info(pi_handle* h, int what);

#ifdef __cplusplus
}
#endif
#endif /* PI_MI355_H_ */
