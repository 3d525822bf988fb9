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
#define PI_MI355_ABI_VERSION 10
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
 * term == NULL means "this grid has no terminal states" (the caller's promise; every sweep, reach and
 * sharded entry point accepts it): the kernels then stream no mask, and a sweep that is not asked for a
 * residual does not read the states' old values either — 4 B per state instead of 9 besides the gather.
 */
int pi_eval_sweep(pi_handle* h, const float* V, float* Vnew, const int32_t* policy,
                  const uint8_t* term, int64_t s_begin, int64_t s_end, float gamma,
                  float* d_delta, void* stream);

/*
 * n_sweeps evaluation sweeps ping-ponging between Va and Vb (sweep 0 reads Va and
 * writes Vb, sweep 1 reads Vb, ...) — the body of policy_evaluation's loop between
 * two host checks (:305-331, SYNC_INTERVAL = 25).  d_delta (nullable) receives the
 * residual of the LAST sweep only, which is the only one the reference looks at.
 * The newest iterate is in Vb when n_sweeps is odd, in Va when even.
 * Terminal states keep their value: sweep 0 copies it from Va into Vb, after which both buffers hold it
 * and the later sweeps of the batch neither store it again nor stream the old values for it (without a
 * residual request they read 5 B per state besides the gather instead of 9, and write only non-terminal
 * states) — same values in both buffers as a copy on every sweep gives.
 * Ranges of up to 2^20 states are launch-bound (a few microseconds per sweep): there the whole
 * batch, including the residual fold, is replayed as ONE hipGraph (built on first use per
 * argument set, cached in the handle) instead of n_sweeps host launches.  Grids of up to 12 288
 * states (4 096 in 4-D, 1 024 in 6-D) swept whole go further: ONE workgroup keeps the value table in LDS and
 * each state's successor cell, weights and reward in registers and runs the entire batch in one
 * launch (pi_eval_resident_kernel); the two newest iterates land in Va / Vb as above.
 */
int pi_eval_sweeps(pi_handle* h, float* Va, float* Vb, const int32_t* policy,
                   const uint8_t* term, int64_t s_begin, int64_t s_end, float gamma,
                   int n_sweeps, float* d_delta, void* stream);

/*
 * Optional, once per terminal mask (the reference computes its mask once, at allocation: :150-160): lets the
 * later sweeps of pi_eval_sweeps / pi_eval_sweeps_sharded batches and pi_improve_sweep(_sharded) launches (an
 * improvement sweep never touches terminal states, :253) visit only the NON-terminal states of their range — for the
 * sharded driver: of every launch range of the rank — through a list of
 * their indices the library builds here (one device-to-host copy of the mask, one upload of the list; blocks
 * on `stream`).  Worth it where terminal regions cut through many waves — double cartpole 25^6: 35 % of the
 * states are terminal and 16 % of the waves are partly idle — and skipped where it is not: the list is kept
 * only when at least 3 % of the grid's lane slots would be idle otherwise and the grid has 2^20 states or more
 * (pi_info 16 = length of the list in use, 0 = none).  Results are identical with and without it.
 * Contract: the bytes behind d_term must not change while the list is in use; call again after changing
 * them, or with d_term == NULL to drop the list.  Calls given another mask pointer, or a state range that is not
 * inside the listed one, ignore the list.  pi_prepare_mask lists the whole grid (4 B per live state on the device,
 * n / 8 + n / 8 bytes of index on the host); pi_prepare_mask_range lists [s_begin, s_end) only — what a rank of a
 * sharded run needs (its launches never leave its shard): 1 / world of the list, of the index and of the host pass.
 */
int pi_prepare_mask(pi_handle* h, const uint8_t* d_term, void* stream);
int pi_prepare_mask_range(pi_handle* h, const uint8_t* d_term, int64_t s_begin, int64_t s_end, void* stream);
/* The list pi_prepare_mask built (ascending flat indices of the non-terminal states of the listed range), copied into
 * d_out (device, `capacity` entries) on `stream`; returns its length (0: no list is in use; d_out == NULL: length only;
 * -1: error).  For inspection and tests: the sweeps use the list inside the library. */
int64_t pi_live_list(pi_handle* h, int32_t* d_out, int64_t capacity, void* stream);

/*
 * Optional bracket around one policy_evaluation (:300-336), for handles with a live-state list: under a FIXED
 * policy a live state whose successor is terminal has V'(s) = reward + gamma * 0 in every sweep, so once both
 * Jacobi buffers hold that value it need not be visited again.  pi_eval_begin filters the live list down to the
 * states that bootstrap under `policy` (one launch, one 8-byte read-back; blocks on `stream`; the shorter list is
 * kept when it saves at least 3 %), and until pi_eval_end whole-grid pi_eval_sweeps batches with this policy
 * pointer use it for every sweep whose source AND destination buffer have been written by a full sweep since
 * pi_eval_begin (the library tracks the buffers; with the reference's ping-pong that is every sweep after the
 * evaluation's second).  Results are identical with and without the bracket.  pi_info 17 = entries in use.
 * Contract: between begin and end neither the policy array nor the value buffers are written by anyone but
 * the evaluation sweeps; pi_improve_sweep / pi_value_sweep end the bracket by themselves.
 */
int pi_eval_begin(pi_handle* h, const int32_t* policy, const uint8_t* term, void* stream);
int pi_eval_end(pi_handle* h);

/*
 * The whole policy_evaluation loop (:300-336) in ONE launch, for grids the LDS-resident kernel
 * holds (pi_info 13 > 0: up to 12 288 states in 2-D, 4 096 in 4-D, 1 024 in 6-D) and, beyond those, for
 * launch-bound grids of up to 2^17 states in 2-D / 4-D (pi_info 19 > 0 = workgroups of the dataflow kernel:
 * the iterates travel between workgroups as tagged 8-byte granules, no grid barrier between sweeps; BASELINE
 * config C2, pendulum 200 x 200, is one); fails on every other grid: up to max_sweeps Jacobi sweeps of the whole grid under
 * `policy`, the residual looked at on sweeps 0, check_interval, 2 check_interval, ... (the
 * reference's SYNC_INTERVAL = 25) and on the last one, stopping at the first residual below theta.
 * V is updated in place (the newest iterate); *d_sweeps receives the number of sweeps done,
 * *d_delta (nullable) the last residual looked at, d_residual_log[k] every residual looked at
 * (k-th look; at least max_sweeps / check_interval + 2 floats).  Same arithmetic, same sweep
 * count and same V as the same loop driven from the host through pi_eval_sweeps.
 * On 2-D grids of ~4 000 to 2^16 states among those (pi_info 30 > 0) the evaluation first runs on the CUs of ONE XCD, with
 * the hand-off through that XCD's L2 (pi_xcd_kernel; placement checked at run time; the call then synchronises `stream`
 * once); when that launch cannot go through V is untouched and the LDS-resident or the dataflow kernel runs the evaluation.
 * Dataflow kernel: every device-side wait is bounded (PI_MI355_FLOW_TIMEOUT seconds, default 2; the kernel needs all its
 * workgroups resident at once, which another stream or process holding CUs can prevent); when a workgroup gives up,
 * *d_sweeps = -1 and V holds what it held before the call (a finish kernel is the only writer of V) — the caller runs the
 * same evaluation through pi_eval_sweeps.
 */
int pi_policy_evaluation(pi_handle* h, float* V, const int32_t* policy, const uint8_t* term, float gamma,
                         double theta, int max_sweeps, int check_interval, int32_t* d_sweeps, float* d_delta,
                         float* d_residual_log, void* stream);

/*
 * The reference's whole run() (:357-370) — policy_evaluation (:300-336), policy_improvement (:338-355), until no entry
 * of the policy changes or max_pi_iter rounds are done — in ONE launch (pi_info 34 > 0), for grids the LDS-resident
 * kernel holds (pi_info 13 > 0; BASELINE config C1, pendulum 50 x 50, is one: V and the policy stay in one CU's LDS) and
 * for 2-D grids of ~4 000 to 2^16 states (pi_info 30 > 0; BASELINE config C2, pendulum 200 x 200, is one: the
 * kernel runs on the CUs of one XCD, the iterates travel through that XCD's L2 as tagged granules, a thread keeps the
 * actions of its states in registers).  V and policy are updated in place (the last iterate, the last policy); each
 * evaluation does up to max_eval_sweeps sweeps with the residual looked at every check_interval sweeps exactly as
 * pi_policy_evaluation does.
 * d_result[0] = rounds done (>= 1), d_result[1] = 1 when the policy is stable; d_iter_log[4 r ..] = {sweeps of round r's
 * evaluation, bits of its last residual, policy entries changed by its improvement, 0} (4 * max_pi_iter words).
 * Same arithmetic, sweep counts, V and policy as the same loop driven through pi_policy_evaluation / pi_improve_sweep.
 * XCD-local kernel: every device-side wait is bounded and workgroup placement is checked, not assumed: when the launch
 * could not go through, d_result[0] < 0 and V and policy hold what they held before the call — the caller runs the loop
 * itself.  Asynchronous on `stream`.  pi_info 33 = whole runs launched.
 */
int pi_policy_iteration(pi_handle* h, float* V, int32_t* policy, const uint8_t* term, float gamma, double theta,
                        int max_eval_sweeps, int check_interval, int max_pi_iter, int32_t* d_result, uint32_t* d_iter_log,
                        void* stream);

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
 * Which planes of V can the states of [s_begin, s_end) read, under ANY action?  Along dimension
 * `dim`: d_bitmap (device, ceil(grid_shape[dim] / 32) uint32 words, zeroed here) receives one bit
 * per index: that of every successor cell and the one above it.  No counterpart in the reference
 * (it has no multi-GPU path); the multi-GPU plan uses dim = 0 to exchange only reachable planes
 * between ranks instead of all-gathering V after every sweep; the other
 * dimensions tell how wide a halo a shard along them would need.
 */
int pi_reach_planes(pi_handle* h, const uint8_t* term, int64_t s_begin, int64_t s_end, int dim,
                    uint32_t* d_bitmap, void* stream);

/*
 * The same question at a finer grain.  "Units" are the leading `depth` dimensions flattened: depth 1
 * = the planes of dimension 0, depth 2 = the rows (i0, i1), each a contiguous run of
 * n_states / (grid_shape[0] * grid_shape[1]) states.  d_bitmap (device, ceil(units / 32) words,
 * zeroed here) receives one bit per unit some state of [s_begin, s_end) can read under ANY action.
 * Every env moves a position by dt * velocity, so the rows a shard reaches in a neighbouring plane
 * are the ones with the right sign and size of velocity: 2-4x fewer values than whole planes.
 * pi_reach_depth_max: 2 for grids of three or more dimensions with up to 2^17 such rows, else 1.
 */
int pi_reach_depth_max(pi_handle* h);
int pi_reach_units(pi_handle* h, const uint8_t* term, int64_t s_begin, int64_t s_end, int depth,
                   uint32_t* d_bitmap, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Multi-GPU (one process per GPU, RCCL over xGMI).  SURVEY.md section 8b/8e: the reference is a
 * single-device loop (:300-336), so these entry points have no line to replace — they are what a
 * maintainer binds to run the same loop over several GPUs.  Partition: rank r owns the contiguous
 * state range [r * per, min((r + 1) * per, n)), per = ceil(n / world); every rank keeps full-size
 * V buffers of per * world floats.
 * ------------------------------------------------------------------------------------------- */

/* 128-byte RCCL id: created once (any rank), handed to every rank out of band. */
int pi_comm_unique_id(void* id128);
/* Join the RCCL communicator on the handle's device (collective: every rank calls it). */
int pi_comm_init(pi_handle* h, int rank, int world, const void* id128);
/* In-process transport for tests: `world` handles of ONE process, one host thread each, exchange
 * through device-to-device copies ordered by HIP events — same stream semantics as RCCL. */
int pi_comm_init_local(pi_handle* h, int rank, int world, const char* group_name);
/*
 * Peer-to-peer transport (csrc/pi_p2p.cpp): no RCCL on the data path.  The sending GPU STORES its halo rows straight
 * into the receiving rank's buffer (mapped with hipIpcOpenMemHandle; the stores travel over xGMI) and the two sides
 * hand-shake through counters in a small uncached flag page per rank — the rendezvous of an ncclSend / ncclRecv pair
 * without the ~30 us group latency, which at C4 @ 8 is of the order of a rank's whole sweep; the scalar reductions go
 * through the same pages.  One process per rank.  Everything above the transport (pi_exchange_plan,
 * pi_eval_sweeps_sharded, the collectives below) is unchanged.  No reference counterpart
 * (src/cuda_policy_iteration.py:300-336 is a single-device loop).
 *   pi_p2p_describe   registers up to 4 device buffers of this rank — the ones sends and receives will name: both V
 *                     buffers and the policy; same sizes on every rank, addresses are symmetric (offset o of buffer b
 *                     goes to offset o of the peer's buffer b) — allocates the rank's flag page and fills a 512-byte
 *                     descriptor (process id, device, IPC handles).  The caller all-gathers the descriptors of all ranks
 *                     out of band, as it hands the RCCL id around (MPI, files, a socket: 512 bytes per rank);
 *   pi_comm_init_p2p  descs = world x 512 bytes ordered by rank: maps the peers' buffers and pages, builds the transport's
 *                     kernels (hipRTC, cache_dir as pi_compile) and installs it on the handle.  Collective in the sense
 *                     that every rank must call it before any rank exchanges.
 * Every device-side wait is bounded by PI_MI355_COMM_TIMEOUT seconds (default 120); a peer that never arrives is
 * reported by the next reduction or all-gather (which therefore block on their stream), never a hung wave.
 * pi_p2p_compile_check: the transport's device code builds for gfx950 (needs no GPU).
 */
int pi_p2p_describe(pi_handle* h, int rank, int world, const void* const* bufs, const int64_t* bytes, int n_bufs,
                    void* desc512);
int pi_comm_init_p2p(pi_handle* h, int rank, int world, const void* descs, const char* cache_dir);
int pi_p2p_compile_check(const char* cache_dir);
int pi_comm_destroy(pi_handle* h);
/* Fused exchange: when the plan is row-exact (pi_comm_info(h, 5)) and the transport is the peer-to-peer one, the
 * swept-first launch of every sharded evaluation sweep is pi_eval_push_kernel (csrc/pi_push_kernels.hip): the lane that
 * stores V'(s) stores it into the peers that read the row as well, one-wave kernels hand-shake in front of it and behind
 * it, everything on the caller's stream — no copy kernel, no second stream (pi_comm_info(h, 6) == 1;
 * PI_MI355_P2P_FUSED=0 keeps the copy kernel); on grids with terminal states whose shard keeps a live-state list
 * (pi_prepare_mask_range) the later sweeps of every batch are fused the same way.  The kernel lives in a second module of the handle that is built on
 * demand from the same translation unit (pi_set_option 5 builds it ahead of time). */
/* 0 rank, 1 world, 2 transport (1 RCCL, 2 in-process, 3 peer-to-peer), 3 plan (0 none, 1 all-gather, 2 halo),
 * 4 granularity of the plan's reach probe (1 planes of dimension 0, 2 rows (i0, i1)), 5 row-exact plan (0 | 1),
 * 6 fused exchange (0 | 1), 7 destination masks cut down to the pairs (i_0, i_v) each peer reads (0 | 1; any memory order
 * with dimension 0 slowest), 8 values one fused sweep delivers, counted per receiver (-1: not a fused plan), 9 / 10 entries of
 * the swept-first / interior state lists of a row-exact or state-exact plan (-1: the plan sweeps ranges). */
int pi_comm_info(pi_handle* h, int what);

/* In-place collectives on the caller's stream: shard r of the buffer lives at r * shard_elems. */
int pi_allgather_V(pi_handle* h, float* V_full, int64_t shard_elems, void* stream);
int pi_allgather_policy(pi_handle* h, int32_t* policy_full, int64_t shard_elems, void* stream);
int pi_allreduce_max_f32(pi_handle* h, float* d_value, void* stream);
int pi_allreduce_sum_u32(pi_handle* h, uint32_t* d_value, void* stream);

/*
 * Pure host logic (needs no GPU and no communicator): which pieces of V' travel after a sweep.
 *   reach[r * g0 + p] != 0  <=>  rank r's shard can read unit p, the g0 units being consecutive
 *   runs of stride0 states (planes of dimension 0, or rows (i0, i1): pi_reach_units)
 * Writes up to `cap` segments {src, dst, a, b} ("src sends V'[a, b) to dst") and returns how many
 * there are (-1 on bad arguments).  Every rank derives the same list from the same bitmaps.
 */
int64_t pi_plan_segments(int world, int64_t g0, int64_t stride0, int64_t n_states, int64_t per,
                         const uint8_t* reach, int64_t* segs, int64_t cap);

/*
 * Collective: measure this shard's reach (pi_reach_units at pi_reach_depth_max), all-gather the
 * bitmaps, derive the segments and choose the exchange.  mode 0 = choose (halo unless some rank would receive more
 * than 60 % of an all-gather), 1 = all-gather, 2 = halo; overlap != 0 sweeps the planes peers
 * wait for first and sends them on a second stream while the interior is swept.
 * info (nullable, 5 values): mode chosen (1 | 2), elements received / sent per sweep by this rank,
 * number of send ranges, number of interior ranges.  Blocks on `stream` (one-off).
 */
int pi_exchange_plan(pi_handle* h, const uint8_t* term, int64_t per, int mode, int overlap,
                     int64_t* info, void* stream);
/* The launch ranges of the current plan: up to cap triples {kind, begin, end}, kind 0 = swept first
 * (peers wait for rows inside it), 1 = interior (swept while the halo travels).  Returns how many
 * there are (-1 without a plan).  A plan without overlap reports the shard as one kind-0 range; a plan whose swept-first
 * set is a list of single states (pi_comm_info(h, 7): the fused exchange in a memory order whose rows are all reachable)
 * reports the coarse row ranges it was cut from. */
int64_t pi_plan_ranges(pi_handle* h, int64_t* ranges, int64_t cap);
/*
 * One part of a sharded evaluation sweep WITHOUT the exchange, launched exactly as pi_eval_sweeps_sharded launches it:
 * part 0 = what the peers wait for (swept first), part 1 = the interior; both together are one sweep of the shard.
 * On grids without terminal states the plan may be ROW-EXACT (pi_comm_info(h, 5) == 1; PI_MI355_ROW_EXACT=0 / 1
 * forces it): part 0 is then the list of exactly the rows that travel, swept in one launch of the list kernel, and
 * part 1 the list of all other states of the shard, instead of a few contiguous ranges that also hold rows nobody
 * waits for.  No reference counterpart (src/cuda_policy_iteration.py:300-336 is a single-device loop); for measurements
 * and tests.
 */
int pi_eval_sweep_part(pi_handle* h, const float* V, float* Vnew, const int32_t* policy, const uint8_t* term, int part,
                       float gamma, void* stream);
/* Make this rank's freshly written shard of V_full visible where the other ranks read it. */
int pi_exchange_V(pi_handle* h, float* V_full, void* stream);
/*
 * pi_eval_sweeps over this rank's shard with the exchange after every sweep — the 25-sweep batch
 * between two host checks (:305-331) without returning to the host.  d_delta (nullable) receives
 * the residual of the last sweep, already MAX-reduced over all ranks.
 */
int pi_eval_sweeps_sharded(pi_handle* h, float* Va, float* Vb, const int32_t* policy,
                           const uint8_t* term, float gamma, int n_sweeps, float* d_delta,
                           void* stream);
/* pi_improve_sweep over this rank's shard; d_changed (nullable) is SUM-reduced over all ranks. */
int pi_improve_sweep_sharded(pi_handle* h, const float* V, int32_t* policy, const uint8_t* term,
                             float gamma, uint32_t* d_changed, void* stream);

/*
 * Probes used by the parity tests (not on the hot path): run the compiled plugin /
 * interpolation on m arbitrary points.  All pointers are device pointers.
 *   pi_probe_step   : states (m,D), acts (m) -> next (m,D), reward (m), done (m)
 *   pi_probe_interp : pts (m,D) -> idxs (m,2^D) int32, wgts (m,2^D) float
 *                     (get_barycentric_{2,4,6}d, reference corner order)
 *   pi_probe_coords : out[(s - s_begin) * D + d] = coordinate d of grid node s, computed the way
 *                     the sweeps compute it (flat index -> per-dimension indices -> LDS bin tables), with
 *                     `chunks_per_workgroup` chunks per workgroup — states_space[s] of :84-87
 */
int pi_probe_step(pi_handle* h, const float* states, const float* acts, float* next,
                  float* reward, uint8_t* done, int64_t m, void* stream);
int pi_probe_interp(pi_handle* h, const float* pts, int32_t* idxs, float* wgts, int64_t m,
                    void* stream);
int pi_probe_coords(pi_handle* h, int64_t s_begin, int64_t s_end, float* out, int chunks_per_workgroup,
                    void* stream);
/*
 * The workgroup -> chunk schedule a sweep launch would use (host only, no launch; tests and tools): a launch of
 * `block`-thread workgroups taking `chunks_per_workgroup` chunks each over `count` units (states, or entries of a state
 * list) from unit `first`, where `total` units stand for the whole grid (n_states for a state range).
 * out6 = {grid x, grid y, period, phase, chunks per workgroup, 0}: period == 0 is the slab schedule (grid y = 1; XCD x =
 * workgroup index mod 8 walks the x-th contiguous eighth of the groups); otherwise the STRIP schedule (pi_set_option 7,
 * PI_MI355_STRIP, pi_info 35): the groups are cut into periods of `period` groups — a plane of a slow memory
 * dimension —, the launch is two-dimensional (y = period p, x = 8 r + XCD) and XCD x takes groups
 * [floor((x period + rot) / 8), floor(((x + 1) period + rot) / 8)), rot = 3 p mod 8, of EVERY period p, so that what an
 * XCD's L2 sees between the two sweeps that read a line of V is an eighth of a plane instead of a whole one.  Placement only: results do not depend on it.  No reference counterpart
 * (the reference launches one thread per state in index order, src/cuda_policy_iteration.py:305-318).
 */
int pi_plan_schedule(pi_handle* h, int block, int64_t first, int64_t count, int64_t total, int chunks_per_workgroup,
                     uint64_t* out6);

/* ---------------------------------------------------------------------------------------------
 * Inference: the reference's CPU helper utils/barycentric.py as one batched
 * device kernel — get_barycentric_weights_and_indices (:15-77) and get_optimal_action (:80-112)
 * for m states per launch, same arithmetic type for type (float64 cell widths, the POINT clamped
 * to the bounds, weights = float64 products rounded once to float32, corners in the order of the
 * caller's corner_bits table — itertools.product, MSB first, in the reference :233).  Independent of
 * pi_handle: needs no env plugin.
 *   pi_infer_create     host arrays; corner_bits is (n_corners = 2^D, D) int32 of 0/1; device = -1
 *                       builds the kernel only (compile check without a GPU); cache_dir as pi_compile.
 *                       The grid and the corner table are compiled INTO the kernel (hipRTC, one code
 *                       object per grid, cached on disk like the sweep kernels)
 *   pi_infer_set_policy host arrays: the greedy policy (n_states int32) and the action values; copied
 *                       to the device once, validated (every entry an index into action_space)
 *   pi_infer_query      d_points (m, D) float32 on the device; any of the three outputs may be null:
 *                       d_actions_out (m) float32 = sum_c w[c] * action_space[policy[idx[c]]] (ascending c),
 *                       d_weights_out (m, 2^D) float32, d_indices_out (m, 2^D) int32.  Asynchronous.
 * ------------------------------------------------------------------------------------------- */
typedef struct pi_infer pi_infer;
pi_infer* pi_infer_create(int device, int D, const float* lo, const float* hi, const int32_t* grid_shape,
                          const int32_t* strides, const int32_t* corner_bits, int64_t n_corners,
                          const char* cache_dir);
void pi_infer_destroy(pi_infer* h);
int pi_infer_set_policy(pi_infer* h, const int32_t* policy, int64_t n_states, const float* action_space,
                        int n_actions);
int pi_infer_query(pi_infer* h, const float* d_points, int64_t m, float* d_actions_out, float* d_weights_out,
                   int32_t* d_indices_out, void* stream);

/* Tuning: 0 = chunks per workgroup of the evaluation sweeps, 1 = of the improvement / value sweeps
 * (1..64), 2 = replay small evaluation batches as hipGraphs (0 | 1), 3 = run whole-grid batches of
 * small grids in the LDS-resident kernel (0 | 1), 5 = build the fused swept-first kernel of the peer-to-peer exchange
 * now (value != 0; after pi_compile; a compile check on host-only handles), 6 = keep the live-state list of
 * pi_prepare_mask even when it fills no idle lanes (0 | 1; the fused exchange of a sharded run delivers from the list sweeps),
 * 8 = report that the last pi_policy_iteration launch on the XCD-local kernel did not go through (1: count it — two
 * failures switch that kernel off for the handle —, 2: and switch it off now),
 * 7 = STRIP SCHEDULE of the sweeps: states per period (see pi_plan_schedule; -1 = the library's choice for the grid, the
 * default; 0 = slab schedule; PI_MI355_STRIP in the environment at pi_create sets the same), 4 = MEMORY ORDER of the dimensions (before pi_compile, once):
 * digit k (base 8) of the value is the dimension — numbered as in pi_create and in step_dynamics' arguments —
 * that is stored as memory dimension k, 0 = slowest; e.g. 03120 (octal) = order (0, 2, 1, 3).  From then on
 * EVERY flat state index of this ABI (s_begin / s_end, the entries of V, policy and the mask, the indices the
 * interpolation probe returns) refers to that order: index = sum_k i_{order[k]} * stride_k with row-major
 * strides over the permuted shape.  The reference has one order, the meshgrid's (:84-87); which dimensions
 * are slow decides how far apart in memory the 2^D corners of a successor cell lie and how long a value
 * stays useful in an XCD's L2 — the best order beats the user's by 7-11 % on the evaluation sweeps of the
 * big BASELINE grids (tools/dim_order_sweep.py).  The order-sensitive arithmetic (corner weights as products
 * over the dimensions, the fmaf chain over the corners, :580-614, :616-649) stays in the caller's dimension
 * order, so results do not depend on the memory order, bit for bit; neither do they depend on options 0-3 and 7. */
int pi_set_option(pi_handle* h, int what, int64_t value);

/* Introspection: 0 n_states, 1 n_actions, 2 D, 3 chunks per workgroup (evaluation), 4 VGPRs of the
 * eval kernel, 5 VGPRs of the improve kernel, 6 compute units, 7 = 1 if the last pi_compile was
 * served from the cache, 8 chunks per workgroup (improvement), 9 cached graphs, 10 graphs enabled,
 * 11 / 12 threads per workgroup (evaluation / improvement), 13 states per thread of the LDS-resident
 * batch kernel (0: grid too big for it), 14 that kernel enabled, 15 checked kernels (pi_debug_report),
 * 35 states per period of the strip schedule in use (0: slab schedule),
 * 16 live states listed by pi_prepare_mask (0: no list in use), 17 entries of the per-evaluation list of
 * pi_eval_begin (0: none), 18 the memory order as set with pi_set_option 4 (octal digits, identity by default),
 * 20+d = 1 if dimension d's interpolation division runs through the proven reciprocal path. */
int64_t pi_info(pi_handle* h, int what);

/*
 * Checked build (sanitizer row: the reference has none; GPU address sanitizers are not
 * available on this platform).  With PI_MI355_DEBUG=1 in the environment when pi_create runs, the handle's
 * kernels check every index they derive from DATA before using it — the action a policy entry names
 * (kind 1: where = flat state, value = the entry) and the cell a successor falls in (kind 2: where = the
 * cell's flat index) — count violations, remember the first and carry on with index 0 instead of reading
 * out of bounds.  pi_debug_report synchronises the device, copies {violations, kind, where, value} of the
 * sweeps launched since the last report to out4 (host) and clears them; it fails on a handle built
 * without the checks.  Results of a run without violations are those of the unchecked kernels.
 */
int pi_debug_report(pi_handle* h, uint32_t* out4);

#ifdef __cplusplus
}
#endif
#endif /* PI_MI355_H_ */
