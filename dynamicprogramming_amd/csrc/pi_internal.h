// pi_internal.h — shared between pi_api.cpp (handles, hipRTC, launches) and pi_comm.cpp
// (multi-GPU transports and the sharded sweep driver).  Not part of the C ABI.
#pragma once

#include "pi_mi355.h"

#include <hip/hip_runtime.h>

#include <cstdint>
#include <memory>
#include <string>
#include <vector>

namespace pi {

int fail(const std::string& msg);                 // records the thread-local error, returns 1
const std::string& last_error();

#define PI_HIP(expr)                                                                      \
    do {                                                                                  \
        hipError_t e_ = (expr);                                                           \
        if (e_ != hipSuccess)                                                             \
            return ::pi::fail(std::string(#expr) + ": " + hipGetErrorString(e_));         \
    } while (0)

constexpr int kProbeBlock = 256; // threads per workgroup of the probe kernels' host launches
constexpr int kXcds = 8;         // PI_NXCD
constexpr int kSlots = 256;      // PI_NSLOT

// Makes `device` current for the lifetime of the guard and restores the caller's device after.
struct DeviceGuard {
    int saved = -1;
    bool active = false;
    explicit DeviceGuard(int device) {
        if (device < 0) return;
        if (hipGetDevice(&saved) != hipSuccess) saved = -1;
        if (saved != device) active = hipSetDevice(device) == hipSuccess;
    }
    ~DeviceGuard() {
        if (active && saved >= 0) (void)hipSetDevice(saved);
    }
};

// Transport of a multi-rank run (pi_comm.cpp: RCCL and the in-process test transport; pi_p2p.cpp: peer-to-peer stores
// into IPC-mapped buffers).  send / recv between group_begin and group_end have the stream semantics of an RCCL group:
// everything enqueued on `st` before group_end is visible to the transfer, everything enqueued after it sees the result.
struct Comm {
    int rank = 0, world = 1;
    virtual ~Comm() {}
    virtual const char* kind() const = 0;
    virtual int group_begin() = 0;
    virtual int send(const void* p, size_t bytes, int peer, hipStream_t st) = 0;
    virtual int recv(void* p, size_t bytes, int peer, hipStream_t st) = 0;
    virtual int group_end(hipStream_t st) = 0;
    // every rank contributes `bytes` at full + rank * bytes; afterwards all ranks hold all
    virtual int allgather(void* full, size_t bytes, hipStream_t st) = 0;
    virtual int allreduce_max_f32(float* d, hipStream_t st) = 0;
    virtual int allreduce_sum_u32(uint32_t* d, hipStream_t st) = 0;
    // a sharded batch failed on this rank: stop peers of the same process from waiting (in-process transport)
    virtual void give_up() {}
    // Fused exchange (peer-to-peer transport only): the swept-first kernel stores its rows into the peers' buffers
    // itself (pi_eval_push_kernel), everything on the ONE stream the sweeps run on.  Per sweep:
    //   push_begin   "my receives are posted" to `senders`, wait for the same from `receivers` (one wave); returns the
    //                device table of the receivers' addresses of `local_dst` (the full V' buffer being written)
    //   <the caller launches the push kernel with that table>
    //   push_signal  raise the receivers' data counters — the kernel boundary in front of it is the release
    //   <the caller launches the interior>
    //   push_wait    wait for the senders' data counters (push_wait_deferred: leave it to the next sweep's push_begin, which
    //                then waits for both in its one launch)
    // Same message numbers as send / recv groups, so fused and unfused exchanges may alternate (all ranks alike).
    virtual bool can_push() const { return false; }
    virtual int push_begin(const std::vector<int>&, const std::vector<int>&, float*, float* const**, hipStream_t) { return 1; }
    virtual int push_signal(hipStream_t) { return 1; }
    virtual int push_wait(hipStream_t) { return 1; }
    // instead of push_wait when the very next call on the stream is the push_begin of another fused sweep
    virtual int push_wait_deferred(hipStream_t st) { return push_wait(st); }
};
struct ShardPlan;
struct P2pPending;   // pi_p2p.cpp: what pi_p2p_describe allocated for a pi_comm_init_p2p that has not happened yet

struct GraphEntry {               // one captured batch of evaluation sweeps
    const void *Va = nullptr, *Vb = nullptr, *policy = nullptr, *term = nullptr, *d_delta = nullptr;
    int64_t s_begin = 0, s_end = 0;
    float gamma = 0.0f;
    int n_sweeps = 0;
    hipGraphExec_t exec = nullptr;
    uint64_t stamp = 0;
};

}  // namespace pi

struct pi_handle {
    int device = -1;
    int D = 0;
    int n_actions = 0;
    int64_t n_states = 0;
    std::vector<int32_t> shape;
    std::vector<float> lo, span, rcp;
    std::vector<int> fastdiv;
    // Memory order of the dimensions (pi_set_option 4): memory dimension k (0 = slowest) is user dimension user_of_mem[k];
    // user dimension d lives in memory dimension mem_of_user[d].  shape / lo / span / rcp / fastdiv / the bin tables
    // are kept in MEMORY order; every flat state index of the C ABI is an index in that order.  Identity by default.
    std::vector<int> mem_of_user, user_of_mem;
    bool compiled = false;               // pi_compile has run: the memory order can no longer change
    std::vector<float> tab;              // actions | bins_0 | bins_1 | ...
    float* d_tab = nullptr;
    unsigned int* d_slots = nullptr;     // 2 x kSlots accumulator words: residual bits | changed
    hipModule_t module = nullptr;
    // second module, built on demand from the same translation unit + csrc/pi_push_kernels.hip (ensure_push_module)
    hipModule_t module_push = nullptr;
    hipFunction_t f_eval_push = nullptr, f_reach_pairs = nullptr;
    std::string dynamics_src, cache_dir; // what pi_compile was given (for the second module)
    bool has_cache_dir = false;
    hipFunction_t f_eval = nullptr, f_eval_live = nullptr, f_policy_list = nullptr, f_scan_slots = nullptr, f_mask_list = nullptr, f_improve = nullptr, f_improve_live = nullptr, f_value = nullptr, f_finalize = nullptr,
                  f_reach_planes = nullptr, f_reach_units = nullptr, f_probe_step = nullptr, f_probe_interp = nullptr,
                  f_probe_coords = nullptr, f_resident = nullptr, f_run_resident = nullptr;
    int num_cu = 0;
    int block_eval = 256, block_improve = 256;   // threads per workgroup = states per chunk
    int cpw_eval = 1, cpw_improve = 1;           // chunks a workgroup sweeps
    // Strip schedule of the sweeps (pi_set_option 7, PI_MI355_STRIP): states per period — a plane of a slow memory
    // dimension; every XCD takes its eighth of every period.  0 = the slab schedule (an XCD walks one contiguous run).
    int64_t strip_states = 0;
    int64_t strip_mode = -1;             // -1 the library's choice (resolve_strip), 0 off, > 0 states per period as given
    int vgpr_eval = -1, vgpr_improve = -1;
    bool cache_hit = false;
    bool debug_bounds = false;           // PI_MI355_DEBUG=1 at pi_create: checked kernels (pi_debug_report)
    bool use_graphs = true;
    // LDS-resident evaluation batches (grids of up to ~12 k states): states per thread of the one
    // workgroup (resident_block threads), 0 = this grid is too big; the switch is pi_set_option 3
    int resident_k = 0, resident_block = 1024;
    bool use_resident = true;
    // Dataflow evaluation (pi_eval_flow_kernel): launch-bound grids that do not fit one CU's LDS run a whole policy
    // evaluation in one launch, the iterates travelling between workgroups as tagged granules.  flow: this grid
    // qualifies; f_flow is null when the occupancy query does not admit all workgroups at once.  d_flow: ring of 16
    // granule versions | progress words + status | check slots (owned; flow_bytes).
    bool flow = false;
    int flow_block = 256;
    hipFunction_t f_flow = nullptr, f_flow_finish = nullptr;
    void* d_flow = nullptr;
    size_t flow_bytes = 0;
    // XCD-local evaluation / whole run (pi_xcd_kernel, csrc/pi_onelaunch_kernels.hip): one 1 024-thread workgroup per CU of
    // ONE XCD, the iterates as tagged granules through that XCD's L2, flag granules as the barrier every 32nd sweep.
    // xcd: this grid qualifies; xcd_off: switched off after repeated failures (pi_policy_evaluation counts its own; a failed
    // whole run is reported back through pi_set_option 8).  d_xcd: ring of xcd_ring versions | scratch policy | control
    // words (owned).  Counters for pi_info 30-33.
    bool xcd = false, xcd_off = false;
    int xcd_states = 1024;                               // states per workgroup (PI_XCD_S)
    int xcd_ring = 64;                                   // granule versions of V the kernel keeps (PI_XCD_RING)
    hipFunction_t f_xcd = nullptr, f_xcd_finish = nullptr;
    void* d_xcd = nullptr;
    size_t xcd_bytes = 0;
    int64_t xcd_used = 0, xcd_failed = 0, whole_runs = 0;
    unsigned int* xcd_ctl = nullptr;                     // control words of the last launch (inside d_xcd)
    std::vector<pi::GraphEntry> graphs;
    uint64_t graph_clock = 0;
    // pi_prepare_mask: the non-terminal states of the mask at live_term, ascending (device, owned); in use only
    // when visiting them instead of all states saves enough idle lanes (live_count > 0)
    const uint8_t* live_term = nullptr;
    int32_t* d_live = nullptr;
    int64_t live_count = 0;
    bool live_force = false;             // pi_set_option 6: keep the list whatever the share of idle lanes
    int64_t live_lo = 0, live_hi = 0;    // the state range the list covers (pi_prepare_mask_range; the whole grid otherwise)
    // pi_eval_begin .. pi_eval_end: the live states that bootstrap under the policy at eval_policy (device list,
    // capacity live_count); eval_count < 0: none.  eval_holds: buffers every live state of which has been written
    // by a full sweep since pi_eval_begin — a sweep may use the shorter list only between two such buffers.
    int32_t* d_eval_list = nullptr;
    unsigned long long* d_eval_cursor = nullptr;     // per-block survivor counts -> offsets (+ total)
    int64_t eval_count = -1;
    const int32_t* eval_policy = nullptr;
    std::vector<const float*> eval_holds;
    std::vector<uint64_t> live_bits;     // host: bit s of word s / 64 = state s is live (30 MB for 25^6)
    std::vector<int64_t> live_before;    // host: live states before each 64-state block -> list position of any state
    pi::Comm* comm = nullptr;            // multi-GPU transport (owned; pi_comm.cpp), null = single rank
    pi::ShardPlan* plan = nullptr;       // exchange plan (owned; pi_comm.cpp)
    pi::P2pPending* p2p_pending = nullptr;   // pi_p2p_describe without its pi_comm_init_p2p yet (owned; pi_p2p.cpp)
};

namespace pi {

int check_ready(const pi_handle* h);
int check_range(const pi_handle* h, int64_t s_begin, int64_t s_end);
// One evaluation sweep over [s_begin, s_end) WITHOUT the finalize step: the residual stays in the
// accumulator slots (want_delta) until finalize_delta is called, so several sub-range launches of
// one logical sweep can share it.
// keep_terminals: Vnew already holds the terminal states' values (every sweep of a ping-pong batch but the
// first), so the sweep neither copies them nor streams old values for them.
int launch_eval(pi_handle* h, const float* V, float* Vnew, const int32_t* policy, const uint8_t* term,
                int64_t s_begin, int64_t s_end, float gamma, bool want_delta, hipStream_t st,
                bool keep_terminals = false);
int finalize(pi_handle* h, float* d_delta, uint32_t* d_changed, hipStream_t st);
// The live-state list of pi_prepare_mask.  live_usable: a list is in use, `term` is the mask it was built from and
// [s_begin, s_end) lies inside the range it covers.
// live_span: the run of list entries whose states lie in [s_begin, s_end).  launch_eval_live: one evaluation sweep
// over `count` list entries from position `first` — for sweeps that need not copy terminal values.
bool live_usable(const pi_handle* h, const uint8_t* term, int64_t s_begin, int64_t s_end);
void live_span(const pi_handle* h, int64_t s_begin, int64_t s_end, int64_t* first, int64_t* count);
void live_states(const pi_handle* h, int64_t s_begin, int64_t s_end, std::vector<int32_t>& out);   // appends, ascending
// list_total (lists the caller passes): what the list would count over the whole grid, for the strip schedule; 0 = slab
int launch_eval_live(pi_handle* h, const float* V, float* Vnew, const int32_t* policy, int64_t first, int64_t count,
                     float gamma, bool want_delta, hipStream_t st, const int32_t* list = nullptr, int64_t list_total = 0);
int64_t live_list_total(const pi_handle* h, int64_t entries);
// PiSched of pi_sweep_kernels.hip (same layout) and the planner that fills it: the launch grid (x, y)
struct Sched {
    int cpw;
    unsigned int period, phase;
};
struct Grid2 { unsigned x, y; };
Grid2 plan_launch(const pi_handle* h, int block, int64_t first, int64_t count, int64_t total, int cpw, Sched* sc);
// pi_eval_push_kernel over `count` entries of `list` with their destination masks: V'(s) goes to Vnew[s] and to
// peers[j][s] for every bit j of dest[k] (ensure_push_module builds the kernel the first time)
int ensure_push_module(pi_handle* h);
int launch_eval_push(pi_handle* h, const float* V, float* Vnew, const int32_t* policy, const int32_t* list,
                     const uint8_t* dest, float* const* d_peers, int n_peers, int64_t count, float gamma, bool want_delta,
                     hipStream_t st);
// Reach of [s_begin, s_end) under all actions in pairs (i_0, i_v), v = the memory dimension of the user's dimension 1:
// bit i_0 * g_v + i_v of d_bitmap (zeroed by the caller; g_0 * g_v bits).  Second module; 3-D and up, g_0 * g_v <= 2^17.
bool pairs_possible(const pi_handle* h);
int reach_pairs(pi_handle* h, const uint8_t* term, int64_t s_begin, int64_t s_end, uint32_t* d_bitmap, hipStream_t st);
void drop_eval_list(pi_handle* h);     // the policy may have changed: forget the per-evaluation list
void release_comm(pi_handle* h);      // pi_comm.cpp: tears down the transport and the exchange plan
void drop_p2p_pending(pi_handle* h);  // pi_p2p.cpp
double comm_timeout_seconds();        // PI_MI355_COMM_TIMEOUT (default 120): how long a rank waits for a peer
// hipRTC (gfx950, -O3 -ffp-contract=off) or the on-disk code-object cache -> image of one translation unit
int compile_image(const std::string& src, const char* cache_dir, char* log, size_t log_len,
                  std::vector<char>& image, bool* cache_hit);

}  // namespace pi
