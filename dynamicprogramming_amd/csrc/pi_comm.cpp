// pi_comm.cpp — the multi-GPU half of libpi_mi355.so: transports, the exchange plan and the
// sharded sweep driver (SURVEY.md section 8e / 8b "must export": pi_comm_init, pi_allgather_V,
// pi_allreduce_*).  The reference has no counterpart: src/cuda_policy_iteration.py:300-336 is a
// single-device loop.
//
// Partition: contiguous flat-index shards of `per` = ceil(n / world) states (slabs along
// dimension 0); every rank keeps a full-size V, sweeps its shard and then makes the new values
// visible where other ranks will read them, either
//   * halo exchange — exactly the dimension-0 plane runs each peer can reach (measured once for
//     ALL actions with pi_reach_planes_kernel), as grouped ncclSend/ncclRecv in place in V', the
//     planes peers wait for swept first and sent on a second HIP stream while the interior is
//     swept; or
//   * all-gather of the shards (ncclAllGather, in place) when the reachable bands cover most
//     of the grid anyway.
// Two transports implement the same small interface: RCCL (one process per GPU, xGMI), and an
// in-process one (several handles of ONE process, one host thread each, device-to-device copies
// ordered by HIP events) that exists so the stream ordering of the overlapped exchange can be
// tested with real kernels on a single GPU.

#include "pi_internal.h"

#include <rccl/rccl.h>

#include <algorithm>
#include <array>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <map>
#include <mutex>

using pi::fail;

namespace pi {

struct ShardPlan {
    int64_t per = 0, s_begin = 0, s_end = 0;
    bool halo = false;
    int depth = 1;                                           // granularity of the reach probe
    struct Seg { int src, dst; int64_t a, b; };
    std::vector<Seg> segs;                                   // only those that involve this rank
    std::vector<std::pair<int64_t, int64_t>> send_ranges, interior;
    int64_t recv_elems = 0, send_elems = 0;
    bool broken = false;            // a sharded sweep failed half-way: streams were drained; new communicator + plan needed
    hipStream_t comm_stream = nullptr;
    hipEvent_t ev_ready = nullptr, ev_done = nullptr;
    // Row-exact split (grids without terminal states): exactly the rows peers wait for, as an ascending state list
    // swept by pi_eval_live_kernel in ONE launch, and the rest of the shard as a second list — instead of a few
    // coarse contiguous ranges that also hold rows nobody waits for (C4 at 8 ranks: 80-100 % of the shard).
    bool row_exact = false;
    std::vector<std::pair<int64_t, int64_t>> first_exact, inner_exact;
    int32_t *d_first = nullptr, *d_inner = nullptr;
    int64_t n_first = 0, n_inner = 0;
    // Fused exchange (transports that can_push()): entry k of the swept-first list goes to the peers named by the bits
    // of d_first_dest[k] — bit j = push_peers[j] — stored by the swept-first kernel itself (pi_eval_push_kernel).
    std::vector<int> push_peers, push_senders;
    uint8_t* d_first_dest = nullptr;
    bool push_ok = false;
    // The same for the coarse-range plan of a grid WITH terminal states whose later sweeps of a batch run over the
    // live-state list (pi_prepare_mask_range): one byte per entry of the shard's live span.  Valid for the list it was
    // built from (live_list / live_count).
    uint8_t* d_live_dest = nullptr;
    int64_t live_dest_base = 0;
    const int32_t* live_list = nullptr;
    int64_t live_list_count = 0;
    bool live_push_ok = false;
    bool live_exact = false;            // d_first / d_inner / d_first_dest hold the split of the shard's LIVE states
    bool pair_exact = false;            // destination masks cut down to the pairs (i_0, i_v) each peer reads
    int64_t fused_send_elems = -1;      // (state, receiver) pairs one fused sweep delivers; -1: not a fused plan
    ~ShardPlan() {
        if (d_live_dest) (void)hipFree(d_live_dest);
        if (d_first_dest) (void)hipFree(d_first_dest);
        if (d_first) (void)hipFree(d_first);
        if (d_inner) (void)hipFree(d_inner);
        if (ev_ready) (void)hipEventDestroy(ev_ready);
        if (ev_done) (void)hipEventDestroy(ev_done);
        if (comm_stream) (void)hipStreamDestroy(comm_stream);
    }
};

void drop_plan(pi_handle* h) {
    delete h->plan;
    h->plan = nullptr;
}
void release_comm(pi_handle* h) {
    drop_plan(h);
    delete h->comm;
    h->comm = nullptr;
    drop_p2p_pending(h);
}

// How long a rank waits for a peer before it declares the exchange dead (seconds).  The RCCL
// transport has its own watchdog-free wait on the GPU; what is bounded here are the HOST waits of the
// in-process transport and the host-side polls for an asynchronous RCCL error.
double comm_timeout_seconds() {
    if (const char* e = std::getenv("PI_MI355_COMM_TIMEOUT")) {
        const double v = std::atof(e);
        if (v > 0.0) return v;
    }
    return 120.0;
}


}  // namespace pi

namespace {

#define PI_NCCL(expr)                                                                      \
    do {                                                                                   \
        ncclResult_t r_ = (expr);                                                          \
        if (r_ != ncclSuccess)                                                             \
            return fail(std::string(#expr) + ": " + ncclGetErrorString(r_));               \
    } while (0)

using pi::comm_timeout_seconds;

// ---- RCCL ------------------------------------------------------------------------------
// Failure handling ("RCCL error -> abort with message"): every RCCL call goes
// through check(); on the first error an open group is closed (a group left open would swallow
// every later call of the thread), the communicator is aborted (ncclCommAbort frees the peers'
// resources and unblocks kernels that wait for this rank) and the handle refuses further
// collectives with the recorded message.  After every group / collective the communicator's
// asynchronous error state is polled once (ncclCommGetAsyncError): a peer that died shows up there.
struct RcclComm : pi::Comm {
    ncclComm_t comm = nullptr;
    bool in_group = false;
    std::string dead;                       // non-empty once the communicator has been aborted
    ~RcclComm() override {
        if (comm) (void)ncclCommDestroy(comm);
    }
    const char* kind() const override { return "rccl"; }
    int abort_with(const std::string& why) {
        if (in_group) { (void)ncclGroupEnd(); in_group = false; }
        if (comm) { (void)ncclCommAbort(comm); comm = nullptr; }
        dead = "RCCL communicator aborted (rank " + std::to_string(rank) + " of " + std::to_string(world) + "): " + why;
        return fail(dead);
    }
    int check(ncclResult_t r, const char* what) {
        if (r == ncclSuccess) return 0;
        return abort_with(std::string(what) + ": " + ncclGetErrorString(r));
    }
    int alive() { return dead.empty() && comm ? 0 : fail(dead.empty() ? "no RCCL communicator" : dead); }
    int poll_async() {
        ncclResult_t st = ncclSuccess;
        if (check(ncclCommGetAsyncError(comm, &st), "ncclCommGetAsyncError")) return 1;
        if (st != ncclSuccess && st != ncclInProgress)
            return abort_with(std::string("asynchronous error: ") + ncclGetErrorString(st));
        return 0;
    }
    int group_begin() override {
        if (alive()) return 1;
        if (check(ncclGroupStart(), "ncclGroupStart")) return 1;
        in_group = true;
        return 0;
    }
    int send(const void* p, size_t bytes, int peer, hipStream_t st) override {
        if (alive()) return 1;
        return check(ncclSend(p, bytes, ncclChar, peer, comm, st), "ncclSend");
    }
    int recv(void* p, size_t bytes, int peer, hipStream_t st) override {
        if (alive()) return 1;
        return check(ncclRecv(p, bytes, ncclChar, peer, comm, st), "ncclRecv");
    }
    int group_end(hipStream_t) override {
        if (alive()) return 1;
        in_group = false;
        if (check(ncclGroupEnd(), "ncclGroupEnd")) return 1;
        return poll_async();
    }
    int allgather(void* full, size_t bytes, hipStream_t st) override {
        if (alive()) return 1;
        if (check(ncclAllGather(static_cast<const char*>(full) + (size_t)rank * bytes, full, bytes, ncclChar,
                                comm, st), "ncclAllGather")) return 1;
        return poll_async();
    }
    int allreduce_max_f32(float* d, hipStream_t st) override {
        if (alive()) return 1;
        if (check(ncclAllReduce(d, d, 1, ncclFloat, ncclMax, comm, st), "ncclAllReduce(max)")) return 1;
        return poll_async();
    }
    int allreduce_sum_u32(uint32_t* d, hipStream_t st) override {
        if (alive()) return 1;
        if (check(ncclAllReduce(d, d, 1, ncclUint32, ncclSum, comm, st), "ncclAllReduce(sum)")) return 1;
        return poll_async();
    }
};

// ---- in-process transport (test transport: N handles, N host threads, one process) -------
// send(): records an event on the sender's stream and posts {pointer, bytes, event} in the
// mailbox of the (src, dst) pair.  recv() (deferred to group_end): waits on the host for the
// posting, makes the receiver's stream wait for the sender's event, enqueues the device copy and
// records a completion event.  group_end() finally makes the SENDER's stream wait for those
// completion events, so a later kernel cannot overwrite a buffer that is still being copied —
// the same stream semantics an RCCL send has.
struct LocalGroup {
    std::mutex mu;
    std::condition_variable cv;
    int world = 0, joined = 0, left = 0;
    bool failed = false;                                     // a member gave up: nobody waits any longer
    struct Msg {
        const void* src = nullptr;
        size_t bytes = 0;
        hipEvent_t ready = nullptr, done = nullptr;
        bool copied = false;
    };
    std::vector<std::deque<std::shared_ptr<Msg>>> box;        // [src * world + dst]
    // host-side scalar reductions
    std::vector<double> red;
    int red_arrived = 0, red_epoch = 0;
    double red_result = 0.0;
};
std::mutex g_groups_mu;
std::map<std::string, std::shared_ptr<LocalGroup>> g_groups;

struct LocalComm : pi::Comm {
    std::shared_ptr<LocalGroup> grp;
    std::string name;
    struct PendingRecv { void* p; size_t bytes; int peer; hipStream_t st; };
    std::vector<PendingRecv> recvs;
    std::vector<std::shared_ptr<LocalGroup::Msg>> sent, retired;
    void drop_retired() {
        for (auto& m : retired) {
            (void)hipEventDestroy(m->ready);
            (void)hipEventDestroy(m->done);
        }
        retired.clear();
    }
    ~LocalComm() override {
        (void)hipDeviceSynchronize();
        drop_retired();
        std::lock_guard<std::mutex> lk(g_groups_mu);
        if (grp && ++grp->left == grp->world) g_groups.erase(name);
    }
    const char* kind() const override { return "local"; }
    // Every host wait of this transport is bounded: a peer that failed (or never arrives) turns into
    // an error on all members after PI_MI355_COMM_TIMEOUT seconds instead of a hang.
    template <typename Pred>
    int wait_for(std::unique_lock<std::mutex>& lk, Pred ready, const char* what) {
        const auto limit = std::chrono::duration<double>(comm_timeout_seconds());
        if (!grp->cv.wait_for(lk, limit, [&] { return grp->failed || ready(); }) || grp->failed) {
            grp->failed = true;
            grp->cv.notify_all();
            return fail(std::string("local transport: rank ") + std::to_string(rank) + " gave up waiting for " + what);
        }
        return 0;
    }
    void give_up() override {
        std::lock_guard<std::mutex> lk(grp->mu);
        grp->failed = true;
        grp->cv.notify_all();
    }
    int group_begin() override { return 0; }
    int send(const void* p, size_t bytes, int peer, hipStream_t st) override {
        auto m = std::make_shared<LocalGroup::Msg>();
        m->src = p;
        m->bytes = bytes;
        PI_HIP(hipEventCreateWithFlags(&m->ready, hipEventDisableTiming));
        PI_HIP(hipEventCreateWithFlags(&m->done, hipEventDisableTiming));
        PI_HIP(hipEventRecord(m->ready, st));
        {
            std::lock_guard<std::mutex> lk(grp->mu);
            grp->box[(size_t)rank * world + peer].push_back(m);
        }
        grp->cv.notify_all();
        sent.push_back(m);
        return 0;
    }
    int recv(void* p, size_t bytes, int peer, hipStream_t st) override {
        recvs.push_back({p, bytes, peer, st});
        return 0;
    }
    int group_end(hipStream_t st) override {
        for (const PendingRecv& r : recvs) {
            std::shared_ptr<LocalGroup::Msg> m;
            {
                std::unique_lock<std::mutex> lk(grp->mu);
                auto& q = grp->box[(size_t)r.peer * world + rank];
                if (wait_for(lk, [&] { return !q.empty(); }, "a send of its peer")) { recvs.clear(); return 1; }
                m = q.front();
                q.pop_front();
            }
            if (m->bytes != r.bytes) {
                give_up();
                recvs.clear();
                return fail("local transport: send/recv size mismatch");
            }
            PI_HIP(hipStreamWaitEvent(r.st, m->ready, 0));
            PI_HIP(hipMemcpyAsync(r.p, m->src, r.bytes, hipMemcpyDeviceToDevice, r.st));
            PI_HIP(hipEventRecord(m->done, r.st));
            {
                std::lock_guard<std::mutex> lk(grp->mu);
                m->copied = true;
            }
            grp->cv.notify_all();
        }
        recvs.clear();
        for (auto& m : sent) {
            {
                std::unique_lock<std::mutex> lk(grp->mu);
                if (wait_for(lk, [&] { return m->copied; }, "its peer to receive")) return 1;
            }
            PI_HIP(hipStreamWaitEvent(st, m->done, 0));
        }
        // both streams have enqueued their waits; the events are destroyed in batches, after a
        // synchronisation, so none is released while a stream still refers to it
        for (auto& m : sent) retired.push_back(m);
        sent.clear();
        if (retired.size() >= 512) {
            PI_HIP(hipDeviceSynchronize());
            drop_retired();
        }
        return 0;
    }
    int allgather(void* full, size_t bytes, hipStream_t st) override {
        for (int p = 0; p < world; ++p)
            if (p != rank && send(static_cast<const char*>(full) + (size_t)rank * bytes, bytes, p, st)) return 1;
        for (int p = 0; p < world; ++p)
            if (p != rank && recv(static_cast<char*>(full) + (size_t)p * bytes, bytes, p, st)) return 1;
        return group_end(st);
    }
    // scalar reductions go through the host (this transport only exists for tests)
    int reduce_host(double mine, bool is_max, double* out) {
        std::unique_lock<std::mutex> lk(grp->mu);
        const int epoch = grp->red_epoch;
        grp->red.push_back(mine);
        if (++grp->red_arrived == world) {
            double r = is_max ? grp->red[0] : 0.0;
            for (double v : grp->red) r = is_max ? std::max(r, v) : r + v;
            grp->red_result = r;
            grp->red_arrived = 0;
            grp->red.clear();
            ++grp->red_epoch;
            grp->cv.notify_all();
        } else if (wait_for(lk, [&] { return grp->red_epoch != epoch; }, "the other ranks of a reduction")) {
            return 1;
        }
        *out = grp->red_result;
        return 0;
    }
    int allreduce_max_f32(float* d, hipStream_t st) override {
        float v;
        PI_HIP(hipMemcpyAsync(&v, d, sizeof v, hipMemcpyDeviceToHost, st));
        PI_HIP(hipStreamSynchronize(st));
        double r;
        if (reduce_host((double)v, true, &r)) return 1;
        v = (float)r;
        PI_HIP(hipMemcpyAsync(d, &v, sizeof v, hipMemcpyHostToDevice, st));
        PI_HIP(hipStreamSynchronize(st));
        return 0;
    }
    int allreduce_sum_u32(uint32_t* d, hipStream_t st) override {
        uint32_t v;
        PI_HIP(hipMemcpyAsync(&v, d, sizeof v, hipMemcpyDeviceToHost, st));
        PI_HIP(hipStreamSynchronize(st));
        double r;
        if (reduce_host((double)v, false, &r)) return 1;
        v = (uint32_t)r;
        PI_HIP(hipMemcpyAsync(d, &v, sizeof v, hipMemcpyHostToDevice, st));
        PI_HIP(hipStreamSynchronize(st));
        return 0;
    }
};

int need_comm(pi_handle* h) {
    if (pi::check_ready(h)) return 1;
    if (!h->comm) return fail("no communicator on this handle: call pi_comm_init first");
    return 0;
}
int need_plan(pi_handle* h) {
    if (need_comm(h)) return 1;
    if (!h->plan) return fail("no exchange plan on this handle: call pi_exchange_plan first");
    if (h->plan->broken)
        return fail("a sharded sweep failed on this handle and its exchange was abandoned; the communicator is not usable "
                    "any more (RCCL: aborted; in-process: its group gave up).  Recover on EVERY rank: pi_comm_destroy, a new "
                    "pi_comm_init (fresh id) / pi_comm_init_local (fresh group name), then pi_exchange_plan");
    return 0;
}

// Post the halo sends/receives of one freshly swept buffer on `st` (one transport group).
int post_exchange(pi_handle* h, float* full, hipStream_t st) {
    pi::Comm* c = h->comm;
    const pi::ShardPlan* p = h->plan;
    if (c->group_begin()) return 1;
    for (const auto& s : p->segs) {
        const size_t bytes = (size_t)(s.b - s.a) * sizeof(float);
        const int rc = s.src == c->rank ? c->send(full + s.a, bytes, s.dst, st) : c->recv(full + s.a, bytes, s.src, st);
        if (rc) {                       // the transport has closed its group (RCCL: and aborted); keep its message
            const std::string why = pi::last_error();
            (void)c->group_end(st);
            return fail(why);
        }
    }
    return c->group_end(st);
}

// Fault injection for the tests (PI_MI355_DEBUG_DROP_DELIVERY=k): every k-th destination mask of a fused plan is cleared —
// the state is still swept, its value is just not delivered — so that a test can show its poisoning check FAILS when a
// delivery is missing (tests/test_gpu_p2p.py).  Off unless the variable is set.
void drop_deliveries_for_tests(std::vector<uint8_t>& dest) {
    const char* e = std::getenv("PI_MI355_DEBUG_DROP_DELIVERY");
    const int k = e ? std::atoi(e) : 0;
    if (k > 0)
        for (size_t i = (size_t)k - 1; i < dest.size(); i += (size_t)k) dest[i] = 0;
}

// This rank's reach bitmap (`units` bits on the device, produced by `probe`) -> one byte per unit -> every rank's, through
// the transport: out[r * units + u] != 0 <=> rank r reads unit u.  Collective; blocks on `st` (planning is one-off).
template <typename Probe>
int gather_reach(pi::Comm* c, int64_t units, Probe probe, hipStream_t st, std::vector<uint8_t>& out, const char* what) {
    const size_t words = (size_t)(units + 31) / 32;
    uint32_t* d_bits = nullptr;
    uint8_t* d_all = nullptr;
    PI_HIP(hipMalloc((void**)&d_bits, words * sizeof(uint32_t)));
    std::vector<uint32_t> bits(words, 0u);
    int rc = hipMemsetAsync(d_bits, 0, words * sizeof(uint32_t), st) == hipSuccess ? probe(d_bits) : 1;
    if (!rc && hipMemcpyAsync(bits.data(), d_bits, words * sizeof(uint32_t), hipMemcpyDeviceToHost, st) != hipSuccess) rc = 1;
    if (!rc && hipStreamSynchronize(st) != hipSuccess) rc = 1;
    (void)hipFree(d_bits);
    if (rc) return fail(std::string(what) + ": reach probe failed: " + pi::last_error());
    const size_t row = (size_t)(units + 3) / 4 * 4;              // padded to whole words for the transport
    std::vector<uint8_t> all((size_t)c->world * row, 0);
    for (int64_t u = 0; u < units; ++u) all[(size_t)c->rank * row + (size_t)u] = (bits[(size_t)u >> 5] >> (u & 31)) & 1u;
    PI_HIP(hipMalloc((void**)&d_all, all.size()));
    hipError_t e = hipMemcpyAsync(d_all, all.data(), all.size(), hipMemcpyHostToDevice, st);
    if (e == hipSuccess) rc = c->allgather(d_all, row, st);
    if (e == hipSuccess && !rc) e = hipMemcpyAsync(all.data(), d_all, all.size(), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess && !rc) e = hipStreamSynchronize(st);
    (void)hipFree(d_all);
    if (e != hipSuccess) return fail(std::string(what) + ": " + hipGetErrorString(e));
    if (rc) return 1;
    out.resize((size_t)c->world * (size_t)units);
    for (int r = 0; r < c->world; ++r) std::memcpy(&out[(size_t)r * units], &all[(size_t)r * row], (size_t)units);
    return 0;
}

}  // namespace

extern "C" {

int pi_comm_unique_id(void* id128) {
    if (!id128) return fail("null argument");
    ncclUniqueId id;
    PI_NCCL(ncclGetUniqueId(&id));
    static_assert(sizeof id == 128, "RCCL unique id is 128 bytes");
    std::memcpy(id128, &id, sizeof id);
    return 0;
}

int pi_comm_init(pi_handle* h, int rank, int world, const void* id128) {
    if (!h || !id128) return fail("null argument");
    if (h->device < 0) return fail("host-only handle cannot own a communicator");
    if (world < 1 || rank < 0 || rank >= world) return fail("rank/world out of range");
    pi::DeviceGuard guard(h->device);
    std::unique_ptr<RcclComm> c(new RcclComm);
    c->rank = rank;
    c->world = world;
    ncclUniqueId id;
    std::memcpy(&id, id128, sizeof id);
    PI_NCCL(ncclCommInitRank(&c->comm, world, id, rank));
    pi::release_comm(h);
    h->comm = c.release();
    return 0;
}

int pi_comm_init_local(pi_handle* h, int rank, int world, const char* group_name) {
    if (!h || !group_name) return fail("null argument");
    if (h->device < 0) return fail("host-only handle cannot own a communicator");
    if (world < 1 || rank < 0 || rank >= world) return fail("rank/world out of range");
    std::unique_ptr<LocalComm> c(new LocalComm);
    c->rank = rank;
    c->world = world;
    c->name = group_name;
    {
        std::lock_guard<std::mutex> lk(g_groups_mu);
        auto& g = g_groups[c->name];
        if (!g) {
            g = std::make_shared<LocalGroup>();
            g->world = world;
            g->box.resize((size_t)world * world);
        }
        if (g->world != world) return fail("local group exists with a different world size");
        ++g->joined;
        c->grp = g;
    }
    pi::release_comm(h);
    h->comm = c.release();
    return 0;
}

int pi_comm_destroy(pi_handle* h) {
    if (!h) return fail("null handle");
    pi::DeviceGuard guard(h->device);
    pi::release_comm(h);
    return 0;
}

int pi_comm_info(pi_handle* h, int what) {
    if (!h || !h->comm) return -1;
    switch (what) {
        case 0: return h->comm->rank;
        case 1: return h->comm->world;
        case 2: return std::strcmp(h->comm->kind(), "rccl") == 0 ? 1 : std::strcmp(h->comm->kind(), "p2p") == 0 ? 3 : 2;
        case 3: return h->plan ? (h->plan->halo ? 2 : 1) : 0;
        case 4: return h->plan ? h->plan->depth : 0;
        case 5: return h->plan && h->plan->row_exact ? 1 : 0;
        case 6: return h->plan && ((h->plan->row_exact && h->plan->push_ok) || h->plan->live_push_ok) ? 1 : 0;
        case 7: return h->plan && h->plan->pair_exact ? 1 : 0;
        case 8: return h->plan ? (int)std::min<int64_t>(h->plan->fused_send_elems, INT32_MAX) : -1;
        case 9: return h->plan && (h->plan->row_exact || h->plan->live_exact) ? (int)std::min<int64_t>(h->plan->n_first, INT32_MAX) : -1;
        case 10: return h->plan && (h->plan->row_exact || h->plan->live_exact) ? (int)std::min<int64_t>(h->plan->n_inner, INT32_MAX) : -1;
        default: return -1;
    }
}

int pi_allgather_V(pi_handle* h, float* V_full, int64_t shard_elems, void* stream) {
    if (need_comm(h)) return 1;
    if (!V_full || shard_elems < 0) return fail("bad argument");
    pi::DeviceGuard guard(h->device);
    return h->comm->allgather(V_full, (size_t)shard_elems * sizeof(float), (hipStream_t)stream);
}

int pi_allgather_policy(pi_handle* h, int32_t* policy_full, int64_t shard_elems, void* stream) {
    if (need_comm(h)) return 1;
    if (!policy_full || shard_elems < 0) return fail("bad argument");
    pi::DeviceGuard guard(h->device);
    return h->comm->allgather(policy_full, (size_t)shard_elems * sizeof(int32_t), (hipStream_t)stream);
}

int pi_allreduce_max_f32(pi_handle* h, float* d_value, void* stream) {
    if (need_comm(h)) return 1;
    pi::DeviceGuard guard(h->device);
    return h->comm->allreduce_max_f32(d_value, (hipStream_t)stream);
}

int pi_allreduce_sum_u32(pi_handle* h, uint32_t* d_value, void* stream) {
    if (need_comm(h)) return 1;
    pi::DeviceGuard guard(h->device);
    return h->comm->allreduce_sum_u32(d_value, (hipStream_t)stream);
}

// Pure host logic (no GPU, no communicator): which pieces of V' must travel between ranks.
//   reach[r * g0 + p] != 0  <=>  rank r's shard can read dimension-0 plane p (any action)
// Output: up to `cap` segments {src, dst, a, b} = "src sends V'[a, b) to dst" (every rank derives
// the same list from the same bitmaps).  Returns the number of segments (may exceed cap: call
// again with a larger buffer), or -1 on bad arguments.
int64_t pi_plan_segments(int world, int64_t g0, int64_t stride0, int64_t n_states, int64_t per,
                         const uint8_t* reach, int64_t* segs, int64_t cap) {
    if (world < 1 || g0 < 1 || stride0 < 1 || per < 1 || !reach) { fail("bad argument"); return -1; }
    int64_t count = 0;
    for (int dst = 0; dst < world; ++dst) {
        const uint8_t* need = reach + (size_t)dst * g0;
        int64_t p = 0;
        while (p < g0) {
            if (!need[p]) { ++p; continue; }
            int64_t q = p;
            while (q < g0 && need[q]) ++q;
            const int64_t lo = p * stride0, hi = std::min(q * stride0, n_states);
            for (int src = 0; src < world; ++src) {
                const int64_t a = std::max(lo, (int64_t)src * per);
                const int64_t b = std::min(hi, std::min(((int64_t)src + 1) * per, n_states));
                if (src != dst && a < b) {
                    if (segs && count < cap) {
                        segs[4 * count + 0] = src;
                        segs[4 * count + 1] = dst;
                        segs[4 * count + 2] = a;
                        segs[4 * count + 3] = b;
                    }
                    ++count;
                }
            }
            p = q;
        }
    }
    return count;
}

// mode: 0 = choose (halo unless a rank would receive > 60 % of an all-gather), 1 = all-gather,
// 2 = halo.  `overlap` = 0 disables the send-first / interior-while-travelling split.
// info[0] = mode chosen (1 all-gather, 2 halo), info[1] = elements this rank receives per sweep,
// info[2] = elements it sends, info[3] = number of send ranges, info[4] = interior ranges.
int pi_exchange_plan(pi_handle* h, const uint8_t* term, int64_t per, int mode, int overlap,
                     int64_t* info, void* stream) {
    if (need_comm(h)) return 1;
    if (per < 1) return fail("bad argument");
    pi::DeviceGuard guard(h->device);
    hipStream_t st = (hipStream_t)stream;
    pi::Comm* c = h->comm;
    // Units the reach is measured in: rows (i0, i1) where the grid has them (2-4x fewer values to
    // exchange than whole planes of dimension 0), planes otherwise; PI_MI355_REACH_DEPTH overrides.
    int depth = pi_reach_depth_max(h);
    if (const char* e = std::getenv("PI_MI355_REACH_DEPTH")) depth = std::max(1, std::min(depth, std::atoi(e)));
    int64_t g0 = 1;
    for (int d = 0; d < depth; ++d) g0 *= h->shape[d];
    const int64_t n = h->n_states, stride0 = n / g0;
    if (per * c->world < n) return fail("per * world < n_states");
    std::unique_ptr<pi::ShardPlan> plan(new pi::ShardPlan);
    plan->per = per;
    plan->depth = depth;
    plan->s_begin = std::min((int64_t)c->rank * per, n);
    plan->s_end = std::min(plan->s_begin + per, n);
    if (mode != 1) {
        // reach bitmap of this shard -> all ranks
        std::vector<uint8_t> reach;
        if (gather_reach(c, g0, [&](uint32_t* d_bits) { return pi_reach_units(h, term, plan->s_begin, plan->s_end, depth, d_bits, stream); },
                         st, reach, "exchange plan"))
            return 1;
        const int64_t count = pi_plan_segments(c->world, g0, stride0, n, per, reach.data(), nullptr, 0);
        std::vector<int64_t> segs((size_t)count * 4);
        pi_plan_segments(c->world, g0, stride0, n, per, reach.data(), segs.data(), count);
        std::vector<int64_t> recv(c->world, 0);
        for (int64_t i = 0; i < count; ++i) recv[segs[4 * i + 1]] += segs[4 * i + 3] - segs[4 * i + 2];
        const int64_t full = per * (c->world - 1);
        const int64_t worst = *std::max_element(recv.begin(), recv.end());
        plan->halo = mode == 2 || (double)worst <= 0.6 * (double)full;
        if (plan->halo) {
            for (int64_t i = 0; i < count; ++i) {
                pi::ShardPlan::Seg s = {(int)segs[4 * i], (int)segs[4 * i + 1], segs[4 * i + 2], segs[4 * i + 3]};
                if (s.src == c->rank) { plan->segs.push_back(s); plan->send_elems += s.b - s.a; }
                else if (s.dst == c->rank) { plan->segs.push_back(s); plan->recv_elems += s.b - s.a; }
            }
        }
    }
    if (plan->halo && overlap) {
        // sub-ranges of this shard that peers wait for (swept first), and the rest
        std::vector<std::pair<int64_t, int64_t>> cuts;
        for (const auto& s : plan->segs)
            if (s.src == c->rank) cuts.push_back({s.a, s.b});
        std::sort(cuts.begin(), cuts.end());
        // What is swept first may be coarser than what is sent: pieces less than two planes of
        // dimension 0 (at least 2^16 states) apart become ONE launch, so a shard is still swept
        // in about three launches (low boundary planes, high boundary planes, interior) although
        // the segments that travel are single rows.
        // The gap is bounded by an eighth of the shard: with shards of ~10 planes (C4 at 8 ranks) two
        // planes would merge every send range into the whole shard and leave nothing to sweep while
        // the halo travels (profiles/r02/halo_granularity.txt: "1 / 0").
        const int64_t gap = std::min(std::max<int64_t>(int64_t(1) << 16, 2 * (n / h->shape[0])),
                                     std::max<int64_t>((plan->s_end - plan->s_begin) / 8, 1));
        for (const auto& r : cuts) {
            if (!plan->send_ranges.empty() && r.first <= plan->send_ranges.back().second + gap)
                plan->send_ranges.back().second = std::max(plan->send_ranges.back().second, r.second);
            else
                plan->send_ranges.push_back(r);
        }
        // a sliver between the shard's edge and its first / last send range joins that range
        if (!plan->send_ranges.empty()) {
            if (plan->send_ranges.front().first - plan->s_begin < gap) plan->send_ranges.front().first = plan->s_begin;
            if (plan->s_end - plan->send_ranges.back().second < gap) plan->send_ranges.back().second = plan->s_end;
        }
        int64_t pos = plan->s_begin;
        for (const auto& r : plan->send_ranges) {
            if (r.first > pos) plan->interior.push_back({pos, r.first});
            pos = std::max(pos, r.second);
        }
        if (pos < plan->s_end) plan->interior.push_back({pos, plan->s_end});
        // ---- what a transport that can deliver from inside the sweep (can_push) adds to the plan --------------------
        const char* fused_env = std::getenv("PI_MI355_P2P_FUSED");
        bool fused_wanted = c->can_push() && !(fused_env && std::atoi(fused_env) == 0) && pi::ensure_push_module(h) == 0;
        if (c->can_push()) {
            // every rank must take the fused branches below or none: a rank whose second module did not build (a failed
            // hipRTC compile, a corrupt cache entry) would otherwise leave its peers waiting in the pair probe until the
            // communicator's time limit.  One scalar sum: fused only if it is wanted everywhere.
            uint32_t* d_votes = nullptr;
            PI_HIP(hipMalloc((void**)&d_votes, sizeof(uint32_t)));
            const uint32_t mine = fused_wanted ? 1u : 0u;
            uint32_t all = 0;
            hipError_t e = hipMemcpyAsync(d_votes, &mine, sizeof mine, hipMemcpyHostToDevice, st);
            int rc = e == hipSuccess ? c->allreduce_sum_u32(d_votes, st) : 1;
            if (!rc && hipMemcpyAsync(&all, d_votes, sizeof all, hipMemcpyDeviceToHost, st) != hipSuccess) rc = 1;
            if (!rc && hipStreamSynchronize(st) != hipSuccess) rc = 1;
            (void)hipFree(d_votes);
            if (rc) return fail("exchange plan: the ranks could not agree on the fused exchange");
            fused_wanted = all == (uint32_t)c->world;
        }
        // Pair refinement.  With the velocity that moves coordinate 0 somewhere else than in memory dimension 1 (the fast
        // single-GPU orders put it along the lanes) the rows (i0, i1) above are all reachable and the segments are whole
        // bands of planes; the fused exchange delivers per STATE, so the destination masks below are cut down to the
        // pairs (i_0, i_v) every peer really reads (pi_reach_pairs_kernel) — the triangle again, in any memory order.
        // The segments stay as they are: the unfused exchanges (a batch's first sweep on grids with terminal states,
        // value iteration, all-gathers) travel whole.  Collective: every rank takes this branch or none (same grid,
        // same order, same transport, same environment).
        std::vector<uint8_t> need2;
        int64_t units = 0, gv = 1, stv = 1;
        const int64_t st0 = n / h->shape[0];
        {
            const char* e = std::getenv("PI_MI355_PAIR_REACH");
            const int v = h->D >= 2 ? h->mem_of_user[1] : 0;
            if (fused_wanted && pi::pairs_possible(h) && v != 1 && !(e && std::atoi(e) == 0)) {
                gv = h->shape[v];
                for (int d = v + 1; d < h->D; ++d) stv *= h->shape[d];
                units = (int64_t)h->shape[0] * gv;
                if (gather_reach(c, units, [&](uint32_t* d_bits) { return pi::reach_pairs(h, term, plan->s_begin, plan->s_end, d_bits, st); },
                                 st, need2, "exchange plan (pairs)"))
                    return 1;
                plan->pair_exact = true;
            }
        }
        auto wanted_by = [&](int dst, int64_t q) {           // does rank dst read state q at all (pair level)?
            return need2.empty() || need2[(size_t)dst * (size_t)units + (size_t)((q / st0) * gv + (q / stv) % gv)] != 0;
        };
        // bit of a receiver in the destination masks: the j-th distinct one, at most 8 (more: keep the copy kernel)
        auto peer_bit = [&](int dst, uint8_t* bit) {
            auto it = std::find(plan->push_peers.begin(), plan->push_peers.end(), dst);
            if (it == plan->push_peers.end()) {
                if (plan->push_peers.size() == 8) return false;
                plan->push_peers.push_back(dst);
                it = plan->push_peers.end() - 1;
            }
            *bit = (uint8_t)(1u << (it - plan->push_peers.begin()));
            return true;
        };
        auto note_sender = [&](int src) {
            if (std::find(plan->push_senders.begin(), plan->push_senders.end(), src) == plan->push_senders.end())
                plan->push_senders.push_back(src);
        };
        auto upload_lists = [&](const std::vector<int32_t>& first, const std::vector<int32_t>& inner) -> int {
            plan->n_first = (int64_t)first.size();
            plan->n_inner = (int64_t)inner.size();
            PI_HIP(hipMalloc((void**)&plan->d_first, std::max<size_t>(first.size(), 1) * sizeof(int32_t)));
            PI_HIP(hipMalloc((void**)&plan->d_inner, std::max<size_t>(inner.size(), 1) * sizeof(int32_t)));
            PI_HIP(hipMemcpyAsync(plan->d_first, first.data(), first.size() * sizeof(int32_t), hipMemcpyHostToDevice, st));
            PI_HIP(hipMemcpyAsync(plan->d_inner, inner.data(), inner.size() * sizeof(int32_t), hipMemcpyHostToDevice, st));
            PI_HIP(hipStreamSynchronize(st));
            plan->row_exact = true;
            return 0;
        };
        int want = -1;
        if (const char* e = std::getenv("PI_MI355_ROW_EXACT")) want = std::atoi(e) != 0 ? 1 : 0;
        if (plan->pair_exact && term == nullptr && !cuts.empty() && want != 0) {
            // State-exact lists (grids without terminal states): swept first = the states of this shard some peer reads,
            // each with the mask of those peers; the rest is the interior.  Only the fused exchange can deliver them.
            const int64_t len = plan->s_end - plan->s_begin;
            std::vector<uint8_t> mark((size_t)len, 0);
            bool fits = true;
            for (const auto& sg : plan->segs) {
                if (sg.dst == c->rank) { note_sender(sg.src); continue; }
                uint8_t bit = 0;
                if (!peer_bit(sg.dst, &bit)) { fits = false; break; }
                for (int64_t q = sg.a; q < sg.b; ++q)
                    if (wanted_by(sg.dst, q)) mark[(size_t)(q - plan->s_begin)] |= bit;
            }
            if (fits) {
                std::vector<int32_t> first, inner;
                std::vector<uint8_t> dest;
                for (int64_t q = plan->s_begin; q < plan->s_end; ++q) {
                    const uint8_t m = mark[(size_t)(q - plan->s_begin)];
                    if (m) { first.push_back((int32_t)q); dest.push_back(m); }
                    else inner.push_back((int32_t)q);
                }
                if (!first.empty() && (int64_t)first.size() < len) {
                    if (upload_lists(first, inner)) return 1;
                    drop_deliveries_for_tests(dest);
                    PI_HIP(hipMalloc((void**)&plan->d_first_dest, dest.size()));
                    PI_HIP(hipMemcpy(plan->d_first_dest, dest.data(), dest.size(), hipMemcpyHostToDevice));
                    plan->push_ok = true;
                    plan->fused_send_elems = 0;
                    for (uint8_t m : dest) plan->fused_send_elems += __builtin_popcount(m);
                }
            }
            if (!plan->push_ok) { plan->push_peers.clear(); plan->push_senders.clear(); }
        }
        // Row-exact alternative: the union of the rows that travel, nothing merged across gaps.  Taken when the grid
        // has no terminal states (the list kernel visits listed states only and copies nothing) and it moves at least a
        // quarter of the coarse swept-first states into the interior; PI_MI355_ROW_EXACT=0 / 1 forces it off / on.
        if (!plan->row_exact) {
            std::vector<std::pair<int64_t, int64_t>> exact;
            for (const auto& r : cuts) {
                if (!exact.empty() && r.first <= exact.back().second)
                    exact.back().second = std::max(exact.back().second, r.second);
                else
                    exact.push_back(r);
            }
            int64_t exact_states = 0, coarse_states = 0;
            for (const auto& r : exact) exact_states += r.second - r.first;
            for (const auto& r : plan->send_ranges) coarse_states += r.second - r.first;
            const bool possible = term == nullptr && !exact.empty() && exact_states < plan->s_end - plan->s_begin;
            const bool pays = exact.size() > plan->send_ranges.size() && 4 * exact_states <= 3 * coarse_states;
            if (possible && (want == 1 || (want < 0 && pays))) {
                std::vector<int32_t> first, inner;
                first.reserve((size_t)exact_states);
                inner.reserve((size_t)(plan->s_end - plan->s_begin - exact_states));
                int64_t at = plan->s_begin;
                for (const auto& r : exact) {
                    if (r.first > at) plan->inner_exact.push_back({at, r.first});
                    for (int64_t q = at; q < r.first; ++q) inner.push_back((int32_t)q);
                    for (int64_t q = r.first; q < r.second; ++q) first.push_back((int32_t)q);
                    at = r.second;
                }
                if (at < plan->s_end) plan->inner_exact.push_back({at, plan->s_end});
                for (int64_t q = at; q < plan->s_end; ++q) inner.push_back((int32_t)q);
                plan->first_exact = exact;
                if (upload_lists(first, inner)) return 1;
                // who reads which listed row: one byte per entry
                std::vector<uint8_t> dest(first.size(), 0);
                std::vector<int64_t> start(exact.size() + 1, 0);
                for (size_t i = 0; i < exact.size(); ++i) start[i + 1] = start[i] + (exact[i].second - exact[i].first);
                bool fits = fused_wanted;
                for (const auto& sg : plan->segs) {
                    if (!fits) break;
                    if (sg.dst == c->rank) { note_sender(sg.src); continue; }
                    uint8_t bit = 0;
                    if (!peer_bit(sg.dst, &bit)) { fits = false; break; }
                    // the range of `exact` that holds [a, b): the last one that starts at or before a
                    size_t i = (size_t)(std::upper_bound(exact.begin(), exact.end(), std::make_pair(sg.a, INT64_MAX)) - exact.begin()) - 1;
                    if (i >= exact.size() || sg.a < exact[i].first || sg.b > exact[i].second) { fits = false; break; }
                    for (int64_t q = sg.a; q < sg.b; ++q) dest[(size_t)(start[i] + (q - exact[i].first))] |= bit;
                }
                if (fits && !dest.empty()) {
                    drop_deliveries_for_tests(dest);
                    PI_HIP(hipMalloc((void**)&plan->d_first_dest, dest.size()));
                    PI_HIP(hipMemcpy(plan->d_first_dest, dest.data(), dest.size(), hipMemcpyHostToDevice));
                    plan->push_ok = true;
                } else {
                    plan->push_peers.clear();
                    plan->push_senders.clear();
                }
            }
        }
        if (!plan->row_exact && fused_wanted && pi::live_usable(h, term, plan->s_begin, plan->s_end)) {
            // later sweeps of a batch visit the shard's live states only: who reads which of them
            int64_t base = 0, total = 0;
            pi::live_span(h, plan->s_begin, plan->s_end, &base, &total);
            std::vector<uint8_t> dest((size_t)total, 0);
            std::vector<int32_t> states;
            bool fits = total > 0;
            for (const auto& sg : plan->segs) {
                if (!fits) break;
                if (sg.dst == c->rank) { note_sender(sg.src); continue; }
                uint8_t bit = 0;
                if (!peer_bit(sg.dst, &bit)) { fits = false; break; }
                int64_t first = 0, count = 0;
                pi::live_span(h, sg.a, sg.b, &first, &count);
                if (need2.empty()) {
                    for (int64_t k = first - base; k < first - base + count; ++k) dest[(size_t)k] |= bit;
                } else {
                    states.clear();
                    pi::live_states(h, sg.a, sg.b, states);              // the listed states of [a, b), ascending
                    for (int64_t k = 0; k < count; ++k)
                        if (wanted_by(sg.dst, states[(size_t)k])) dest[(size_t)(first - base + k)] |= bit;
                }
            }
            if (fits) {
                plan->live_list = h->d_live;
                plan->live_list_count = h->live_count;
                plan->live_push_ok = true;
                plan->fused_send_elems = 0;
                for (uint8_t m : dest) plan->fused_send_elems += __builtin_popcount(m);
                // State-exact lists for these sweeps too: swept first = the live states some peer reads (with their
                // masks), interior = the shard's other live states — instead of the coarse ranges, which in the orders
                // that keep the coupling velocity off memory dimension 1 are whole bands of planes.
                std::vector<int32_t> every, first, inner;
                std::vector<uint8_t> dfirst;
                pi::live_states(h, plan->s_begin, plan->s_end, every);
                if ((int64_t)every.size() == total) {
                    for (int64_t k = 0; k < total; ++k) {
                        if (dest[(size_t)k]) { first.push_back(every[(size_t)k]); dfirst.push_back(dest[(size_t)k]); }
                        else inner.push_back(every[(size_t)k]);
                    }
                }
                if (!first.empty() && !inner.empty()) {
                    drop_deliveries_for_tests(dfirst);
                    plan->n_first = (int64_t)first.size();
                    plan->n_inner = (int64_t)inner.size();
                    PI_HIP(hipMalloc((void**)&plan->d_first, first.size() * sizeof(int32_t)));
                    PI_HIP(hipMalloc((void**)&plan->d_inner, inner.size() * sizeof(int32_t)));
                    PI_HIP(hipMalloc((void**)&plan->d_first_dest, dfirst.size()));
                    PI_HIP(hipMemcpy(plan->d_first, first.data(), first.size() * sizeof(int32_t), hipMemcpyHostToDevice));
                    PI_HIP(hipMemcpy(plan->d_inner, inner.data(), inner.size() * sizeof(int32_t), hipMemcpyHostToDevice));
                    PI_HIP(hipMemcpy(plan->d_first_dest, dfirst.data(), dfirst.size(), hipMemcpyHostToDevice));
                    plan->live_exact = true;
                } else {
                    PI_HIP(hipMalloc((void**)&plan->d_live_dest, dest.size()));
                    PI_HIP(hipMemcpy(plan->d_live_dest, dest.data(), dest.size(), hipMemcpyHostToDevice));
                    plan->live_dest_base = base;
                }
            } else {
                plan->push_peers.clear();
                plan->push_senders.clear();
            }
        }
        PI_HIP(hipStreamCreateWithFlags(&plan->comm_stream, hipStreamNonBlocking));
        PI_HIP(hipEventCreateWithFlags(&plan->ev_ready, hipEventDisableTiming));
        PI_HIP(hipEventCreateWithFlags(&plan->ev_done, hipEventDisableTiming));
    }
    if (!plan->halo) {
        plan->recv_elems = per * (c->world - 1);
        plan->send_elems = per;
    }
    if (info) {
        info[0] = plan->halo ? 2 : 1;
        info[1] = plan->recv_elems;
        info[2] = plan->send_elems;
        info[3] = (int64_t)plan->send_ranges.size();
        info[4] = (int64_t)plan->interior.size();
    }
    pi::drop_plan(h);
    h->plan = plan.release();
    return 0;
}

// The launch ranges of the current plan: up to `cap` triples {kind, begin, end} — kind 0 = swept
// first (peers wait for rows inside it), kind 1 = interior (swept while the halo travels); a plan
// without overlap (all-gather, or overlap disabled) reports the whole shard as one kind-0 range.
int64_t pi_plan_ranges(pi_handle* h, int64_t* ranges, int64_t cap) {
    if (need_plan(h)) return -1;
    const pi::ShardPlan* p = h->plan;
    std::vector<std::array<int64_t, 3>> all;
    if (p->halo && p->comm_stream != nullptr && p->row_exact && !p->first_exact.empty()) {
        for (const auto& r : p->first_exact) all.push_back({0, r.first, r.second});
        for (const auto& r : p->inner_exact) all.push_back({1, r.first, r.second});
    } else if (p->halo && p->comm_stream != nullptr) {
        for (const auto& r : p->send_ranges) all.push_back({0, r.first, r.second});
        for (const auto& r : p->interior) all.push_back({1, r.first, r.second});
    } else if (p->s_end > p->s_begin) {
        all.push_back({0, p->s_begin, p->s_end});
    }
    for (size_t i = 0; ranges && i < all.size() && (int64_t)i < cap; ++i)
        for (int k = 0; k < 3; ++k) ranges[3 * i + k] = all[i][k];
    return (int64_t)all.size();
}

// One part of a sharded evaluation sweep WITHOUT the exchange, exactly as pi_eval_sweeps_sharded launches it:
// part 0 = what peers wait for (swept first), part 1 = the interior.  For measurements and tests: both parts
// together are one sweep of the shard.  A plan without overlap has everything in part 0.
int pi_eval_sweep_part(pi_handle* h, const float* V, float* Vnew, const int32_t* policy, const uint8_t* term, int part,
                       float gamma, void* stream) {
    if (need_plan(h)) return 1;
    if (!V || !Vnew || !policy || V == Vnew) return fail("bad device pointer");
    if (part != 0 && part != 1) return fail("part must be 0 (swept first) or 1 (interior)");
    pi::DeviceGuard guard(h->device);
    hipStream_t st = (hipStream_t)stream;
    const pi::ShardPlan* p = h->plan;
    if (!(p->halo && p->comm_stream != nullptr))
        return part == 0 ? pi::launch_eval(h, V, Vnew, policy, term, p->s_begin, p->s_end, gamma, false, st) : 0;
    if (p->row_exact && term == nullptr)
        return pi::launch_eval_live(h, V, Vnew, policy, 0, part == 0 ? p->n_first : p->n_inner, gamma, false, st,
                                    part == 0 ? p->d_first : p->d_inner);
    for (const auto& r : (part == 0 ? p->send_ranges : p->interior))
        if (pi::launch_eval(h, V, Vnew, policy, term, r.first, r.second, gamma, false, st)) return 1;
    return 0;
}

// Make this rank's freshly written shard of `V_full` visible where the other ranks read it.
int pi_exchange_V(pi_handle* h, float* V_full, void* stream) {
    if (need_plan(h)) return 1;
    pi::DeviceGuard guard(h->device);
    hipStream_t st = (hipStream_t)stream;
    if (!h->plan->halo) return h->comm->allgather(V_full, (size_t)h->plan->per * sizeof(float), st);
    return post_exchange(h, V_full, st);
}

// n_sweeps evaluation sweeps of this rank's shard, ping-ponging between Va and Vb exactly like
// pi_eval_sweeps, with the exchange after every sweep; d_delta (nullable) receives the residual of
// the LAST sweep, already reduced (MAX) over all ranks.  Nothing returns to the host in between.
int pi_eval_sweeps_sharded(pi_handle* h, float* Va, float* Vb, const int32_t* policy,
                           const uint8_t* term, float gamma, int n_sweeps, float* d_delta,
                           void* stream) {
    if (need_plan(h)) return 1;
    if (n_sweeps < 0) return fail("n_sweeps < 0");
    if (!Va || !Vb || !policy) return fail("null device pointer");
    if (Va == Vb) return fail("Va and Vb must be different buffers (Jacobi sweep)");
    if (n_sweeps == 0) return 0;
    pi::DeviceGuard guard(h->device);
    hipStream_t st = (hipStream_t)stream;
    pi::ShardPlan* p = h->plan;
    const bool overlap = p->halo && p->comm_stream != nullptr;
    // One launch range of one sweep: the later sweeps of a batch (k >= 1) visit only the range's live states when
    // the handle holds a list for this mask (pi_prepare_mask) — the first one copies the terminal values.
    const bool live = pi::live_usable(h, term, p->s_begin, p->s_end);
    auto sweep = [&](const float* src, float* dst, int64_t a, int64_t b, bool want, int k) -> int {
        if (k >= 1 && live) {
            int64_t first = 0, count = 0;
            pi::live_span(h, a, b, &first, &count);
            return pi::launch_eval_live(h, src, dst, policy, first, count, gamma, want, st);
        }
        return pi::launch_eval(h, src, dst, policy, term, a, b, gamma, want, st, k >= 1);
    };
    auto batch = [&]() -> int {
        for (int k = 0; k < n_sweeps; ++k) {
            const float* src = (k & 1) ? Vb : Va;
            float* dst = (k & 1) ? Va : Vb;
            const bool want = k == n_sweeps - 1 && d_delta != nullptr;
            if (overlap && p->row_exact && term == nullptr && p->push_ok) {
                // fused exchange, everything on this stream: the swept-first kernel stores its rows into the peers'
                // buffers itself, one-wave kernels hand-shake in front of it and behind it
                pi::Comm* c = h->comm;
                float* const* table = nullptr;
                if (c->push_begin(p->push_peers, p->push_senders, dst, &table, st)) return 1;
                if (pi::launch_eval_push(h, src, dst, policy, p->d_first, p->d_first_dest, table, (int)p->push_peers.size(),
                                         p->n_first, gamma, want, st)) return 1;
                if (c->push_signal(st)) return 1;
                if (pi::launch_eval_live(h, src, dst, policy, 0, p->n_inner, gamma, want, st, p->d_inner)) return 1;
                if (k + 1 < n_sweeps ? c->push_wait_deferred(st) : c->push_wait(st)) return 1;     // the next sweep is fused too
            } else if (overlap && k >= 1 && live && p->live_push_ok && p->live_list == h->d_live &&
                       p->live_list_count == h->live_count) {
                // the same fused exchange for the later sweeps of a batch on a grid with terminal states: the live
                // states of the ranges peers wait for are swept by the push kernel, which delivers them (terminal
                // states' values never change: every rank's buffers hold them since the batch's first sweep)
                pi::Comm* c = h->comm;
                float* const* table = nullptr;
                if (c->push_begin(p->push_peers, p->push_senders, dst, &table, st)) return 1;
                if (p->live_exact) {
                    if (pi::launch_eval_push(h, src, dst, policy, p->d_first, p->d_first_dest, table, (int)p->push_peers.size(),
                                             p->n_first, gamma, want, st)) return 1;
                    if (c->push_signal(st)) return 1;
                    if (pi::launch_eval_live(h, src, dst, policy, 0, p->n_inner, gamma, want, st, p->d_inner)) return 1;
                } else {
                    for (const auto& r : p->send_ranges) {
                        int64_t first = 0, count = 0;
                        pi::live_span(h, r.first, r.second, &first, &count);
                        if (pi::launch_eval_push(h, src, dst, policy, h->d_live + first, p->d_live_dest + (first - p->live_dest_base),
                                                 table, (int)p->push_peers.size(), count, gamma, want, st)) return 1;
                    }
                    if (c->push_signal(st)) return 1;
                    for (const auto& r : p->interior)
                        if (sweep(src, dst, r.first, r.second, want, k)) return 1;
                }
                if (k + 1 < n_sweeps ? c->push_wait_deferred(st) : c->push_wait(st)) return 1;     // k + 1 >= 1: fused too
            } else if (overlap) {
                if (p->row_exact && term == nullptr) {
                    if (pi::launch_eval_live(h, src, dst, policy, 0, p->n_first, gamma, want, st, p->d_first)) return 1;
                } else {
                    for (const auto& r : p->send_ranges)
                        if (sweep(src, dst, r.first, r.second, want, k)) return 1;
                }
                PI_HIP(hipEventRecord(p->ev_ready, st));
                PI_HIP(hipStreamWaitEvent(p->comm_stream, p->ev_ready, 0));
                if (post_exchange(h, dst, p->comm_stream)) return 1;
                PI_HIP(hipEventRecord(p->ev_done, p->comm_stream));
                if (p->row_exact && term == nullptr) {
                    if (pi::launch_eval_live(h, src, dst, policy, 0, p->n_inner, gamma, want, st, p->d_inner)) return 1;
                } else {
                    for (const auto& r : p->interior)
                        if (sweep(src, dst, r.first, r.second, want, k)) return 1;
                }
                PI_HIP(hipStreamWaitEvent(st, p->ev_done, 0));
            } else {
                if (sweep(src, dst, p->s_begin, p->s_end, want, k)) return 1;
                if (pi_exchange_V(h, dst, stream)) return 1;
            }
        }
        if (d_delta) {
            if (p->s_end == p->s_begin) PI_HIP(hipMemsetAsync(d_delta, 0, sizeof(float), st));
            else if (pi::finalize(h, d_delta, nullptr, st)) return 1;
            return h->comm->allreduce_max_f32(d_delta, st);
        }
        return 0;
    };
    if (batch() == 0) return 0;
    // Failure half-way through the batch: leave a DEFINED state behind.  Drain both streams (a
    // launch or an event may be pending on either), clear the residual slots a finished sweep may
    // have filled, and retire the plan — the peers are at an unknown sweep, so the next sharded call
    // must be preceded by a new communicator and a collective pi_exchange_plan (need_plan says how).  The error of the
    // failing call is kept.
    const std::string why = pi::last_error();
    h->comm->give_up();                                                      // in-process peers stop waiting at once
    if (p->comm_stream) (void)hipStreamSynchronize(p->comm_stream);
    (void)hipStreamSynchronize(st);
    (void)hipMemsetAsync(h->d_slots, 0, 2 * pi::kSlots * sizeof(unsigned int), st);
    (void)hipGetLastError();
    p->broken = true;
    return fail("pi_eval_sweeps_sharded abandoned (streams drained, exchange plan retired): " + why);
}

// Greedy improvement of this rank's shard; d_changed (nullable) = entries changed, summed over ranks.
int pi_improve_sweep_sharded(pi_handle* h, const float* V, int32_t* policy, const uint8_t* term,
                             float gamma, uint32_t* d_changed, void* stream) {
    if (need_plan(h)) return 1;
    if (pi_improve_sweep(h, V, policy, term, h->plan->s_begin, h->plan->s_end, gamma, d_changed, stream)) return 1;
    if (d_changed) {
        pi::DeviceGuard guard(h->device);
        return h->comm->allreduce_sum_u32(d_changed, (hipStream_t)stream);
    }
    return 0;
}

}  // extern "C"
