// pi_p2p.cpp — third transport of libpi_mi355.so: peer-to-peer stores into IPC-mapped buffers, no RCCL on the data
// path.  One process per GPU, as with RCCL.
//
// Why: at C4 @ 8 a rank computes ~68 us per sweep and the grouped ncclSend / ncclRecv of the halo exchange costs
// ~30 us of group latency on top of the transfer.  Here the sender's GPU stores the rows straight into the receiver's
// V' buffer (hipIpcOpenMemHandle mapping; the stores travel over xGMI) and the two sides hand-shake through 32-bit
// counters in a small uncached flag page that every rank owns and every peer maps:
//
//     receiver, at group_end on its stream:   ack[sender's page][me]   = k      "my receive number k is posted: the
//                                                                                 region you will store into is free"
//     sender, ONE kernel on its stream:       wait ack[my page][receiver] >= k;  copy all segments of the group into the
//                                             peers' buffers;  system-scope fence;  data[receiver's page][me] = k
//     receiver:                               wait data[my page][sender] >= k    (then its next kernel starts with the
//                                                                                 usual acquire and sees the rows)
//
// which is the rendezvous an RCCL send / recv pair performs, in three small launches per group (signal + wait, copy, wait)
// and two one-way flag writes of latency.  Scalar reductions (residual MAX, change-count SUM) go through the same pages
// (pi_p2p_reduce_kernel), so a run needs no RCCL communicator at all.  Addresses are SYMMETRIC: a rank sends from
// offset o of a registered buffer into offset o of the peer's registered buffer of the same index — what the in-place
// halo exchange and the in-place all-gathers of this library do.
//
// Fused form (row-exact plans; push_begin / push_signal / push_wait below, csrc/pi_push_kernels.hip): the swept-first
// kernel of a sweep stores its rows into the peers itself, the copy kernel and the second stream disappear, and the same
// counters are raised / awaited by one-wave kernels on the sweep's own stream.
//
// Bootstrap is the caller's (like the 128-byte RCCL id): pi_p2p_describe fills a 512-byte descriptor (process id,
// device, IPC handles of the registered buffers and of the flag page), the caller all-gathers the descriptors any way
// it likes (torch.distributed.all_gather_object in transport.py) and hands all of them to pi_comm_init_p2p.
//
// Every device-side wait is bounded (PI_MI355_COMM_TIMEOUT): a peer that never arrives sets this rank's error word,
// later waits return at once, and the next reduction / all-gather reports it on the host.

#include "pi_internal.h"

#include <unistd.h>

#include <algorithm>
#include <cstring>
#include <map>

using pi::fail;

__asm__(
    ".section .rodata\n"
    ".global pi_embedded_p2p\n"
    "pi_embedded_p2p:\n"
    ".incbin \"" PI_CSRC_DIR "/pi_p2p_kernels.hip\"\n"
    ".byte 0\n"
    ".previous\n");
extern "C" const char pi_embedded_p2p[];

namespace {

constexpr int kMaxRanks = 16;                 // PI_P2P_MAX
constexpr int kMaxBufs = 4;
constexpr uint32_t kMagic = 0x50325031u;      // "P2P1"
constexpr size_t kDescBytes = 512;
// flag page: u32 ack[16] | u32 data[16] | u32 error, block counter | ... | u64 red[2][16] at 256 | scratch at 4096
constexpr size_t kOffAck = 0, kOffData = 64, kOffError = 128, kOffCounter = 132, kOffRed = 256, kOffScratch = 4096;
constexpr size_t kScratchBytes = size_t(1) << 20;
constexpr size_t kPageBytes = kOffScratch + kScratchBytes;

struct DescBuf {
    uint64_t ptr, bytes, base_offset;          // the registered range; its offset inside the allocation `ipc` names
    hipIpcMemHandle_t ipc;
};
struct Desc {
    uint32_t magic, version;
    int32_t rank, world, device, n_bufs;
    int64_t pid;
    DescBuf buf[kMaxBufs + 1];                 // [0] = this rank's flag page, [1 ..] = the caller's buffers
};
static_assert(sizeof(hipIpcMemHandle_t) == 64, "HIP IPC handle is 64 bytes");
static_assert(sizeof(Desc) <= kDescBytes, "descriptor must fit its 512-byte slot");

// mirrors of the kernel argument structs (pi_p2p_kernels.hip)
struct Flags {
    uint32_t* ptr[kMaxRanks];
    uint32_t value[kMaxRanks];
    int n;
};
struct Seg {
    float* dst;
    const float* src;
    long long first;
};
struct Red {
    unsigned long long* peer_slot[kMaxRanks];
    const unsigned long long* mine;
    int world;
    unsigned int epoch;
    int op;
};

}  // namespace

namespace pi {

struct P2pPending {
    char* page = nullptr;
    Desc desc = {};
    ~P2pPending() {
        if (page) (void)hipFree(page);
    }
};

void drop_p2p_pending(pi_handle* h) {
    if (!h->p2p_pending) return;
    DeviceGuard guard(h->device);
    delete h->p2p_pending;
    h->p2p_pending = nullptr;
}

}  // namespace pi

namespace {

struct P2pComm : pi::Comm {
    int device = -1;
    hipModule_t module = nullptr;
    hipFunction_t f_sigwait = nullptr, f_push = nullptr, f_reduce = nullptr;
    char* page = nullptr;                                   // this rank's flag page (owned)
    struct Buf { char* base; size_t bytes; };
    std::vector<Buf> mine;                                  // [0] = scratch region of the page, [1 ..] = the caller's
    std::vector<std::vector<char*>> theirs;                 // [peer][buffer] -> address in THIS process
    std::vector<char*> peer_page;                           // [peer] (own page at [rank])
    std::vector<void*> opened;                              // what hipIpcOpenMemHandle returned (closed on teardown)
    std::vector<uint32_t> sent_n, recv_n;                   // message numbers per peer
    uint32_t red_epoch = 0;
    unsigned long long ticks = 0;
    std::string dead;
    struct Op { bool is_send; char* local; size_t bytes; int peer; };
    std::vector<Op> ops;
    struct Table { std::vector<Seg> key; Seg* d = nullptr; int vec4 = 0; };
    std::vector<Table> tables;                              // device copies of the segment lists seen so far
    // fused exchange: the receivers' addresses of a local buffer (device arrays of pointers), and the flags the two
    // one-wave kernels behind the push kernel will raise / wait for
    struct PeerTable { float* local; std::vector<int> peers; float** d = nullptr; };
    std::vector<PeerTable> peer_tables;
    Flags pending_done = {}, pending_data = {}, carry = {};

    ~P2pComm() override {
        pi::DeviceGuard guard(device);
        (void)hipDeviceSynchronize();
        for (auto& t : tables)
            if (t.d) (void)hipFree(t.d);
        for (auto& t : peer_tables)
            if (t.d) (void)hipFree(t.d);
        for (void* p : opened) (void)hipIpcCloseMemHandle(p);
        if (module) (void)hipModuleUnload(module);
        if (page) (void)hipFree(page);
    }
    const char* kind() const override { return "p2p"; }
    int alive() const { return dead.empty() ? 0 : fail(dead); }

    uint32_t* flag(char* pg, size_t off, int slot) const { return reinterpret_cast<uint32_t*>(pg + off) + slot; }
    uint32_t* error_word() const { return flag(page, kOffError, 0); }

    // (buffer index, offset) of a local address range, -1 when it is not inside a registered buffer
    int locate(const void* p, size_t bytes, size_t* off) const {
        const char* q = static_cast<const char*>(p);
        for (size_t b = 0; b < mine.size(); ++b)
            if (q >= mine[b].base && q + bytes <= mine[b].base + mine[b].bytes) {
                *off = (size_t)(q - mine[b].base);
                return (int)b;
            }
        return -1;
    }
    int launch(hipFunction_t f, unsigned grid, unsigned block, void** args, hipStream_t st) {
        PI_HIP(hipModuleLaunchKernel(f, grid, 1, 1, block, 1, 1, 0, st, args, nullptr));
        return 0;
    }

    int group_begin() override {
        if (alive()) return 1;
        ops.clear();
        return 0;
    }
    int send(const void* p, size_t bytes, int peer, hipStream_t) override {
        if (alive()) return 1;
        if (peer < 0 || peer >= world || peer == rank) return fail("p2p transport: bad peer");
        size_t off;
        if (locate(p, bytes, &off) < 0)
            return fail("p2p transport: send from a buffer that was not registered with pi_p2p_describe");
        if (bytes % 4 != 0 || off % 4 != 0) return fail("p2p transport: transfers are whole 32-bit words");
        ops.push_back({true, const_cast<char*>(static_cast<const char*>(p)), bytes, peer});
        return 0;
    }
    int recv(void* p, size_t bytes, int peer, hipStream_t) override {
        if (alive()) return 1;
        if (peer < 0 || peer >= world || peer == rank) return fail("p2p transport: bad peer");
        size_t off;
        if (locate(p, bytes, &off) < 0)
            return fail("p2p transport: receive into a buffer that was not registered with pi_p2p_describe");
        ops.push_back({false, static_cast<char*>(p), bytes, peer});
        return 0;
    }
    // The segment list of this group on the device (uploaded the first time a list is seen: the halo exchange has two,
    // one per Jacobi buffer).
    int table_for(const std::vector<Seg>& segs, int vec4, Seg** out) {
        for (auto& t : tables)
            if (t.vec4 == vec4 && t.key.size() == segs.size() &&
                std::memcmp(t.key.data(), segs.data(), segs.size() * sizeof(Seg)) == 0) {
                *out = t.d;
                return 0;
            }
        if (tables.size() >= 16) {                           // all-gathers of many different buffers: start over
            PI_HIP(hipDeviceSynchronize());
            for (auto& t : tables) (void)hipFree(t.d);
            tables.clear();
        }
        Table t;
        t.key = segs;
        t.vec4 = vec4;
        PI_HIP(hipMalloc((void**)&t.d, segs.size() * sizeof(Seg)));
        PI_HIP(hipMemcpy(t.d, segs.data(), segs.size() * sizeof(Seg), hipMemcpyHostToDevice));
        tables.push_back(t);
        *out = t.d;
        return 0;
    }
    int group_end(hipStream_t st) override {
        if (alive()) return 1;
        std::vector<int> senders, receivers;
        for (const Op& o : ops) {
            auto& v = o.is_send ? receivers : senders;
            if (std::find(v.begin(), v.end(), o.peer) == v.end()) v.push_back(o.peer);
        }
        uint32_t* err = error_word();
        Flags none = {};
        // 1. one wave: my receives of this group are posted — tell every sender its target region is free — and wait
        //    until the receivers of my sends have said the same (the copy kernel then starts only when its targets are
        //    free, instead of parking up to 1 024 workgroups on the CUs the interior sweep is using)
        Flags posted = {}, acks = {}, done = {};
        for (int p : senders) {
            posted.ptr[posted.n] = flag(peer_page[p], kOffAck, rank);
            posted.value[posted.n++] = ++recv_n[p];
        }
        for (int p : receivers) {
            const uint32_t k = ++sent_n[p];
            acks.ptr[acks.n] = flag(page, kOffAck, p);
            acks.value[acks.n++] = k;
            done.ptr[done.n] = flag(peer_page[p], kOffData, rank);
            done.value[done.n++] = k;
        }
        {
            void* args[] = {&posted, &acks, &ticks, &err};
            if (launch(f_sigwait, 1, 64, args, st)) return 1;
        }
        // 2. my sends: one kernel stores every segment into the peers' buffers and raises their data counters
        if (!receivers.empty()) {
            std::vector<Seg> segs;
            long long units = 0;
            int vec4 = 1;
            for (const Op& o : ops)
                if (o.is_send) {
                    size_t off = 0;
                    const int b = locate(o.local, o.bytes, &off);
                    char* dst = theirs[o.peer][b] + off;
                    if (((uintptr_t)dst | (uintptr_t)o.local | o.bytes) & 15u) vec4 = 0;
                    segs.push_back({reinterpret_cast<float*>(dst), reinterpret_cast<const float*>(o.local), (long long)o.bytes});
                }
            for (Seg& s : segs) {                            // `first` held the byte count so far: turn into a prefix of units
                const long long n = s.first / (vec4 ? 16 : 4);
                s.first = units;
                units += n;
            }
            segs.push_back({nullptr, nullptr, units});
            Seg* d_segs = nullptr;
            if (table_for(segs, vec4, &d_segs)) return 1;
            int n_segs = (int)segs.size() - 1;
            uint32_t* counter = flag(page, kOffCounter, 0);
            const unsigned grid = (unsigned)std::max<long long>(1, std::min<long long>((units + 1023) / 1024, 1024));
            void* args[] = {&d_segs, &n_segs, &vec4, &none, &done, &ticks, &err, &counter};
            if (launch(f_push, grid, 256, args, st)) return 1;
        }
        // 3. wait for the data of my receives
        if (!senders.empty()) {
            Flags data = {};
            for (int p : senders) {
                data.ptr[data.n] = flag(page, kOffData, p);
                data.value[data.n++] = recv_n[p];
            }
            void* args[] = {&none, &data, &ticks, &err};
            if (launch(f_sigwait, 1, 64, args, st)) return 1;
        }
        ops.clear();
        return 0;
    }
    bool can_push() const override { return true; }
    int push_begin(const std::vector<int>& receivers, const std::vector<int>& senders, float* local_dst,
                   float* const** table, hipStream_t st) override {
        if (alive()) return 1;
        size_t off = 0;
        const int b = locate(local_dst, 4, &off);
        if (b < 0) return fail("p2p transport: the buffer being swept into was not registered with pi_p2p_describe");
        PeerTable* hit = nullptr;
        for (auto& t : peer_tables)
            if (t.local == local_dst && t.peers == receivers) hit = &t;
        if (!hit) {
            PeerTable t;
            t.local = local_dst;
            t.peers = receivers;
            std::vector<float*> host;
            for (int p : receivers) {
                if (p < 0 || p >= world || p == rank) return fail("p2p transport: bad peer");
                host.push_back(reinterpret_cast<float*>(theirs[p][b] + off));
            }
            PI_HIP(hipMalloc((void**)&t.d, std::max<size_t>(host.size(), 1) * sizeof(float*)));
            PI_HIP(hipMemcpy(t.d, host.data(), host.size() * sizeof(float*), hipMemcpyHostToDevice));
            peer_tables.push_back(t);
            hit = &peer_tables.back();
        }
        *table = hit->d;
        Flags posted = {}, acks = {};
        pending_done = {};
        pending_data = {};
        for (int p : senders) {
            posted.ptr[posted.n] = flag(peer_page[p], kOffAck, rank);
            posted.value[posted.n++] = ++recv_n[p];
            pending_data.ptr[pending_data.n] = flag(page, kOffData, p);
            pending_data.value[pending_data.n++] = recv_n[p];
        }
        for (int p : receivers) {
            const uint32_t k = ++sent_n[p];
            acks.ptr[acks.n] = flag(page, kOffAck, p);
            acks.value[acks.n++] = k;
            pending_done.ptr[pending_done.n] = flag(peer_page[p], kOffData, rank);
            pending_done.value[pending_done.n++] = k;
        }
        // a push_wait the caller left pending (two fused sweeps back to back): its data counters are awaited by THIS
        // launch together with the acks — one one-wave kernel between two sweeps instead of two
        if (carry.n > 0 && carry.n + acks.n <= kMaxRanks) {
            for (int i = 0; i < carry.n; ++i) {
                acks.ptr[acks.n] = carry.ptr[i];
                acks.value[acks.n++] = carry.value[i];
            }
            carry = {};
        }
        uint32_t* err = error_word();
        if (carry.n > 0) {                                   // did not fit: wait for it on its own, first
            Flags none = {};
            void* wargs[] = {&none, &carry, &ticks, &err};
            if (launch(f_sigwait, 1, 64, wargs, st)) return 1;
            carry = {};
        }
        if (posted.n == 0 && acks.n == 0) return 0;
        void* args[] = {&posted, &acks, &ticks, &err};
        return launch(f_sigwait, 1, 64, args, st);
    }
    // The next call on this stream is the push_begin of another fused sweep: let IT wait for this sweep's data.
    int push_wait_deferred(hipStream_t) override {
        if (alive()) return 1;
        carry = pending_data;
        pending_data = {};
        return 0;
    }
    int push_signal(hipStream_t st) override {
        if (alive()) return 1;
        if (pending_done.n == 0) return 0;
        Flags none = {};
        uint32_t* err = error_word();
        void* args[] = {&pending_done, &none, &ticks, &err};
        return launch(f_sigwait, 1, 64, args, st);
    }
    int push_wait(hipStream_t st) override {
        if (alive()) return 1;
        if (pending_data.n == 0) return 0;
        Flags none = {};
        uint32_t* err = error_word();
        void* args[] = {&none, &pending_data, &ticks, &err};
        return launch(f_sigwait, 1, 64, args, st);
    }
    // Has a wait of this rank timed out?  Blocks on `st` (called where the host synchronises anyway).
    int health(hipStream_t st) {
        uint32_t bits = 0;
        PI_HIP(hipMemcpyAsync(&bits, error_word(), sizeof bits, hipMemcpyDeviceToHost, st));
        PI_HIP(hipStreamSynchronize(st));
        if (bits == 0u) return 0;
        char hex[16];
        std::snprintf(hex, sizeof hex, "%08x", bits);
        dead = "p2p transport: rank " + std::to_string(rank) + " of " + std::to_string(world) +
               " gave up waiting for a peer (error word 0x" + hex + ": bit i = the i-th peer of a group, or rank i of a "
               "reduction, did not arrive within PI_MI355_COMM_TIMEOUT); the communicator is spent";
        return fail(dead);
    }
    int allgather(void* full, size_t bytes, hipStream_t st) override {
        if (alive()) return 1;
        size_t off;
        char* base = static_cast<char*>(full);
        const bool staged = locate(full, bytes * (size_t)world, &off) < 0;
        if (staged) {                                        // not a registered buffer (the reach bitmaps of a plan)
            if (bytes * (size_t)world > kScratchBytes)
                return fail("p2p transport: all-gather of an unregistered buffer larger than the 1 MiB staging area");
            base = page + kOffScratch;
            PI_HIP(hipMemcpyAsync(base + (size_t)rank * bytes, static_cast<char*>(full) + (size_t)rank * bytes, bytes,
                                  hipMemcpyDeviceToDevice, st));
        }
        if (group_begin()) return 1;
        for (int p = 0; p < world; ++p)
            if (p != rank && send(base + (size_t)rank * bytes, bytes, p, st)) return 1;
        for (int p = 0; p < world; ++p)
            if (p != rank && recv(base + (size_t)p * bytes, bytes, p, st)) return 1;
        if (group_end(st)) return 1;
        if (staged) PI_HIP(hipMemcpyAsync(full, base, bytes * (size_t)world, hipMemcpyDeviceToDevice, st));
        return health(st);
    }
    int reduce(void* d, int op, hipStream_t st) {
        if (alive()) return 1;
        Red r = {};
        const unsigned int e = ++red_epoch;
        const size_t row = kOffRed + (size_t)(e & 1u) * kMaxRanks * sizeof(unsigned long long);
        for (int p = 0; p < world; ++p)
            r.peer_slot[p] = reinterpret_cast<unsigned long long*>(peer_page[p] + row) + rank;
        r.mine = reinterpret_cast<const unsigned long long*>(page + row);
        r.world = world;
        r.epoch = e;
        r.op = op;
        uint32_t* err = error_word();
        void* args[] = {&d, &r, &ticks, &err};
        if (launch(f_reduce, 1, 64, args, st)) return 1;
        return health(st);
    }
    // values >= 0 (residuals): their bit patterns order like unsigned integers
    int allreduce_max_f32(float* d, hipStream_t st) override { return reduce(d, 0, st); }
    int allreduce_sum_u32(uint32_t* d, hipStream_t st) override { return reduce(d, 1, st); }
};

}  // namespace

extern "C" {

int pi_p2p_describe(pi_handle* h, int rank, int world, const void* const* bufs, const int64_t* bytes, int n_bufs,
                    void* desc512) {
    if (!h || !desc512 || (n_bufs > 0 && (!bufs || !bytes))) return fail("null argument");
    if (h->device < 0) return fail("host-only handle cannot own a communicator");
    if (world < 2 || world > kMaxRanks || rank < 0 || rank >= world)
        return fail("pi_p2p_describe: 2 <= world <= 16, 0 <= rank < world");
    if (n_bufs < 0 || n_bufs > kMaxBufs) return fail("pi_p2p_describe: at most 4 buffers can be registered");
    pi::DeviceGuard guard(h->device);
    pi::drop_p2p_pending(h);
    std::unique_ptr<pi::P2pPending> pend(new pi::P2pPending);
    // the flag page: uncached device memory where the platform has it and can share it (flags are polled across GPUs),
    // else fine-grained, else plain device memory
    for (int attempt = 0; attempt < 3 && !pend->page; ++attempt) {
        void* p = nullptr;
        hipError_t e = attempt == 0   ? hipExtMallocWithFlags(&p, kPageBytes, hipDeviceMallocUncached)
                       : attempt == 1 ? hipExtMallocWithFlags(&p, kPageBytes, hipDeviceMallocFinegrained)
                                      : hipMalloc(&p, kPageBytes);
        hipIpcMemHandle_t probe;
        if (e == hipSuccess && hipIpcGetMemHandle(&probe, p) != hipSuccess) {
            (void)hipFree(p);
            e = hipErrorInvalidValue;
        }
        if (e == hipSuccess) pend->page = static_cast<char*>(p);
        else (void)hipGetLastError();
    }
    if (!pend->page) return fail("pi_p2p_describe: no shareable device memory for the flag page");
    PI_HIP(hipMemset(pend->page, 0, kPageBytes));
    PI_HIP(hipDeviceSynchronize());
    Desc& d = pend->desc;
    d.magic = kMagic;
    d.version = 1;
    d.rank = rank;
    d.world = world;
    d.device = h->device;
    d.n_bufs = n_bufs;
    d.pid = (int64_t)getpid();
    for (int b = 0; b <= n_bufs; ++b) {
        void* p = b == 0 ? (void*)pend->page : const_cast<void*>(bufs[b - 1]);
        const size_t len = b == 0 ? kPageBytes : (size_t)bytes[b - 1];
        if (!p || len == 0) return fail("pi_p2p_describe: null or empty buffer");
        hipDeviceptr_t base = nullptr;
        size_t alloc = 0;
        PI_HIP(hipMemGetAddressRange(&base, &alloc, (hipDeviceptr_t)p));
        const size_t off = (size_t)(static_cast<char*>(p) - static_cast<char*>((void*)base));
        if (off + len > alloc) return fail("pi_p2p_describe: a buffer runs past the end of its allocation");
        d.buf[b].ptr = (uint64_t)(uintptr_t)p;
        d.buf[b].bytes = (uint64_t)len;
        d.buf[b].base_offset = (uint64_t)off;
        PI_HIP(hipIpcGetMemHandle(&d.buf[b].ipc, (void*)base));
    }
    std::memset(desc512, 0, kDescBytes);
    std::memcpy(desc512, &d, sizeof d);
    h->p2p_pending = pend.release();
    return 0;
}

int pi_comm_init_p2p(pi_handle* h, int rank, int world, const void* descs, const char* cache_dir) {
    if (!h || !descs) return fail("null argument");
    if (!h->p2p_pending) return fail("pi_comm_init_p2p: call pi_p2p_describe first (same handle)");
    pi::DeviceGuard guard(h->device);
    std::unique_ptr<pi::P2pPending> pend(h->p2p_pending);
    h->p2p_pending = nullptr;
    const Desc& me = pend->desc;
    if (me.rank != rank || me.world != world) return fail("pi_comm_init_p2p: rank / world differ from pi_p2p_describe");
    std::vector<Desc> all((size_t)world);
    for (int p = 0; p < world; ++p) {
        std::memcpy(&all[p], static_cast<const char*>(descs) + (size_t)p * kDescBytes, sizeof(Desc));
        const Desc& d = all[p];
        if (d.magic != kMagic || d.version != 1) return fail("pi_comm_init_p2p: descriptor " + std::to_string(p) + " is not one of pi_p2p_describe's");
        if (d.rank != p || d.world != world) return fail("pi_comm_init_p2p: descriptors must be ordered by rank and agree on world");
        if (d.n_bufs != me.n_bufs) return fail("pi_comm_init_p2p: ranks registered different numbers of buffers");
        for (int b = 0; b <= d.n_bufs; ++b)
            if (d.buf[b].bytes != me.buf[b].bytes)
                return fail("pi_comm_init_p2p: registered buffer " + std::to_string(b) + " has different sizes on ranks " +
                            std::to_string(p) + " and " + std::to_string(rank) + " (addresses are symmetric)");
        for (int q = 0; q < p; ++q)
            if (all[q].pid == d.pid)
                return fail("pi_comm_init_p2p: ranks " + std::to_string(q) + " and " + std::to_string(p) +
                            " live in one process — one process per rank (several handles of one process: pi_comm_init_local)");
    }
    if (std::memcmp(&all[rank], &me, sizeof(Desc)) != 0) return fail("pi_comm_init_p2p: descriptor of this rank is not the one pi_p2p_describe returned");

    std::unique_ptr<P2pComm> c(new P2pComm);
    c->rank = rank;
    c->world = world;
    c->device = h->device;
    c->page = pend->page;
    pend->page = nullptr;
    c->ticks = (unsigned long long)(pi::comm_timeout_seconds() * 1.0e8);        // wall_clock64: 100 MHz
    c->sent_n.assign((size_t)world, 0u);
    c->recv_n.assign((size_t)world, 0u);
    c->mine.push_back({c->page + kOffScratch, kScratchBytes});
    for (int b = 1; b <= me.n_bufs; ++b) c->mine.push_back({reinterpret_cast<char*>((uintptr_t)me.buf[b].ptr), (size_t)me.buf[b].bytes});
    c->theirs.assign((size_t)world, std::vector<char*>());
    c->peer_page.assign((size_t)world, nullptr);
    for (int p = 0; p < world; ++p) {
        if (p == rank) {
            c->peer_page[p] = c->page;
            for (auto& b : c->mine) c->theirs[p].push_back(b.base);
            continue;
        }
        // several registered buffers may live in one allocation (a caching allocator's segment): open each once
        std::map<std::string, char*> seen;
        for (int b = 0; b <= all[p].n_bufs; ++b) {
            const DescBuf& db = all[p].buf[b];
            const std::string key(reinterpret_cast<const char*>(&db.ipc), sizeof db.ipc);
            auto it = seen.find(key);
            if (it == seen.end()) {
                void* mapped = nullptr;
                const hipError_t e = hipIpcOpenMemHandle(&mapped, db.ipc, hipIpcMemLazyEnablePeerAccess);
                if (e != hipSuccess)
                    return fail(std::string("pi_comm_init_p2p: hipIpcOpenMemHandle (rank ") + std::to_string(p) + ", buffer " +
                                std::to_string(b) + "): " + hipGetErrorString(e));
                c->opened.push_back(mapped);
                it = seen.emplace(key, static_cast<char*>(mapped)).first;
            }
            char* at = it->second + db.base_offset;
            if (b == 0) {
                c->peer_page[p] = at;
                c->theirs[p].push_back(at + kOffScratch);
            } else {
                c->theirs[p].push_back(at);
            }
        }
    }
    std::vector<char> image;
    const std::string src = std::string("// generated by libpi_mi355 (peer-to-peer transport) for gfx950\n") + pi_embedded_p2p;
    if (pi::compile_image(src, cache_dir, nullptr, 0, image, nullptr)) return 1;
    PI_HIP(hipModuleLoadData(&c->module, image.data()));
    PI_HIP(hipModuleGetFunction(&c->f_sigwait, c->module, "pi_p2p_sigwait_kernel"));
    PI_HIP(hipModuleGetFunction(&c->f_push, c->module, "pi_p2p_push_kernel"));
    PI_HIP(hipModuleGetFunction(&c->f_reduce, c->module, "pi_p2p_reduce_kernel"));
    pi::release_comm(h);
    h->comm = c.release();
    return 0;
}

/* The device code of the transport builds for gfx950 (no GPU needed): __graft_entry__.build() and the CPU tests. */
int pi_p2p_compile_check(const char* cache_dir) {
    std::vector<char> image;
    const std::string src = std::string("// generated by libpi_mi355 (peer-to-peer transport) for gfx950\n") + pi_embedded_p2p;
    return pi::compile_image(src, cache_dir, nullptr, 0, image, nullptr);
}

}  // extern "C"
