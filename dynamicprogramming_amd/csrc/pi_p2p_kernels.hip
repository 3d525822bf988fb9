// pi_p2p_kernels.hip — device side of the peer-to-peer transport (pi_p2p.cpp): halo rows are STORED by the sending
// GPU straight into the receiving rank's V' buffer (mapped with hipIpcOpenMemHandle; the stores travel over xGMI) and
// hand-shaken through 32-bit counters in a small uncached "flag page" every rank owns.  No reference counterpart
// (src/cuda_policy_iteration.py:300-336 is a single-device loop); it replaces the grouped ncclSend / ncclRecv of the
// RCCL transport, whose group latency (~30 us) is of the order of a rank's whole sweep at C4 @ 8.
//
// Compiled by hipRTC for gfx950 on its own (no grid, no env plugin).  Every wait is BOUNDED: a counter that does not
// arrive within `ticks` of the 100 MHz wall clock sets a bit in the rank's error word, and every later wait of that
// rank returns at once — the host finds the word at the next reduction (pi_p2p.cpp: health) — so a dead peer is an
// error, never a hung wave.
//
// Counters only ever grow (message number of a (sender, receiver) pair), compared wrap-safe.

#define PI_P2P_MAX 16                     // ranks a flag page has slots for

struct PiP2pFlags {                        // up to PI_P2P_MAX (address, value) pairs, by value in the kernel arguments
    unsigned int* ptr[PI_P2P_MAX];
    unsigned int value[PI_P2P_MAX];
    int n;
};

struct PiP2pSeg {                          // one run of floats: dst lives on a peer; `first` = units in front of it
    float* dst;
    const float* src;
    long long first;
};

struct PiP2pRed {
    unsigned long long* peer_slot[PI_P2P_MAX];   // slot [epoch parity][my rank] in every rank's page (own page included)
    const unsigned long long* mine;              // slots [epoch parity][0 .. world) of this rank's page
    int world;
    unsigned int epoch;
    int op;                                      // 0 = max of non-negative float32, 1 = sum of uint32
};

__device__ __forceinline__ unsigned int pi_p2p_load(const unsigned int* p) {
    return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ void pi_p2p_store(unsigned int* p, unsigned int v) {
    __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
// Spin until *p has reached `value`; false when it has not within `ticks` (or the rank already failed).
__device__ __forceinline__ bool pi_p2p_await(const unsigned int* p, unsigned int value, unsigned long long ticks,
                                             const unsigned int* error_word) {
    if ((int)(pi_p2p_load(p) - value) >= 0) return true;
    if (pi_p2p_load(error_word) != 0u) return false;
    const unsigned long long t0 = wall_clock64();
    while ((int)(pi_p2p_load(p) - value) < 0) {
        if (wall_clock64() - t0 > ticks) return false;
        __builtin_amdgcn_s_sleep(32);
    }
    return true;
}

// Head of a group, one wave: first RAISE "my receive number value[i] is posted" at every sender of this group (its
// target region is free from now on), then WAIT until every receiver of this group's sends has posted its own.  Signals
// strictly before waits on every rank, so two ranks that send to each other cannot wait for each other.
// Tail of a group: sig.n == 0, wait = the data counters of this group's senders.  error bits 0..15: which pair timed out.
extern "C" __global__ void __launch_bounds__(64)
pi_p2p_sigwait_kernel(PiP2pFlags sig, PiP2pFlags wait, unsigned long long ticks, unsigned int* error_word) {
    const int i = (int)threadIdx.x;
    if (sig.n > 0) {
        __threadfence_system();
        if (i < sig.n) pi_p2p_store(sig.ptr[i], sig.value[i]);
    }
    if (i < wait.n && !pi_p2p_await(wait.ptr[i], wait.value[i], ticks, error_word)) atomicOr(error_word, 1u << i);
    __threadfence_system();
}

// The copy of a group: (optionally wait until every receiver has posted its receive — acks; the host launches the
// one-wave kernel above for that instead, so that this grid never parks on the CUs), copy all segments into the peers'
// buffers, then — the LAST workgroup to finish, after a system-scope fence — raise the data counters (done).
// `vec4` != 0: every segment is 16-byte aligned on both sides and a multiple of four floats long; units are float4.
extern "C" __global__ void __launch_bounds__(256)
pi_p2p_push_kernel(const PiP2pSeg* __restrict__ segs, int n_segs, int vec4, PiP2pFlags acks, PiP2pFlags done,
                   unsigned long long ticks, unsigned int* error_word, unsigned int* block_counter) {
    __shared__ int s_ok, s_last;
    const int tid = (int)threadIdx.x;
    if (tid == 0) { s_ok = 1; s_last = 0; }
    __syncthreads();
    if (tid < acks.n && !pi_p2p_await(acks.ptr[tid], acks.value[tid], ticks, error_word)) {
        atomicOr(error_word, 0x10000u << tid);
        s_ok = 0;
    }
    __syncthreads();
    if (s_ok) {
        const long long total = segs[n_segs].first;
        const long long step = (long long)gridDim.x * 256;
        for (long long u = (long long)blockIdx.x * 256 + tid; u < total; u += step) {
            int lo = 0, hi = n_segs;                       // the segment that holds unit u: last one with first <= u
            while (hi - lo > 1) {
                const int mid = (lo + hi) >> 1;
                if (segs[mid].first <= u) lo = mid; else hi = mid;
            }
            const long long k = u - segs[lo].first;
            if (vec4) reinterpret_cast<float4*>(segs[lo].dst)[k] = reinterpret_cast<const float4*>(segs[lo].src)[k];
            else segs[lo].dst[k] = segs[lo].src[k];
        }
    }
    __threadfence_system();
    __syncthreads();
    if (tid == 0) {
        const unsigned int before = atomicAdd(block_counter, 1u);
        if (before == gridDim.x - 1) {
            atomicExch(block_counter, 0u);
            s_last = 1;
        }
    }
    __syncthreads();
    if (s_last && s_ok) {
        __threadfence_system();
        if (tid < done.n) pi_p2p_store(done.ptr[tid], done.value[tid]);
    }
}

// All-reduce of ONE word without RCCL: every rank writes (epoch << 32 | bits) into slot [epoch & 1][its rank] of every
// rank's page and waits until its own page holds all `world` contributions of this epoch.  Two slot rows by epoch
// parity: a peer can only be one reduction ahead (it needs this rank's contribution to finish the next one).
// error bits 0..15.
extern "C" __global__ void __launch_bounds__(64)
pi_p2p_reduce_kernel(unsigned int* d, PiP2pRed r, unsigned long long ticks, unsigned int* error_word) {
    const int lane = (int)threadIdx.x;
    const unsigned int bits = *d;
    unsigned int got = 0u;
    if (lane < r.world) {
        __hip_atomic_store(r.peer_slot[lane], ((unsigned long long)r.epoch << 32) | bits, __ATOMIC_RELEASE,
                           __HIP_MEMORY_SCOPE_SYSTEM);
        bool ok = pi_p2p_load(error_word) == 0u;
        const unsigned long long t0 = wall_clock64();
        unsigned long long v = 0ull;
        while (ok) {
            v = __hip_atomic_load(r.mine + lane, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
            if ((unsigned int)(v >> 32) == r.epoch) break;
            if (wall_clock64() - t0 > ticks) ok = false;
            __builtin_amdgcn_s_sleep(32);
        }
        if (!ok) atomicOr(error_word, 1u << lane);
        got = ok ? (unsigned int)v : 0u;
    }
    unsigned int acc = got;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned int t = __shfl_xor(acc, o, 64);
        if (r.op == 0) {                                  // non-negative floats order like their bit patterns
            const float a = __uint_as_float(acc), b = __uint_as_float(t);
            acc = b > a ? t : acc;
        } else {
            acc += t;
        }
    }
    if (lane == 0) *d = acc;
}
