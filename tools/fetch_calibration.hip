// tools/fetch_calibration.hip — what do rocprofv3's FETCH_SIZE / WRITE_SIZE report on gfx950 for the access SHAPES of
// the Bellman-backup sweeps?
//
// MI355X_MICROARCH.md (section HBM) calibrates FETCH_SIZE for one shape only — wide coalesced streaming reads, 16 B per
// lane: the counter reports exactly half the bytes — and says "other access widths are uncalibrated: calibrate on a
// known byte count in your own access pattern before trusting an absolute".  The sweeps read V through 4-byte-aligned
// 8-byte loads of neighbouring lanes that OVERLAP (lane i reads floats [b + i, b + i + 1]), the policy through a 4-byte
// non-temporal stream, the mask through a 1-byte non-temporal stream, and write V' as a 4-byte stream.  Every kernel
// below touches a KNOWN number of bytes of a buffer that is far larger than L2 (8 x 4 MiB) and the Infinity Cache
// (256 MiB), each byte (or each line) exactly once per launch:
//
//   cal_stream16 / 8 / 4     coalesced loads of 16 / 8 / 4 bytes per lane over the whole buffer
//   cal_stream4_nt / 1_nt    the same with non-temporal loads (4 bytes: the policy stream; 1 byte: the mask stream)
//   cal_overlap8             lane i of a wave loads 8 bytes at byte offset 4 i (4-byte aligned, overlapping pairs): the
//                            sweeps' corner-pair load along the lane dimension; every float of the buffer is read twice
//                            by neighbouring lanes, every LINE once
//   cal_gather8_line         every lane loads 8 bytes from a line of its own, lines in a scrambled order, each 128-byte
//                            line of the buffer touched exactly once (known: lines; what is fetched per line is the question)
//   cal_gather8_half         the same per 64-byte half line: both halves of every line, the second half a long time after
//                            the first (another half of the grid)
//   cal_store4 / cal_store16 coalesced stores of 4 / 16 bytes per lane over the whole buffer (WRITE_SIZE)
//
// Run under rocprofv3 with --pmc FETCH_SIZE, then --pmc WRITE_SIZE, then the raw request counters
// (TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_BUBBLE_sum), each with --kernel-trace only (tools/fetch_calibration.sh);
// tools/fetch_calibration_fold.py turns the counter CSVs into profiles/rNN/fetch_calibration.txt / .json.
// Every kernel is launched three times; the fold takes the mean of the last two.  The program itself prints the
// bytes every kernel is known to touch and its own timing (HIP events).
//
// build + run (GPU box):  hipcc --offload-arch=gfx950 -O2 tools/fetch_calibration.hip -o /tmp/fetch_cal && /tmp/fetch_cal
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CHECK(e)                                                                          \
    do {                                                                                  \
        hipError_t r_ = (e);                                                              \
        if (r_ != hipSuccess) {                                                           \
            std::fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(r_));                  \
            std::exit(1);                                                                 \
        }                                                                                 \
    } while (0)

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef f2 f2u __attribute__((aligned(4)));          // what the sweep kernels use: 4-byte aligned pairs

constexpr int kBlock = 256;
constexpr int kPerThread = 8;                        // independent loads in flight per lane

__device__ __forceinline__ void sink_if(float acc, float* sink) {
    if (acc == 123456.789f) *sink = acc;             // never true for the zero-filled buffer: keeps the loads alive
}

template <typename T, bool NT>
__device__ __forceinline__ void stream_body(const T* __restrict__ buf, size_t count, float* sink) {
    // block b covers elements [b * kBlock * kPerThread, ...): kPerThread coalesced rows of kBlock elements
    const size_t base = (size_t)blockIdx.x * kBlock * kPerThread + threadIdx.x;
    float acc = 0.0f;
#pragma unroll
    for (int u = 0; u < kPerThread; ++u) {
        const size_t i = base + (size_t)u * kBlock;
        if (i < count) {
            T v = NT ? __builtin_nontemporal_load(buf + i) : buf[i];
            if constexpr (sizeof(T) == 16) acc += v.x + v.y + v.z + v.w;
            else if constexpr (sizeof(T) == 8) acc += v.x + v.y;
            else acc += (float)v;
        }
    }
    sink_if(acc, sink);
}
// the kernel NAME is what tools/fetch_calibration_fold.py keys on
extern "C" __global__ void __launch_bounds__(kBlock) cal_stream16(const f4* b, size_t n, float* s) { stream_body<f4, false>(b, n, s); }
extern "C" __global__ void __launch_bounds__(kBlock) cal_stream8(const f2* b, size_t n, float* s) { stream_body<f2, false>(b, n, s); }
extern "C" __global__ void __launch_bounds__(kBlock) cal_stream4(const float* b, size_t n, float* s) { stream_body<float, false>(b, n, s); }
extern "C" __global__ void __launch_bounds__(kBlock) cal_stream4_nt(const float* b, size_t n, float* s) { stream_body<float, true>(b, n, s); }
extern "C" __global__ void __launch_bounds__(kBlock) cal_stream1_nt(const unsigned char* b, size_t n, float* s) { stream_body<unsigned char, true>(b, n, s); }

// lane i loads the 8 bytes at float index (row start + i): 4-byte aligned overlapping pairs, as the sweeps' corner-pair
// loads along the lane dimension
extern "C" __global__ void __launch_bounds__(kBlock) cal_overlap8(const float* __restrict__ buf, size_t floats, float* sink) {
    const size_t base = (size_t)blockIdx.x * kBlock * kPerThread + threadIdx.x;
    float acc = 0.0f;
#pragma unroll
    for (int u = 0; u < kPerThread; ++u) {
        const size_t i = base + (size_t)u * kBlock;
        if (i + 1 < floats) {
            const f2 v = *reinterpret_cast<const f2u*>(buf + i);
            acc += v.x + v.y;
        }
    }
    sink_if(acc, sink);
}

// One 8-byte load per 128-byte line (cal_gather8_line) or per 64-byte half line (cal_gather8_half: the two halves of a
// line are half a launch apart).  The line a lane touches is a multiplicative scramble of its global index (odd
// multiplier, power-of-two line count: a permutation), so neighbouring lanes hit unrelated lines and every unit is
// touched exactly once.
template <bool HALVES>
__device__ __forceinline__ void gather8_body(const char* __restrict__ buf, size_t lines, float* sink) {
    const size_t base = (size_t)blockIdx.x * kBlock * kPerThread + threadIdx.x;
    const size_t units = HALVES ? lines * 2 : lines;
    float acc = 0.0f;
#pragma unroll
    for (int u = 0; u < kPerThread; ++u) {
        const size_t k = base + (size_t)u * kBlock;
        if (k < units) {
            const size_t line = ((k & (lines - 1)) * 0x9E3779B1ull) & (lines - 1);
            const size_t off = line * 128u + (HALVES ? (k / lines) * 64u : 0u);      // first all low halves, then all high halves
            const f2 v = *reinterpret_cast<const f2*>(buf + off);
            acc += v.x + v.y;
        }
    }
    sink_if(acc, sink);
}
extern "C" __global__ void __launch_bounds__(kBlock) cal_gather8_line(const char* b, size_t lines, float* s) { gather8_body<false>(b, lines, s); }
extern "C" __global__ void __launch_bounds__(kBlock) cal_gather8_half(const char* b, size_t lines, float* s) { gather8_body<true>(b, lines, s); }

template <typename T>
__device__ __forceinline__ void store_body(T* __restrict__ buf, size_t count, float value) {
    const size_t base = (size_t)blockIdx.x * kBlock * kPerThread + threadIdx.x;
#pragma unroll
    for (int u = 0; u < kPerThread; ++u) {
        const size_t i = base + (size_t)u * kBlock;
        if (i < count) {
            if constexpr (sizeof(T) == 16) buf[i] = T{value, value, value, value};
            else buf[i] = value;
        }
    }
}
extern "C" __global__ void __launch_bounds__(kBlock) cal_store4(float* b, size_t n, float v) { store_body<float>(b, n, v); }
extern "C" __global__ void __launch_bounds__(kBlock) cal_store16(f4* b, size_t n, float v) { store_body<f4>(b, n, v); }

static unsigned blocks_for(size_t items) { return (unsigned)((items + (size_t)kBlock * kPerThread - 1) / ((size_t)kBlock * kPerThread)); }

template <typename F>
static void run(const char* name, size_t known_bytes, const char* what, F launch) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {              // the fold takes the last two dispatches of every kernel
        CHECK(hipEventRecord(e0, nullptr));
        launch();
        CHECK(hipEventRecord(e1, nullptr));
        CHECK(hipEventSynchronize(e1));
        float ms = 0.0f;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (rep > 0 && ms < best) best = ms;
    }
    CHECK(hipGetLastError());
    std::printf("%-18s known_bytes %14zu  %8.3f ms  %8.1f GB/s  # %s\n", name, known_bytes, best,
                (double)known_bytes / (best * 1e-3) / 1e9, what);
}

int main(int argc, char** argv) {
    // 1 GiB: 4x the Infinity Cache, 32x the L2s together — nothing a launch reads is left over from the launch before
    const size_t bytes = argc > 1 ? (size_t)std::atoll(argv[1]) << 20 : (size_t)1 << 30;
    if (bytes & (bytes - 1)) { std::fprintf(stderr, "size (MiB) must be a power of two\n"); return 2; }
    char* buf = nullptr;
    float* sink = nullptr;
    CHECK(hipMalloc((void**)&buf, bytes));
    CHECK(hipMalloc((void**)&sink, 256));
    CHECK(hipMemset(buf, 0, bytes));
    CHECK(hipDeviceSynchronize());
    const size_t lines = bytes / 128;
    std::printf("# buffer %zu bytes (%zu lines of 128 B); every kernel 3 launches, time = best of the last two\n", bytes, lines);
    run("cal_stream16", bytes, "16 B/lane coalesced loads, whole buffer once (the guide's calibrated shape: FETCH_SIZE = 1/2)",
        [&] { cal_stream16<<<blocks_for(bytes / 16), kBlock>>>((const f4*)buf, bytes / 16, sink); });
    run("cal_stream8", bytes, "8 B/lane coalesced loads",
        [&] { cal_stream8<<<blocks_for(bytes / 8), kBlock>>>((const f2*)buf, bytes / 8, sink); });
    run("cal_stream4", bytes, "4 B/lane coalesced loads",
        [&] { cal_stream4<<<blocks_for(bytes / 4), kBlock>>>((const float*)buf, bytes / 4, sink); });
    run("cal_stream4_nt", bytes, "4 B/lane coalesced non-temporal loads (the sweeps' policy stream)",
        [&] { cal_stream4_nt<<<blocks_for(bytes / 4), kBlock>>>((const float*)buf, bytes / 4, sink); });
    run("cal_stream1_nt", bytes / 4, "1 B/lane coalesced non-temporal loads over a quarter of the buffer (the sweeps' mask stream)",
        [&] { cal_stream1_nt<<<blocks_for(bytes / 4), kBlock>>>((const unsigned char*)buf, bytes / 4, sink); });
    run("cal_overlap8", bytes, "8 B/lane at 4-byte stride, 4-byte aligned (the sweeps' corner-pair load along the lanes); every line once",
        [&] { cal_overlap8<<<blocks_for(bytes / 4), kBlock>>>((const float*)buf, bytes / 4, sink); });
    run("cal_gather8_line", lines * 128, "one 8-byte load per 128-B line, scrambled order, every line once (known_bytes counts whole lines)",
        [&] { cal_gather8_line<<<blocks_for(lines), kBlock>>>(buf, lines, sink); });
    run("cal_gather8_half", lines * 128, "one 8-byte load per 64-B half line, the two halves of a line half a launch apart",
        [&] { cal_gather8_half<<<blocks_for(lines * 2), kBlock>>>(buf, lines, sink); });
    run("cal_store4", bytes, "4 B/lane coalesced stores (the sweeps' V' stream)",
        [&] { cal_store4<<<blocks_for(bytes / 4), kBlock>>>((float*)buf, bytes / 4, 0.0f); });
    run("cal_store16", bytes, "16 B/lane coalesced stores",
        [&] { cal_store16<<<blocks_for(bytes / 16), kBlock>>>((f4*)buf, bytes / 16, 0.0f); });
    CHECK(hipFree(buf));
    CHECK(hipFree(sink));
    return 0;
}
