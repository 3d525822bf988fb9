// tools/tcp_gather_bench.hip — how many distinct 128-byte lines per second can a CU's vector L1
// (TCP) serve to wave-wide gather loads on gfx950?
//
// The evaluation sweep's corner gather is a divergent load: the 64 lanes of one
// global_load_dwordx2 touch 15-35 different lines (tools/gather_lines.py).  The guides give no
// throughput for that case, so this measures it: every wave issues 8-byte loads (and, for the
// floor of a coalesced load, 1-, 4- and 16-byte ones) whose 64 lanes fall into `lines` different
// 128-byte lines (lanes of a group share one line, consecutive words), eight independent loads in
// flight, for
//   L1 : one 16 KiB window (128 lines) that stays resident in every CU's 32 KiB L1 (all hits);
//   L2 : a 2 MiB window shared by all workgroups (L1 misses, L2 hits).
// Reported: wave-loads/s and line look-ups/s per CU, and cycles per wave-load at the shader clock
// measured with s_memtime / s_memrealtime.  The sweep kernel's own rate (TCP_TOTAL_CACHE_ACCESSES
// per CU per second, profiles/r02/counters_*.json) is to be read against the L1 row.
//
// build + run (GPU box):  hipcc --offload-arch=gfx950 -O2 tools/tcp_gather_bench.hip -o /tmp/tgb && /tmp/tgb
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

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
typedef f4 f4u __attribute__((aligned(4)));
__device__ inline float fold(unsigned char v) { return (float)v; }
__device__ inline float fold(float v) { return v; }
__device__ inline float fold(f2 v) { return v.x + v.y; }
__device__ inline float fold(f4 v) { return v.x + v.y + v.z + v.w; }

struct Stamp {
    unsigned long long t0, t1, r0, r1;
};

// window_lines: power of two; the pattern of one load covers `lines` lines starting at a
// wave-uniform line offset that advances by `lines` per load and wraps in the window.
template <typename T>
__global__ void __launch_bounds__(256)
gather_kernel(const float* __restrict__ buf, float* __restrict__ sink, Stamp* __restrict__ stamps,
              int iters, int lines, unsigned int window_lines, unsigned int byte_shift, int interleave) {
    const unsigned int lane = threadIdx.x & 63u;
    // line of this lane within the pattern: neighbouring lanes share a line (grouped), or lanes
    // that share a line sit `lines` apart (interleaved: every quad of lanes spans several lines)
    // interleave == 3 ("shifted runs"): grouped, but every run of lanes that share a line starts half a run late, so
    // that with 16 lines (runs of 4 lanes) every quad of lanes spans TWO lines while the wave still walks 17 runs
    const unsigned int run = 64u / (unsigned int)lines > 0u ? 64u / (unsigned int)lines : 1u;
    const unsigned int group = interleave == 3 ? (lane + run / 2u) / run
                             : interleave ? lane % (unsigned int)lines : lane * (unsigned int)lines / 64u;
    constexpr unsigned int kPerLine = 128u / sizeof(T);                       // elements of T per line
    // with the 4-byte shift the last element of a line would straddle two lines: leave it out
    const unsigned int in_line = interleave == 3 ? (lane + run / 2u) % run
                               : interleave ? lane / (unsigned int)lines : lane % (64u / (unsigned int)lines > 0 ? 64u / (unsigned int)lines : 1u);
    const unsigned int word = in_line % (kPerLine - (byte_shift ? 1u : 0u));
    // byte offset of this lane inside the pattern: line * 128 + its element
    const unsigned int lane_off = group * 128u + word * (unsigned int)sizeof(T);
    const float* base = reinterpret_cast<const float*>(reinterpret_cast<const char*>(buf) + byte_shift);   // 0, or 4: loads aligned to 4 bytes only
    const unsigned int mask = window_lines - 1u;
    // wave-uniform cursor (scalar registers): waves start on different lines
    unsigned int cursor = __builtin_amdgcn_readfirstlane((threadIdx.x >> 6) * (unsigned int)lines);
    float acc[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) acc[u] = 0.0f;
    unsigned long long t0 = __builtin_readcyclecounter(), r0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const unsigned int line0 = cursor & mask;
            const T v = *reinterpret_cast<const T*>(reinterpret_cast<const char*>(base) + (size_t)line0 * 128u + lane_off);
            acc[u] += fold(v);
            cursor += (unsigned int)lines;
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter(), r1 = wall_clock64();
    float s = 0.0f;
#pragma unroll
    for (int u = 0; u < 8; ++u) s += acc[u];
    if (s == 123.456f) sink[0] = s;
    if (lane == 0) {
        Stamp st = {t0, t1, r0, r1};
        stamps[(size_t)blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6)] = st;
    }
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? std::atoi(argv[1]) : 4096;
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    std::printf("# device %s, %d CUs; %d x 8 wave-wide 8-byte loads per wave, 256-thread workgroups, 8 per CU\n",
                prop.gcnArchName, cus, iters);
    const size_t buf_floats = (size_t)64 << 20;                               // 256 MiB
    float* buf;
    float* sink;
    CHECK(hipMalloc((void**)&buf, buf_floats * sizeof(float)));
    CHECK(hipMemset(buf, 0, buf_floats * sizeof(float)));
    CHECK(hipMalloc((void**)&sink, 4));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const int blocks = cus * 8;                                               // 8 waves per SIMD
    Stamp* d_st;
    CHECK(hipMalloc((void**)&d_st, sizeof(Stamp) * blocks * 4));
    std::printf("%-6s %6s %6s %16s %18s %20s %12s\n", "set", "bytes", "lines", "wave-loads/s/CU", "line-lookups/s/CU",
                "cycles/wave-load(CU)", "clock GHz");
    typedef void (*Kern)(const float*, float*, Stamp*, int, int, unsigned int, unsigned int, int);
    struct Width { int bytes; Kern fn; };
    const Width widths[] = {{8, gather_kernel<f2u>}, {1, gather_kernel<unsigned char>}, {4, gather_kernel<float>},
                            {16, gather_kernel<f4u>}};
    struct Set { const char* name; unsigned int window_lines; unsigned int byte_shift; int interleave; };
    // L1: all workgroups read ONE 128-line (16 KiB) window; L2: one 2 MiB window for everybody.
    const Set sets[] = {{"L1", 128u, 0u, 0}, {"L2", 16384u, 0u, 0}, {"L1+4", 128u, 4u, 0},   // +4: every address = 4 mod 8
                        {"L1i", 128u, 0u, 1}, {"L2i", 16384u, 0u, 1},                        // i: interleaved lanes
                        {"L1r", 128u, 0u, 3},                                                // r: runs shifted off the quad grid
                        {"same", 1u, 0u, 0}, {"samei", 1u, 0u, 1}};                          // every load re-reads the same lines
    for (const Width& wd : widths)
    for (const Set& set : sets) {
        if (wd.bytes != 8 && set.window_lines != 128u) continue;              // other widths: L1 floor only
        if (wd.bytes < 8 && set.byte_shift) continue;
        for (int lines : {1, 2, 4, 8, 16, 32, 64}) {
            wd.fn<<<blocks, 256>>>(buf, sink, d_st, 64, lines, set.window_lines, set.byte_shift, set.interleave);
            CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0));
            wd.fn<<<blocks, 256>>>(buf, sink, d_st, iters, lines, set.window_lines, set.byte_shift, set.interleave);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms = 0;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            std::vector<Stamp> h((size_t)blocks * 4);
            CHECK(hipMemcpy(h.data(), d_st, sizeof(Stamp) * h.size(), hipMemcpyDeviceToHost));
            std::vector<double> ghz;
            for (const Stamp& s : h)
                if (s.r1 > s.r0) ghz.push_back((double)(s.t1 - s.t0) / (double)(s.r1 - s.r0) * 0.1);
            std::sort(ghz.begin(), ghz.end());
            const double clock = ghz.empty() ? 0.0 : ghz[ghz.size() / 2];
            const double wave_loads = (double)blocks * 4 * iters * 8;
            const double per_cu = wave_loads / (ms * 1e-3) / cus;
            std::printf("%-6s %6d %6d %16.4g %18.4g %20.2f %12.3f\n", set.name, wd.bytes, lines, per_cu, per_cu * lines,
                        clock * 1e9 / per_cu, clock);
        }
    }
    return 0;
}
