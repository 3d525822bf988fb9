// tools/native/host_asan_driver.cpp — the host side of libpi_mi355 under AddressSanitizer + UBSan (CPU only).
//
// GPU sanitizers do not exist on this platform (sanitizer row), so the device side has
// the checked build (PI_MI355_DEBUG=1) and the HOST side — handles, source generation, hipRTC plumbing,
// the exchange planner, argument validation — is compiled from its sources with the address and undefined-behaviour sanitizers
// into this driver by tools/host_sanitizer_check.py and run without a GPU (device = -1 handles).  Any report
// makes the process exit non-zero.
#include "pi_mi355.h"

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#define CHECK(cond)                                                                     \
    do {                                                                                \
        if (!(cond)) {                                                                  \
            std::fprintf(stderr, "CHECK failed: %s (line %d): %s\n", #cond, __LINE__, pi_last_error()); \
            return 1;                                                                   \
        }                                                                               \
    } while (0)

static const char* kPlugin2 =
    "__device__ void step_dynamics(float x, float v, float a, float* nx, float* nv, float* r, bool* t) {\n"
    "  *nv = v + 0.05f * a - 0.1f * sinf(x); *nx = x + 0.05f * *nv; *r = -fabsf(*nx); *t = *nx > 2.0f; }\n";
static const char* kPlugin4 =
    "__device__ void step_dynamics(float a, float b, float c, float d, float u, float* na, float* nb, float* nc,\n"
    "                              float* nd, float* r, bool* t) {\n"
    "  *na = a + 0.02f * b; *nb = b + 0.02f * u; *nc = c + 0.02f * d; *nd = d - 0.02f * cosf(c); *r = 1.0f; *t = false; }\n";

static pi_handle* make(int D, const std::vector<int32_t>& shape, int n_actions) {
    std::vector<float> lo(D, -1.0f), hi(D, 1.0f), actions(n_actions);
    for (int i = 0; i < n_actions; ++i) actions[i] = (float)i - 1.0f;
    std::vector<std::vector<float>> tabs(D);
    std::vector<const float*> ptrs(D);
    for (int d = 0; d < D; ++d) {
        tabs[d].resize(shape[d]);
        for (int i = 0; i < shape[d]; ++i) tabs[d][i] = -1.0f + 2.0f * (float)i / (float)(shape[d] - 1);
        ptrs[d] = tabs[d].data();
    }
    return pi_create(-1, D, shape.data(), lo.data(), hi.data(), ptrs.data(), actions.data(), n_actions);
}

int main(int argc, char** argv) {
    const char* cache = argc > 1 ? argv[1] : nullptr;
    CHECK(pi_abi_version() == PI_MI355_ABI_VERSION);
    // argument validation: every rejection leaves a message and no handle
    {
        std::vector<int32_t> bad = {4, 4, 4};
        CHECK(make(3, bad, 2) == nullptr && std::strlen(pi_last_error()) > 0);
        std::vector<int32_t> tiny = {1, 4};
        CHECK(make(2, tiny, 2) == nullptr);
        std::vector<int32_t> huge(6, 40);
        CHECK(make(6, huge, 1) == nullptr);
        CHECK(pi_info(nullptr, 0) == -1 || std::strlen(pi_last_error()) > 0);
    }
    // 2-D and 4-D handles: source generation, compilation for gfx950, cache round trip, options, info
    for (int D : {2, 4}) {
        std::vector<int32_t> shape = D == 2 ? std::vector<int32_t>{21, 11} : std::vector<int32_t>{7, 5, 6, 4};
        pi_handle* h = make(D, shape, 3);
        CHECK(h != nullptr);
        const char* plugin = D == 2 ? kPlugin2 : kPlugin4;
        const size_t need = pi_kernel_source(h, plugin, nullptr, 0);
        CHECK(need > 1000);
        std::vector<char> src(need + 1);
        CHECK(pi_kernel_source(h, plugin, src.data(), src.size()) == need && std::strlen(src.data()) == need);
        std::vector<char> small(64);                                  // truncating copy stays inside the buffer
        CHECK(pi_kernel_source(h, plugin, small.data(), small.size()) == need && std::strlen(small.data()) == 63);
        std::vector<char> log(1 << 14);
        CHECK(pi_compile(h, plugin, cache, log.data(), log.size()) == 0);
        CHECK(pi_compile(h, "__device__ void step_dynamics(float a) { nonsense; }", cache, log.data(), log.size()) != 0);
        CHECK(std::strlen(log.data()) > 0);
        CHECK(pi_compile(h, plugin, cache, nullptr, 0) == 0);          // no log buffer
        CHECK(pi_info(h, 0) == (D == 2 ? 231 : 840) && pi_info(h, 1) == 3 && pi_info(h, 2) == D);
        CHECK(pi_set_option(h, 0, 4) == 0 && pi_info(h, 3) == 4 && pi_set_option(h, 0, 0) != 0 && pi_set_option(h, 99, 1) != 0);
        float dummy = 0.0f;
        CHECK(pi_eval_sweep(h, &dummy, &dummy, nullptr, nullptr, 0, 1, 0.9f, nullptr, nullptr) != 0);   // host-only: refuses
        uint32_t rep[4];
        CHECK(pi_debug_report(h, rep) != 0);
        CHECK(pi_prepare_mask(h, nullptr, nullptr) != 0);                     // host-only handle: refused
        int32_t one_policy[1] = {0};
        CHECK(pi_eval_begin(h, one_policy, nullptr, nullptr) != 0 && pi_eval_end(h) == 0 && pi_eval_end(nullptr) != 0);
        pi_destroy(h);
    }
    // exchange planner (host-only): random reach bitmaps, every world size, tiny caps
    {
        const int64_t g0 = 13, stride0 = 35, n = g0 * stride0;
        for (int world = 1; world <= 5; ++world) {
            const int64_t per = (n + world - 1) / world;
            std::vector<uint8_t> reach((size_t)world * g0);
            unsigned int seed = 1234u + (unsigned)world;
            for (auto& b : reach) { seed = seed * 1664525u + 1013904223u; b = (seed >> 24) & 1u; }
            const int64_t count = pi_plan_segments(world, g0, stride0, n, per, reach.data(), nullptr, 0);
            CHECK(count >= 0);
            std::vector<int64_t> segs((size_t)std::max<int64_t>(count, 1) * 4);
            CHECK(pi_plan_segments(world, g0, stride0, n, per, reach.data(), segs.data(), count) == count);
            if (count > 1) {                                           // a cap below the count must not overrun
                std::vector<int64_t> one(4, -7);
                CHECK(pi_plan_segments(world, g0, stride0, n, per, reach.data(), one.data(), 1) == count);
            }
            for (int64_t i = 0; i < count; ++i) {
                const int64_t src = segs[4 * i], dst = segs[4 * i + 1], a = segs[4 * i + 2], b = segs[4 * i + 3];
                CHECK(src >= 0 && src < world && dst >= 0 && dst < world && src != dst && 0 <= a && a < b && b <= n);
            }
        }
        CHECK(pi_plan_segments(0, g0, stride0, n, n, nullptr, nullptr, 0) < 0);
    }
    // inference handle, host-only
    {
        const float lo[4] = {-1, -1, -1, -1}, hi[4] = {1, 1, 1, 1};
        const int32_t shape[4] = {5, 4, 6, 3}, strides[4] = {72, 18, 3, 1};
        int32_t bits[16 * 4];
        for (int c = 0; c < 16; ++c)
            for (int d = 0; d < 4; ++d) bits[c * 4 + d] = (c >> (3 - d)) & 1;
        pi_infer* q = pi_infer_create(-1, 4, lo, hi, shape, strides, bits, 16, cache);
        CHECK(q != nullptr);
        const int32_t pol[1] = {0};
        const float acts[1] = {0.0f};
        CHECK(pi_infer_set_policy(q, pol, 1, acts, 1) != 0);           // host-only handle holds no policy
        CHECK(pi_infer_query(q, nullptr, 4, nullptr, nullptr, nullptr, nullptr) != 0);
        pi_infer_destroy(q);
        bits[5] = 2;
        CHECK(pi_infer_create(-1, 4, lo, hi, shape, strides, bits, 16, cache) == nullptr);
        CHECK(pi_infer_create(-1, 4, lo, hi, shape, strides, bits, 8, cache) == nullptr);
        bits[5] = 0;
        const int32_t foreign[4] = {400, 20, 5, 1}, negative[4] = {72, 18, -3, 1};     // another grid's / negative strides
        CHECK(pi_infer_create(-1, 4, lo, hi, shape, foreign, bits, 16, cache) == nullptr);
        CHECK(pi_infer_create(-1, 4, lo, hi, shape, negative, bits, 16, cache) == nullptr);
    }
    // peer-to-peer transport: device code builds for gfx950; a host-only handle is refused by every entry point
    {
        CHECK(pi_p2p_compile_check(cache) == 0);
        std::vector<int32_t> shape = {6, 5};
        pi_handle* h = make(2, shape, 3);
        CHECK(h != nullptr);
        unsigned char desc[1024] = {0};
        const void* bufs[1] = {desc};
        const int64_t sizes[1] = {64};
        CHECK(pi_p2p_describe(h, 0, 2, bufs, sizes, 1, desc) != 0 && std::strlen(pi_last_error()) > 0);
        CHECK(pi_p2p_describe(nullptr, 0, 2, bufs, sizes, 1, desc) != 0);
        CHECK(pi_comm_init_p2p(h, 0, 2, desc, cache) != 0);              // nothing was described
        CHECK(pi_comm_init_p2p(h, 0, 2, nullptr, cache) != 0);
        CHECK(pi_comm_info(h, 2) == -1);
        pi_destroy(h);
    }
    pi_destroy(nullptr);
    pi_infer_destroy(nullptr);
    std::puts("host_asan_driver: ok");
    return 0;
}
