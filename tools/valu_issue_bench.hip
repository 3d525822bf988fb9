// tools/valu_issue_bench.hip — how many cycles does one wave64 VALU instruction cost on a
// gfx950 SIMD, as a function of the waves resident on that SIMD and of the instruction-level
// parallelism inside a wave?
//
// Settles the question VERDICT r01 raised about the sweep kernels' bound: MI355X_MICROARCH.md
// quotes v_fma_f32 at 2 cycles per wave-instruction (SIMD-32) but 4 cycles for "one wave alone".
// Every kernel below is a loop of 64 unrolled inline-asm instructions on registers only, so
// nothing but VALU issue is measured.  Each wave stamps s_memtime before and after its loop and
// records which CU/SIMD it ran on (HW_REG_HW_ID), so the host can report
//   cyc/instr(SIMD) = wave-loop cycles / (instructions per wave x waves that shared the SIMD)
// for exactly the waves that really were co-resident, plus the chip-wide rate from wall time.
//
// build + run (GPU box):  hipcc --offload-arch=gfx950 -O2 tools/valu_issue_bench.hip -o /tmp/vib && /tmp/vib
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#define CHECK(e)                                                                          \
    do {                                                                                  \
        hipError_t r_ = (e);                                                              \
        if (r_ != hipSuccess) {                                                           \
            std::fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(r_));                  \
            std::exit(1);                                                                 \
        }                                                                                 \
    } while (0)

struct Stamp {
    unsigned long long t0, t1;      // s_memtime (shader clock)
    unsigned long long r0, r1;      // s_memrealtime (100 MHz)
    unsigned int hw_id, pad;
};

constexpr int kUnroll = 64;

// One row = one instruction text "A %k B %k C" repeated over 8 independent accumulators %0..%7
// (operands %8, %9 are loop-invariant inputs), 8 times per loop iteration = 64 instructions.
#define PI_I(k, A, B, C) A "%" #k B "%" #k C "\n\t"
#define PI_REP8(A, B, C) PI_I(0, A, B, C) PI_I(1, A, B, C) PI_I(2, A, B, C) PI_I(3, A, B, C) \
                         PI_I(4, A, B, C) PI_I(5, A, B, C) PI_I(6, A, B, C) PI_I(7, A, B, C)
#define PI_DEP8(A, B, C) PI_I(0, A, B, C) PI_I(0, A, B, C) PI_I(0, A, B, C) PI_I(0, A, B, C) \
                         PI_I(0, A, B, C) PI_I(0, A, B, C) PI_I(0, A, B, C) PI_I(0, A, B, C)

template <typename T> __device__ T pi_seed(int k);
template <> __device__ float pi_seed<float>(int k) { return (float)threadIdx.x * 1e-3f + (float)k; }
template <> __device__ double pi_seed<double>(int k) { return (double)threadIdx.x * 1e-3 + (double)k; }
typedef float f2 __attribute__((ext_vector_type(2)));
template <> __device__ f2 pi_seed<f2>(int k) { f2 r = {(float)threadIdx.x * 1e-3f + (float)k, (float)k}; return r; }
__device__ float pi_fold(float v) { return v; }
__device__ float pi_fold(double v) { return (float)v; }
__device__ float pi_fold(f2 v) { return v.x + v.y; }

#define DEFINE_BENCH(NAME, T, BODY)                                                        \
    __global__ void __launch_bounds__(256) NAME(Stamp* out, float* sink, int iters) {      \
        T a0 = pi_seed<T>(0), a1 = pi_seed<T>(1), a2 = pi_seed<T>(2), a3 = pi_seed<T>(3),  \
          a4 = pi_seed<T>(4), a5 = pi_seed<T>(5), a6 = pi_seed<T>(6), a7 = pi_seed<T>(7);  \
        T x = pi_seed<T>(9), y = pi_seed<T>(11);                                           \
        unsigned long long t0 = __builtin_amdgcn_s_memtime();                              \
        unsigned long long r0 = __builtin_amdgcn_s_memrealtime();                          \
        for (int i = 0; i < iters; ++i) {                                                  \
            _Pragma("unroll") for (int u = 0; u < kUnroll / 8; ++u)                        \
                asm volatile(BODY                                                          \
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), \
                               "+v"(a6), "+v"(a7)                                          \
                             : "v"(x), "v"(y)                                              \
                             : "vcc", "s40", "s41");                                                   \
        }                                                                                  \
        unsigned long long t1 = __builtin_amdgcn_s_memtime();                              \
        unsigned long long r1 = __builtin_amdgcn_s_memrealtime();                          \
        float res = pi_fold(a0) + pi_fold(a1) + pi_fold(a2) + pi_fold(a3) + pi_fold(a4) +  \
                    pi_fold(a5) + pi_fold(a6) + pi_fold(a7);                               \
        if ((threadIdx.x & 63) == 0) {                                                     \
            unsigned int hw, xcc;                                                          \
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));               \
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));             \
            Stamp s = {t0, t1, r0, r1, hw, xcc};                                           \
            out[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = s;                         \
        }                                                                                  \
        if (res == 12345.678f) sink[0] = res;                                              \
    }

// Same kernel with a loop body of 64 hand-laid instructions (no inner repetition): the mixed and
// clustered streams below need more than 8 slots per period.
#define DEFINE_BENCH64(NAME, BODY)                                                         \
    __global__ void __launch_bounds__(256) NAME(Stamp* out, float* sink, int iters) {      \
        float a0 = pi_seed<float>(0), a1 = pi_seed<float>(1), a2 = pi_seed<float>(2),      \
              a3 = pi_seed<float>(3), a4 = pi_seed<float>(4), a5 = pi_seed<float>(5),      \
              a6 = pi_seed<float>(6), a7 = pi_seed<float>(7);                              \
        float x = pi_seed<float>(9), y = pi_seed<float>(11);                               \
        unsigned long long t0 = __builtin_amdgcn_s_memtime();                              \
        unsigned long long r0 = __builtin_amdgcn_s_memrealtime();                          \
        for (int i = 0; i < iters; ++i) {                                                  \
            asm volatile(BODY                                                              \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5),     \
                           "+v"(a6), "+v"(a7)                                              \
                         : "v"(x), "v"(y)                                                  \
                         : "vcc", "s40", "s41");                                           \
        }                                                                                  \
        unsigned long long t1 = __builtin_amdgcn_s_memtime();                              \
        unsigned long long r1 = __builtin_amdgcn_s_memrealtime();                          \
        float res = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;                                 \
        if ((threadIdx.x & 63) == 0) {                                                     \
            unsigned int hw, xcc;                                                          \
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));               \
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));             \
            Stamp s = {t0, t1, r0, r1, hw, xcc};                                           \
            out[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = s;                         \
        }                                                                                  \
        if (res == 12345.678f) sink[0] = res;                                              \
    }

// X(name, type, A, B, C): instruction = A %k B %k C
#define ROWS(X)                                                                            \
    X(v_fma_f32, float, "v_fma_f32 ", ", %8, %9, ", "")                                    \
    X(v_fmac_f32, float, "v_fmac_f32 ", ", %8, %9 ; ", "")                                 \
    X(v_mul_f32, float, "v_mul_f32 ", ", %8, ", "")                                        \
    X(v_add_f32, float, "v_add_f32 ", ", %8, ", "")                                        \
    X(v_sub_f32, float, "v_sub_f32 ", ", %8, ", "")                                        \
    X(v_max_f32, float, "v_max_f32 ", ", %8, ", "")                                        \
    X(v_min3_f32, float, "v_min3_f32 ", ", %8, %9, ", "")                                  \
    X(v_med3_f32, float, "v_med3_f32 ", ", %8, %9, ", "")                                  \
    X(v_mov_b32, float, "v_mov_b32 ", ", %8 ; ", "")                                       \
    X(v_cndmask_b32, float, "v_cndmask_b32 ", ", %8, ", ", vcc")                           \
    X(v_cmp_gt_f32, float, "v_cmp_gt_f32 vcc, ", ", ", "")                                 \
    X(v_cmp_gt_f32_sgpr, float, "v_cmp_gt_f32 s[40:41], ", ", ", "")                       \
    X(v_and_b32, float, "v_and_b32 ", ", %8, ", "")                                        \
    X(v_xor_b32, float, "v_xor_b32 ", ", %8, ", "")                                        \
    X(v_add_u32, float, "v_add_u32 ", ", %8, ", "")                                        \
    X(v_lshlrev_b32, float, "v_lshlrev_b32 ", ", 1, ", "")                                 \
    X(v_lshl_add_u32, float, "v_lshl_add_u32 ", ", %8, 2, ", "")                           \
    X(v_mul_u32_u24, float, "v_mul_u32_u24 ", ", %8, ", "")                                \
    X(v_mad_u32_u24, float, "v_mad_u32_u24 ", ", %8, %9, ", "")                            \
    X(v_mul_lo_u32, float, "v_mul_lo_u32 ", ", %8, ", "")                                  \
    X(v_mul_hi_u32, float, "v_mul_hi_u32 ", ", %8, ", "")                                  \
    X(v_bfe_u32, float, "v_bfe_u32 ", ", ", ", 3, 7")                                      \
    X(v_cvt_i32_f32, float, "v_cvt_i32_f32 ", ", ", "")                                    \
    X(v_cvt_f32_i32, float, "v_cvt_f32_i32 ", ", ", "")                                    \
    X(v_trunc_f32, float, "v_trunc_f32 ", ", ", "")                                        \
    X(v_rndne_f32, float, "v_rndne_f32 ", ", ", "")                                        \
    X(v_ldexp_f32, float, "v_ldexp_f32 ", ", ", ", 1")                                     \
    X(v_rcp_f32, float, "v_rcp_f32 ", ", ", "")                                            \
    X(v_sin_f32, float, "v_sin_f32 ", ", ", "")                                            \
    X(v_div_scale_f32, float, "v_div_scale_f32 ", ", vcc, %8, %9, ", "")                   \
    X(v_div_fmas_f32, float, "v_div_fmas_f32 ", ", %8, %9, ", "")                          \
    X(v_div_fixup_f32, float, "v_div_fixup_f32 ", ", %8, %9, ", "")                        \
    X(v_pk_fma_f32, f2, "v_pk_fma_f32 ", ", %8, %9, ", "")                                 \
    X(v_pk_mul_f32, f2, "v_pk_mul_f32 ", ", %8, ", "")                                     \
    X(v_pk_add_f32, f2, "v_pk_add_f32 ", ", %8, ", "")                                     \
    X(v_fma_f64, double, "v_fma_f64 ", ", %8, %9, ", "")                                   \
    X(v_mul_f64, double, "v_mul_f64 ", ", %8, ", "")                                       \
    X(v_add_f64, double, "v_add_f64 ", ", %8, ", "")                                       \
    X(v_lshl_add_u64, double, "v_lshl_add_u64 ", ", ", ", 2, %8")                          \
    X(s_nop_0, float, "s_nop 0 ; ", " ", "")                                               \
    X(s_mov_b32, float, "s_mov_b32 s40, s41 ; ", " ", "")

#define X_DEF(NAME, T, A, B, C) DEFINE_BENCH(bench_##NAME, T, PI_REP8(A, B, C))
ROWS(X_DEF)
DEFINE_BENCH(bench_dep_v_fma_f32, float, PI_DEP8("v_fma_f32 ", ", %8, %9, ", ""))
DEFINE_BENCH(bench_dep_v_mul_f32, float, PI_DEP8("v_mul_f32 ", ", %8, ", ""))
DEFINE_BENCH(bench_dep_v_max_f32, float, PI_DEP8("v_max_f32 ", ", %8, ", ""))
// a mix like the sweep kernels': FMA-class and other-class instructions alternating
DEFINE_BENCH(bench_mix_fma_max, float,
             PI_I(0, "v_fma_f32 ", ", %8, %9, ", "") PI_I(1, "v_max_f32 ", ", %8, ", "")
             PI_I(2, "v_fma_f32 ", ", %8, %9, ", "") PI_I(3, "v_max_f32 ", ", %8, ", "")
             PI_I(4, "v_fma_f32 ", ", %8, %9, ", "") PI_I(5, "v_max_f32 ", ", %8, ", "")
             PI_I(6, "v_fma_f32 ", ", %8, %9, ", "") PI_I(7, "v_max_f32 ", ", %8, ", ""))
// v_cndmask reading a condition that a v_cmp of the same wave has just produced (the realistic use;
// the bare v_cndmask_b32 row above reads a VCC that no instruction of the loop ever writes and is
// an outlier at ~23 cycles): two instructions per accumulator, so divide the pair's cost by two.
#define PI_CS(k) "v_cmp_gt_f32 vcc, %" #k ", %8\n\tv_cndmask_b32 %" #k ", %8, %" #k ", vcc\n\t"
DEFINE_BENCH(bench_pair_cmp_cndmask, float,
             PI_CS(0) PI_CS(1) PI_CS(2) PI_CS(3))
#define PI_CS64(k) "v_cmp_gt_f32 s[40:41], %" #k ", %8\n\tv_cndmask_b32 %" #k ", %8, %" #k ", s[40:41]\n\t"
DEFINE_BENCH(bench_pair_cmp_cndmask_sgpr, float,
             PI_CS64(0) PI_CS64(1) PI_CS64(2) PI_CS64(3))
DEFINE_BENCH(bench_mix_fma_snop, float,
             PI_I(0, "v_fma_f32 ", ", %8, %9, ", "") "s_nop 0\n\t"
             PI_I(2, "v_fma_f32 ", ", %8, %9, ", "") "s_nop 0\n\t"
             PI_I(4, "v_fma_f32 ", ", %8, %9, ", "") "s_nop 0\n\t"
             PI_I(6, "v_fma_f32 ", ", %8, %9, ", "") "s_nop 0\n\t")

// ---- mixed and clustered streams ----------------------------------------------
// Does an "other"-class instruction (4.15 cycles alone) hide under fp32 fma-class ones (2.3 cycles) only
// when the two classes ALTERNATE, or also at the sweep kernels' real ratio and when each class comes
// in runs, as compiled code has them (cell search: compares/conversions/selects in a row; corner
// weights and the fmaf chain: multiplies and fmas in a row)?  F = v_fma_f32, O = v_max_f32,
// P = v_cmp + v_cndmask pair (2 other-class instructions), C = v_cvt_i32_f32, T = v_rcp_f32; eight
// independent accumulators, cycled, so every stream has the same instruction-level parallelism.
#define F_(k) "v_fma_f32 %" #k ", %8, %9, %" #k "\n\t"
#define O_(k) "v_max_f32 %" #k ", %8, %" #k "\n\t"
#define C_(k) "v_cvt_i32_f32 %" #k ", %" #k "\n\t"
#define T_(k) "v_rcp_f32 %" #k ", %" #k "\n\t"
#define F8 F_(0) F_(1) F_(2) F_(3) F_(4) F_(5) F_(6) F_(7)
#define O8 O_(0) O_(1) O_(2) O_(3) O_(4) O_(5) O_(6) O_(7)
#define C8 C_(0) C_(1) C_(2) C_(3) C_(4) C_(5) C_(6) C_(7)
#define FFO8 F_(0) F_(1) O_(2) F_(3) F_(4) O_(5) F_(6) F_(7)        /* 6 F + 2 O */
#define FFO8b O_(0) F_(1) F_(2) O_(3) F_(4) F_(5) O_(6) F_(7)       /* 5 F + 3 O */
// 2:1 like the evaluation sweep (539 fma-class : 257 other : 9 transcendental per wave), interleaved F F O
DEFINE_BENCH64(bench_ratio21_interleaved, FFO8 FFO8b FFO8 FFO8b FFO8 FFO8b FFO8 FFO8b)          // 44 F + 20 O
// the same 44 : 20, each class in one run
DEFINE_BENCH64(bench_ratio21_one_run, F8 F8 F8 F8 F8 F_(0) F_(1) F_(2) F_(3) O_(4) O_(5) O_(6) O_(7) O8 O8)
// the same 44 : 20 in runs of 8 other-class instructions: 16 F 8 O 16 F 8 O 12 F 4 O
DEFINE_BENCH64(bench_ratio21_runs8, F8 F8 O8 F8 F8 O8 F8 F_(0) F_(1) F_(2) F_(3) O_(4) O_(5) O_(6) O_(7))
// 1:1 in runs of 8 / 16 / 32 (the alternating row above is runs of 1)
DEFINE_BENCH64(bench_half_runs8, F8 O8 F8 O8 F8 O8 F8 O8)
DEFINE_BENCH64(bench_half_runs16, F8 F8 O8 O8 F8 F8 O8 O8)
DEFINE_BENCH64(bench_half_runs32, F8 F8 F8 F8 O8 O8 O8 O8)
// 2:1 with the other class made of conversions instead of max
DEFINE_BENCH64(bench_ratio21_cvt_runs8, F8 F8 C8 F8 F8 C8 F8 F_(0) F_(1) F_(2) F_(3) C_(4) C_(5) C_(6) C_(7))
// the evaluation sweep's mix with its transcendentals: 43 F + 20 O + 1 T, interleaved
DEFINE_BENCH64(bench_ratio_eval_interleaved,
               FFO8 FFO8b FFO8 FFO8b FFO8 FFO8b FFO8 O_(0) F_(1) F_(2) O_(3) F_(4) F_(5) O_(6) T_(7))

struct Row {
    const char* name;
    void (*fn)(Stamp*, float*, int);
};

int main(int argc, char** argv) {
    int iters = argc > 1 ? std::atoi(argv[1]) : 2048;
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    std::printf("# device %s, %d CUs, clock %d kHz; %d iterations x %d unrolled instructions per wave\n",
                prop.gcnArchName, cus, prop.clockRate, iters, kUnroll);
    std::printf("# blocks/CU = waves per SIMD (256-thread blocks, one wave per SIMD each) when the\n"
                "# dispatcher spreads them evenly; `share` is the measured median number of waves that\n"
                "# really sat on one SIMD; cyc/instr(SIMD) = median wave cycles / (instr x share).\n");
#define X_ROW(NAME, T, A, B, C) {#NAME, bench_##NAME},
    const Row rows[] = {ROWS(X_ROW){"v_fma_f32 dependent chain", bench_dep_v_fma_f32},
                        {"v_mul_f32 dependent chain", bench_dep_v_mul_f32},
                        {"v_max_f32 dependent chain", bench_dep_v_max_f32},
                        {"mix fma,max alternating", bench_mix_fma_max},
                        {"mix fma,s_nop alternating", bench_mix_fma_snop},
                        {"pair v_cmp+v_cndmask (vcc)", bench_pair_cmp_cndmask},
                        {"pair v_cmp+v_cndmask (sgpr)", bench_pair_cmp_cndmask_sgpr},
                        {"44 fma : 20 max, interleaved", bench_ratio21_interleaved},
                        {"44 fma : 20 max, runs of 8 max", bench_ratio21_runs8},
                        {"44 fma : 20 max, one run each", bench_ratio21_one_run},
                        {"44 fma : 20 cvt, runs of 8 cvt", bench_ratio21_cvt_runs8},
                        {"43 fma : 20 max : 1 rcp, interl.", bench_ratio_eval_interleaved},
                        {"32 fma : 32 max, runs of 8", bench_half_runs8},
                        {"32 fma : 32 max, runs of 16", bench_half_runs16},
                        {"32 fma : 32 max, runs of 32", bench_half_runs32}};
    float* sink;
    CHECK(hipMalloc((void**)&sink, 4));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    std::printf("%-32s %9s %6s %14s %12s %12s %10s\n", "instruction", "blocks/CU", "share",
                "cyc/instr(wave)", "cyc/instr(SIMD)", "chip Ginstr/s", "clock GHz");
    const char* only = argc > 2 ? argv[2] : nullptr;        // optional: run only the rows whose name contains this
    for (const Row& row : rows) {
        if (only && !std::strstr(row.name, only)) continue;
        for (int per_cu : {1, 2, 4, 8}) {
            const int blocks = cus * per_cu, waves = blocks * 4;
            Stamp* d;
            CHECK(hipMalloc((void**)&d, sizeof(Stamp) * waves));
            row.fn<<<blocks, 256>>>(d, sink, 16);       // warm-up (code, clocks)
            CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0));
            row.fn<<<blocks, 256>>>(d, sink, iters);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms = 0;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            std::vector<Stamp> h(waves);
            CHECK(hipMemcpy(h.data(), d, sizeof(Stamp) * waves, hipMemcpyDeviceToHost));
            CHECK(hipFree(d));
            // HW_ID (gfx9): [3:0] wave, [5:4] simd, [11:8] cu, [12] sh, [15:13] se; plus XCC id
            std::map<unsigned int, int> per_simd;
            std::vector<double> cyc, clk;
            for (const Stamp& s : h) {
                unsigned int key = (s.pad << 16) | (s.hw_id & 0xFF30u);   // xcc | se,sh,cu,simd
                per_simd[key]++;
                cyc.push_back((double)(s.t1 - s.t0));
                if (s.r1 > s.r0) clk.push_back((double)(s.t1 - s.t0) / (double)(s.r1 - s.r0) * 0.1);
            }
            std::vector<int> shares;
            for (auto& kv : per_simd) shares.push_back(kv.second);
            std::sort(shares.begin(), shares.end());
            std::sort(cyc.begin(), cyc.end());
            std::sort(clk.begin(), clk.end());
            const double instr = (double)iters * kUnroll;
            const double med = cyc[cyc.size() / 2];
            const int share = shares[shares.size() / 2];
            const double ghz = clk.empty() ? 0.0 : clk[clk.size() / 2];
            std::printf("%-32s %9d %3d-%-3d %14.2f %12.2f %12.1f %10.2f\n", row.name, per_cu,
                        shares.front(), shares.back(), med / instr, med / (instr * share),
                        instr * waves / (ms * 1e-3) / 1e9, ghz);
        }
    }
    std::printf("# peak if 2 cyc/instr: %d CUs x 4 SIMDs x 2.4 GHz / 2 = %.1f G wave-instr/s\n", cus,
                cus * 4 * 2.4 / 2);
    return 0;
}
