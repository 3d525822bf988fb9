/*
 * pi_math.h — deterministic single-precision sinf / cosf / fmodf.
 *
 * Why this file exists
 * --------------------
 * The Bellman-backup kernels inline a user-supplied `step_dynamics` C string
 * (reference plugin surface: src/cuda_policy_iteration.py:113-125).  Every env the
 * reference ships calls only sinf, cosf, fmodf, fabsf, fmaxf, fminf
 * (runners/[name]_cuda.py dynamics strings).  fabsf/fmaxf/fminf/+,-,*,/ and fmaf are
 * IEEE-exact on both gfx950 and x86-64, but sinf/cosf come from three different
 * libraries (CUDA libdevice on the reference's GPU, ROCm ocml here, glibc in a
 * CPU checker) that differ in the last ulp, which is enough to flip a greedy
 * argmax on a near-tie.  To make "HIP kernel == CPU oracle" a
 * bit-exact statement, the kernel preamble and the oracle both compile THIS
 * header and `#define sinf pi_sinf`, etc.  Every operation below is a single
 * IEEE-754 binary32 operation (mul, add, fma, round-to-nearest-even, int
 * conversion) or integer arithmetic, so with floating-point contraction off
 * (-ffp-contract=off on both compilers) the results are identical bit for bit
 * on the CPU and on the GPU.
 *
 * Accuracy (checked in tests/test_pi_math.py against float64 libm):
 *   pi_sinf / pi_cosf : <= 2 ulp for |x| <= 1e5 (0.3 ulp on average); larger |x|
 *                       goes through a float64 Cody-Waite reduction and stays
 *                       <= 2 ulp to |x| <= 2^30; beyond that the result is
 *                       deterministic and in [-1, 1] but loses accuracy.
 *                       sin(0) = 0, cos(0) = 1 exactly.  NaN/Inf -> NaN.
 *   pi_fmodf          : exact (the IEEE remainder-toward-zero is exactly
 *                       representable), bit-identical to glibc fmodf.
 *
 * The header is plain C99/C++ and HIP device code at once: PI_MATH_FN expands
 * to `__device__ __forceinline__` under hipcc/hipRTC and `static inline` on
 * the host.
 */
#ifndef PI_MATH_H_
#define PI_MATH_H_

#if defined(__HIPCC_RTC__) || defined(__HIP_DEVICE_COMPILE__) || defined(__HIPCC__)
#define PI_MATH_FN __device__ __forceinline__
#define PI_MATH_LDEXP(v, e) ldexpf((v), (e))
#define PI_MATH_BITS_F2U(f) __float_as_uint(f)
#define PI_MATH_BITS_U2F(u) __uint_as_float(u)
#else
#include <math.h>
#include <stdint.h>
#include <string.h>
#define PI_MATH_FN static inline
#define PI_MATH_LDEXP(v, e) ldexpf((v), (e))
static inline unsigned int pi__f2u(float f) { unsigned int u; memcpy(&u, &f, 4); return u; }
static inline float pi__u2f(unsigned int u) { float f; memcpy(&f, &u, 4); return f; }
#define PI_MATH_BITS_F2U(f) pi__f2u(f)
#define PI_MATH_BITS_U2F(u) pi__u2f(u)
#endif

/*
 * sinf / cosf for an fp32-issue-bound GPU kernel.  On MI355X a wave64 v_fma/v_mul/v_add costs
 * ~2.3 cycles of SIMD issue but v_cndmask / v_cmp / v_cvt / v_rndne / shifts cost ~4.2
 * (tools/valu_issue_bench.hip), so the classic "reduce to |r| <= pi/4, evaluate the sine AND
 * the cosine polynomial, select by quadrant" scheme pays more for its selects and conversions
 * than for its arithmetic.  This version uses ONE odd polynomial on |r| <= pi/2 and no select:
 *     sin(x) = (-1)^k sin(r),  r = x - k pi,            k = nearest integer to x / pi
 *     cos(x) = (-1)^k sin(r),  r = (k + 1/2) pi - x,    k = nearest integer to x / pi - 1/2
 * k is taken with the add-a-magic-constant trick (the float 1.5 * 2^23 + k has k in its low
 * mantissa bits, so (-1)^k is one shift of that float's bit pattern — no float->int conversion),
 * r by a three-piece Cody-Waite subtraction with fused multiply-adds (exact products), and
 * sin(r) = r + (r z) S1 + (r z z) P(z), z = r^2, with the -r^3/6 term added on its own so that
 * the end points come out exact (sin(pi/2) = cos(0) = 1).
 */
#define PI__PI_HI 3.14159274101257324219f            /* 0x40490FDB */
#define PI__PI_MID (-8.74227765734758577e-08f)       /* 0xB3BBBD2E */
#define PI__PI_LO (-3.43024902001176374e-15f)        /* 0xA7772CED */
#define PI__PIO2_HI 1.57079637050628662109375f       /* 0x3FC90FDB */
#define PI__PIO2_MID (-4.3711388286737928865e-08f)   /* 0xB33BBD2E */
#define PI__PIO2_LO (-1.7151245100058818728e-15f)    /* 0xA6F72CED */
#define PI__ONE_OVER_PI 0.318309873342514038085938f  /* 0x3EA2F983 */
#define PI__MAGIC 12582912.0f                        /* 1.5 * 2^23: ulp = 1 */
#define PI__FAST_MAX 1.0e5f                          /* |x| beyond this: float64 reduction */

/* sin(r) for |r| <= pi/2 (+ a little): degree-11 odd polynomial (fit error 0.002 ulp). */
PI_MATH_FN float pi__sin_poly(float r) {
    float z = r * r;
    float p = fmaf(z, -2.4070633486417137e-08f, 2.753604803729104e-06f);
    p = fmaf(z, p, -1.9841080938931555e-04f);
    p = fmaf(z, p, 8.33333283662796e-03f);
    float w = r * z;
    float a = fmaf(w, -1.666666716337204e-01f, r);
    return fmaf(w * z, p, a);
}

/* Rare: |x| > PI__FAST_MAX, Inf or NaN.  float64 Cody-Waite; `half` = 0 (sine) or 0.5 (cosine).
 * Returns r (negated for the cosine, see above) and the sign bit of (-1)^k. */
PI_MATH_FN float pi__reduce_big(float x, double half, unsigned int* sign) {
    *sign = 0u;
    if (!(fabsf(x) < 3.0e38f)) return x - x;          /* Inf, NaN -> NaN */
    {
        double xd = (double)x;
        double k = rint(xd * 0.31830988618379067 - half);
        double h = k + half;
        double r = fma(-h, 3.141592653589793, xd);
        r = fma(-h, 1.2246467991473532e-16, r);
        double odd = k - 2.0 * rint(k * 0.5);          /* -1, 0 or 1 without overflowing an int */
        if (odd != 0.0) *sign = 0x80000000u;
        return (float)(half != 0.0 ? -r : r);
    }
}

PI_MATH_FN float pi_sinf(float x) {
    float r;
    unsigned int sign;
    if (fabsf(x) <= PI__FAST_MAX) {
        float t = fmaf(x, PI__ONE_OVER_PI, PI__MAGIC);
        float k = t - PI__MAGIC;
        r = fmaf(-k, PI__PI_HI, x);
        r = fmaf(-k, PI__PI_MID, r);
        r = fmaf(-k, PI__PI_LO, r);
        sign = PI_MATH_BITS_F2U(t) << 31;
    } else {
        r = pi__reduce_big(x, 0.0, &sign);
    }
    return PI_MATH_BITS_U2F(PI_MATH_BITS_F2U(pi__sin_poly(r)) ^ sign);
}

PI_MATH_FN float pi_cosf(float x) {
    float r;
    unsigned int sign;
    if (fabsf(x) <= PI__FAST_MAX) {
        float t = fmaf(x, PI__ONE_OVER_PI, -0.5f) + PI__MAGIC;
        float k = t - PI__MAGIC;
        float h = fmaf(2.0f, k, 1.0f);                  /* 2k + 1: r = h pi/2 - x */
        r = fmaf(h, PI__PIO2_HI, -x);
        r = fmaf(h, PI__PIO2_MID, r);
        r = fmaf(h, PI__PIO2_LO, r);
        sign = PI_MATH_BITS_F2U(t) << 31;
    } else {
        r = pi__reduce_big(x, 0.5, &sign);
    }
    return PI_MATH_BITS_U2F(PI_MATH_BITS_F2U(pi__sin_poly(r)) ^ sign);
}

/*
 * Exact fmodf: result has the sign of x and magnitude < |y|.  Fast exits cover
 * |x| < |y| and |y| <= |x| <= 2|y| (Sterbenz: the subtraction is exact), which
 * is every angle-wrap call in the reference envs; the general path is the
 * classic shift-and-subtract on the integer significands.
 */
/*
 * Exact fmodf without loops or calls, so that a loop-invariant angle wrap — and everything the
 * env computes from it — can be hoisted out of the per-action loop of the improvement sweep
 * (control flow with loops in it pins a value inside the loop it is computed in).
 *
 * One reduction step: for finite 0 < b and a < 2^23 * b, q = trunc(RN(a / b)) is the true
 * integer quotient or one more; a - q*b is then exactly representable (|.| < b, a multiple of
 * ulp(b)) so the fma returns it exactly, and one conditional +b repairs the "one more" case.
 * |x| / |y| can reach 2^277, so up to 12 steps peel 23 bits each, from b * 2^(23*11) downwards;
 * all but the last are skipped (selects) for ordinary arguments.
 */
PI_MATH_FN float pi__fmod_step(float a, float b) {
    float q = truncf(a / b);
    float r = fmaf(-q, b, a);
    r = (r < 0.0f) ? r + b : r;
    r = (r >= b) ? r - b : r;
    return r;
}

PI_MATH_FN float pi_fmodf(float x, float y) {
    const unsigned int ux = PI_MATH_BITS_F2U(x), uy = PI_MATH_BITS_F2U(y);
    const unsigned int sx = ux & 0x80000000u;
    const unsigned int ax = ux & 0x7FFFFFFFu, ay = uy & 0x7FFFFFFFu;
    const float fx = PI_MATH_BITS_U2F(ax), fy = PI_MATH_BITS_U2F(ay);
    /* common case (every angle wrap): |y| a normal number below 2^127 and |x| < 2|y| (false for a
     * NaN/Inf operand): the result is |x| itself or |x| - |y|, which is exact (Sterbenz) */
    if ((ay - 0x00800000u < 0x7E800000u) & (fx < 2.0f * fy)) {
        const float r = fx - (fx >= fy ? fy : 0.0f);
        return PI_MATH_BITS_U2F(PI_MATH_BITS_F2U(r) | sx);
    }
    if ((ay == 0u) | (ax >= 0x7F800000u) | (ay > 0x7F800000u)) {
        const float t = x * y;                          /* y == 0, x Inf/NaN, y NaN -> NaN */
        return t / t;
    }
    if ((ay == 0x7F800000u) | (ax < ay)) return x;      /* fmod(finite, Inf) = x; |x| < |y| */
    {
        /* peel 23 quotient bits per step, largest block first; a block |y| * 2^(23 j) that
         * overflows or exceeds what is left is skipped (every operation is exact, subnormal
         * divisors included) */
        float as = fx;
#pragma unroll
        for (int j = 11; j >= 0; --j) {
            const float blk = PI_MATH_LDEXP(fy, 23 * j);
            const int use = (blk <= as) & (blk <= 3.4028234663852886e38f);
            const float red = pi__fmod_step(as, use ? blk : 1.0f);
            as = use ? red : as;
        }
        return PI_MATH_BITS_U2F(PI_MATH_BITS_F2U(as) | sx);
    }
}

#endif /* PI_MATH_H_ */
