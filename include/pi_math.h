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
 *   pi_sinf / pi_cosf : <= 2 ulp for |x| <= 1e5; larger |x| goes through a
 *                       float64 Cody-Waite reduction and stays <= 2 ulp to
 *                       |x| <= 2^30; beyond that the result is deterministic and
 *                       in [-1, 1] but loses accuracy.  NaN/Inf -> NaN.
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

/* pi/2 split into three binary32 pieces (hi + mid + lo == pi/2 to ~2^-75). */
#define PI__PIO2_HI 1.57079637050628662109375f       /* 0x3FC90FDB */
#define PI__PIO2_MID (-4.3711388286737928865e-08f)   /* 0xB33BBD2E */
#define PI__PIO2_LO (-1.7151245100058818728e-15f)    /* 0xA6F72CED */
#define PI__TWO_OVER_PI 0.636619746685028076171875f  /* 0x3F22F983 */

/* Reduced argument r in about [-pi/4, pi/4] and quadrant q (mod 4) of x. */
PI_MATH_FN float pi__reduce(float x, int* q) {
    float ax = fabsf(x);
    if (ax <= 1.0e5f) {
        float k = rintf(x * PI__TWO_OVER_PI);
        float r = fmaf(-k, PI__PIO2_HI, x);
        r = fmaf(-k, PI__PIO2_MID, r);
        r = fmaf(-k, PI__PIO2_LO, r);
        *q = (int)k;
        return r;
    }
    /* Rare: huge angle.  float64 two-piece Cody-Waite; also the NaN/Inf path. */
    if (!(ax < 3.0e38f)) { *q = 0; return x - x; }   /* Inf, NaN -> NaN */
    {
        double xd = (double)x;
        double k = rint(xd * 0.63661977236758138);
        double r = fma(-k, 1.5707963267948966, xd);
        r = fma(-k, 6.123233995736766e-17, r);
        /* quadrant = k mod 4, computed without overflowing an int */
        double k4 = k - 4.0 * rint(k * 0.25);     /* in [-2, 2] */
        *q = (int)k4;
        return (float)r;
    }
}

/* sin(r), cos(r) on |r| <= pi/4 (+ a little): odd/even minimax polynomials. */
PI_MATH_FN float pi__sin_poly(float r) {
    float z = r * r;
    float p = fmaf(z, -1.9515295891e-4f, 8.3321608736e-3f);
    p = fmaf(z, p, -1.6666654611e-1f);
    return fmaf(r * z, p, r);
}
PI_MATH_FN float pi__cos_poly(float r) {
    float z = r * r;
    float p = fmaf(z, 2.443315711809948e-5f, -1.388731625493765e-3f);
    p = fmaf(z, p, 4.166664568298827e-2f);
    return fmaf(z * z, p, fmaf(-0.5f, z, 1.0f));
}

PI_MATH_FN float pi_sinf(float x) {
    int q;
    float r = pi__reduce(x, &q);
    float s = pi__sin_poly(r);
    float c = pi__cos_poly(r);
    float v = (q & 1) ? c : s;
    return (q & 2) ? -v : v;
}

PI_MATH_FN float pi_cosf(float x) {
    int q;
    float r = pi__reduce(x, &q);
    float s = pi__sin_poly(r);
    float c = pi__cos_poly(r);
    float v = (q & 1) ? s : c;
    return ((q + 1) & 2) ? -v : v;
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
    /* common case (every angle wrap): 0 < |y| < 2^127 finite and |x| <= 2|y| (false for a NaN/Inf
     * operand): |x| - |y| is exact (Sterbenz); subtract once more if |x| == 2|y|; |x| < |y| -> x */
    const int common = (ay - 1u < 0x7EFFFFFFu) & (fx <= 2.0f * fy);
    float r = fx - fy;
    r = (r >= fy) ? r - fy : r;
    r = (ax < ay) ? fx : r;
    if (!common) {
        if ((ay == 0u) | (ax >= 0x7F800000u) | (ay > 0x7F800000u)) {
            const float t = x * y;                      /* y == 0, x Inf/NaN, y NaN -> NaN */
            return t / t;
        }
        if (ay == 0x7F800000u) return x;                /* fmod(finite, Inf) = x */
        if (ax >= ay) {
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
            r = as;
        }
    }
    return PI_MATH_BITS_U2F(PI_MATH_BITS_F2U(r) | sx);
}

#endif /* PI_MATH_H_ */
