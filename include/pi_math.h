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
#define PI_MATH_BITS_F2U(f) __float_as_uint(f)
#define PI_MATH_BITS_U2F(u) __uint_as_float(u)
#else
#include <math.h>
#include <stdint.h>
#include <string.h>
#define PI_MATH_FN static inline
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
PI_MATH_FN float pi_fmodf(float x, float y) {
    unsigned int ux = PI_MATH_BITS_F2U(x), uy = PI_MATH_BITS_F2U(y);
    unsigned int sx = ux & 0x80000000u;
    unsigned int ax = ux & 0x7FFFFFFFu, ay = uy & 0x7FFFFFFFu;
    if (ay == 0u || ax >= 0x7F800000u || ay > 0x7F800000u) {
        float t = x * y;            /* y == 0, x Inf/NaN, y NaN -> NaN */
        return t / t;
    }
    if (ax < ay) return x;
    if (ax == ay) return PI_MATH_BITS_U2F(sx);          /* +-0 */
    {
        float fx = PI_MATH_BITS_U2F(ax), fy = PI_MATH_BITS_U2F(ay);
        if (ay < 0x7F000000u && fx <= 2.0f * fy) {      /* y <= x <= 2y */
            float r = fx - fy;                          /* exact */
            if (r >= fy) r = r - fy;                    /* only when fx == 2fy */
            return PI_MATH_BITS_U2F(PI_MATH_BITS_F2U(r) | sx);
        }
    }
    {
        int ex = (int)(ax >> 23), ey = (int)(ay >> 23);
        unsigned int mx, my;
        if (ex == 0) { mx = ax; ex = 1; while ((mx & 0x00800000u) == 0u) { mx <<= 1; ex--; } }
        else mx = (ax & 0x007FFFFFu) | 0x00800000u;
        if (ey == 0) { my = ay; ey = 1; while ((my & 0x00800000u) == 0u) { my <<= 1; ey--; } }
        else my = (ay & 0x007FFFFFu) | 0x00800000u;
        for (; ex > ey; ex--) {
            if (mx >= my) mx -= my;
            mx <<= 1;
        }
        if (mx >= my) mx -= my;
        if (mx == 0u) return PI_MATH_BITS_U2F(sx);
        while ((mx & 0x00800000u) == 0u) { mx <<= 1; ex--; }
        if (ex > 0) mx = (mx & 0x007FFFFFu) | ((unsigned int)ex << 23);
        else mx >>= (unsigned int)(1 - ex);
        return PI_MATH_BITS_U2F(mx | sx);
    }
}

#endif /* PI_MATH_H_ */
