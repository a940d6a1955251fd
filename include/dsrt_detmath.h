/*
 * dsrt_detmath.h -- deterministic sin / cos / pow, identical bit-for-bit on gfx950 and on x86-64.
 *
 * Why this exists: every sample of a pixel shares one LCG stream
 * (src/gpu_render.cu:990-999), so one last-ulp difference in cosf/sinf/powf flips a branch
 * somewhere and de-synchronises every later sample of that pixel.  The reference calls the CUDA
 * libdevice cosf/sinf (src/gpu_render.cu:104-106, 157-158) and powf (:211, :1019-1021); ROCm's
 * ocml and glibc's libm each round differently in the last place, and none of the three is
 * available on both sides of our parity check.  These functions are built only from IEEE-754
 * +, -, *, /, sqrt and explicit fma, which are correctly rounded on both targets, so the HIP
 * kernel and the CPU oracle agree exactly provided both are compiled with -ffp-contract=off.
 * Accuracy: sin/cos <= ~1.5 ulp on |x| <= 64 (the kernel's argument is phi in [0, 2*pi));
 * pow is computed in double and rounded once, i.e. correctly rounded except with probability
 * ~1e-8.  They are NOT bit-identical to libdevice, ocml or glibc: against any of those the
 * comparison is statistical, and every parity report says so.
 *
 * Used by: the HIP kernels (deep-space-ray-tracer_amd/csrc) and the CPU oracle (oracle/).
 */
#ifndef DSRT_DETMATH_H
#define DSRT_DETMATH_H

#include <stdint.h>
#if defined(__cplusplus)
#include <cmath>
#else
#include <math.h>
#endif

#if defined(__HIPCC__)
#define DSRT_HD __host__ __device__ static inline
#else
#define DSRT_HD static inline
#endif

DSRT_HD uint32_t dsrt_f2u(float f) { uint32_t u; __builtin_memcpy(&u, &f, 4); return u; }
DSRT_HD float    dsrt_u2f(uint32_t u) { float f; __builtin_memcpy(&f, &u, 4); return f; }
DSRT_HD uint64_t dsrt_d2u(double d) { uint64_t u; __builtin_memcpy(&u, &d, 8); return u; }
DSRT_HD double   dsrt_u2d(uint64_t u) { double d; __builtin_memcpy(&d, &u, 8); return d; }

/* Reduce x to r in [-pi/4, pi/4] and quadrant q = k mod 4, x = k*pi/2 + r.
 * Cody-Waite with pi/2 split in three parts; each partial product k*part is exact for |k| < 2^8. */
DSRT_HD float dsrt_reduce_pio2(float x, int* q) {
    const float TWO_OVER_PI = 0.636619772367581343f;
    const float P1 = 1.5703125f;                 /* 0x3FC90000: 8 significant bits        */
    const float P2 = 4.837512969970703125e-4f;   /* 0x39FDAA00: next 15 bits              */
    const float P3 = 7.54978995489188216e-8f;    /* 0x33A22168: remainder of pi/2         */
    float kf = floorf(x * TWO_OVER_PI + 0.5f);
    float r = __builtin_fmaf(-kf, P1, x);
    r = __builtin_fmaf(-kf, P2, r);
    r = __builtin_fmaf(-kf, P3, r);
    *q = ((int)kf) & 3;
    return r;
}

DSRT_HD float dsrt_sin_poly(float r) {
    float z = r * r;
    float p = -1.9515295891e-4f;
    p = __builtin_fmaf(p, z, 8.3321608736e-3f);
    p = __builtin_fmaf(p, z, -1.6666654611e-1f);
    return __builtin_fmaf(r * z, p, r);
}

DSRT_HD float dsrt_cos_poly(float r) {
    float z = r * r;
    float p = 2.443315711809948e-5f;
    p = __builtin_fmaf(p, z, -1.388731625493765e-3f);
    p = __builtin_fmaf(p, z, 4.166664568298827e-2f);
    float c = __builtin_fmaf(z * z, p, __builtin_fmaf(-0.5f, z, 1.0f));
    return c;
}

DSRT_HD float dsrt_sinf(float x) {
    int q;
    float r = dsrt_reduce_pio2(x, &q);
    float s = (q & 1) ? dsrt_cos_poly(r) : dsrt_sin_poly(r);
    return (q & 2) ? -s : s;
}

DSRT_HD float dsrt_cosf(float x) {
    int q;
    float r = dsrt_reduce_pio2(x, &q);
    float c = (q & 1) ? dsrt_sin_poly(r) : dsrt_cos_poly(r);
    return ((q + 1) & 2) ? -c : c;
}

/* log2 of a positive, finite, normal-or-subnormal double; ~1e-16 relative. */
DSRT_HD double dsrt_log2_pos(double x) {
    int e = 0;
    uint64_t u = dsrt_d2u(x);
    if ((u >> 52) == 0) { x = x * 18014398509481984.0; /* 2^54 */ u = dsrt_d2u(x); e = -54; }
    e += (int)((u >> 52) & 0x7FF) - 1023;
    u = (u & 0x000FFFFFFFFFFFFFull) | 0x3FF0000000000000ull;
    double m = dsrt_u2d(u);                       /* [1, 2) */
    if (m > 1.4142135623730951) { m = m * 0.5; e += 1; }   /* [sqrt(1/2), sqrt(2)) */
    double s = (m - 1.0) / (m + 1.0);             /* |s| <= 0.1716 */
    double z = s * s;
    /* ln(m) = 2 s (1 + z/3 + z^2/5 + ... + z^11/23) */
    double p = 1.0 / 23.0;
    p = p * z + 1.0 / 21.0;
    p = p * z + 1.0 / 19.0;
    p = p * z + 1.0 / 17.0;
    p = p * z + 1.0 / 15.0;
    p = p * z + 1.0 / 13.0;
    p = p * z + 1.0 / 11.0;
    p = p * z + 1.0 / 9.0;
    p = p * z + 1.0 / 7.0;
    p = p * z + 1.0 / 5.0;
    p = p * z + 1.0 / 3.0;
    p = p * z + 1.0;
    double ln_m = 2.0 * s * p;
    return (double)e + ln_m * 1.4426950408889634;  /* 1/ln 2 */
}

/* 2^t for |t| < 1000; ~1e-16 relative. */
DSRT_HD double dsrt_exp2_d(double t) {
    double nf = floor(t + 0.5);
    double f = (t - nf) * 0.6931471805599453;     /* |f| <= 0.3466 */
    double p = 1.0 / 6227020800.0;                /* 1/13! */
    p = p * f + 1.0 / 479001600.0;
    p = p * f + 1.0 / 39916800.0;
    p = p * f + 1.0 / 3628800.0;
    p = p * f + 1.0 / 362880.0;
    p = p * f + 1.0 / 40320.0;
    p = p * f + 1.0 / 5040.0;
    p = p * f + 1.0 / 720.0;
    p = p * f + 1.0 / 120.0;
    p = p * f + 1.0 / 24.0;
    p = p * f + 1.0 / 6.0;
    p = p * f + 0.5;
    p = p * f + 1.0;
    p = p * f + 1.0;
    int n = (int)nf;
    /* scale by 2^n through the exponent field; n is far inside the normal range for every caller */
    uint64_t bits = (uint64_t)(int64_t)(n + 1023) << 52;
    return p * dsrt_u2d(bits);
}

/* powf replacement for the kernel's two uses: schlick (y = 5, src/gpu_render.cu:211) and the
 * output gamma (y = 1/gamma, src/gpu_render.cu:1019-1021).  x < 0 returns NaN for every y that
 * is not handled by a special case below (the kernel never passes one). */
DSRT_HD float dsrt_powf(float x, float y) {
    if (y == 0.0f) return 1.0f;
    if (x != x || y != y) return x + y;
    if (y == 1.0f) return x;
    if (y == 0.5f) return (x == 0.0f) ? 0.0f : sqrtf(x);     /* sqrt is the correctly rounded x^0.5 */
    if (y == 5.0f) { double d = (double)x; double d2 = d * d; return (float)(d2 * d2 * d); }
    if (y == 2.0f) { double d = (double)x; return (float)(d * d); }
    if (x == 0.0f) return (y > 0.0f) ? 0.0f : dsrt_u2f(0x7F800000u);
    if (x < 0.0f) return dsrt_u2f(0x7FC00000u);
    if (x == 1.0f) return 1.0f;
    if (dsrt_f2u(x) == 0x7F800000u) return (y > 0.0f) ? x : 0.0f;
    double t = (double)y * dsrt_log2_pos((double)x);
    if (t > 200.0) return dsrt_u2f(0x7F800000u);
    if (t < -200.0) return 0.0f;
    return (float)dsrt_exp2_d(t);
}

#endif /* DSRT_DETMATH_H */
