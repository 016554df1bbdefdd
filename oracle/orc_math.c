/*
 * orc_math.c -- ORACLE (test infrastructure only): scalar math primitives.
 *
 * Build with -O2 -ffp-contract=off -fno-fast-math: every floating-point operation below is one
 * IEEE-754 operation in the written order.  The HIP kernels implement the same operation
 * sequences independently (eorb_slam_amd/csrc/dev_math.h); neither includes the other.
 *
 * libm restatements (parity unpinned against the reference's own binary; pinned against the host
 * glibc by tests/test_oracle_math.py and oracle/check_libm.c):
 *   expf  : glibc >= 2.28 sysdeps/ieee754/flt-32/e_expf.c (ARM optimized-routines algorithm):
 *           N=32 table of 2^(i/32), degree-3 polynomial, all in double, one final rounding.
 *           Evaluated WITHOUT fused multiply-add (== glibc's generic / sse2 ifunc variant).
 *           Host glibc 2.35 (fma ifunc variant) agrees on all but 1 of 307 232 769 floats in
 *           [-104, -2^-30] (x=-0x1.f8cbb2p+5, 1 ulp; glibc-fma is the one not correctly rounded
 *           there) and on every float in [-32, 0).
 *   sinf/cosf : glibc >= 2.28 s_sinf.c / s_cosf.c / sincosf.h (fast path |x| < 120).
 *           Host glibc 2.35 agrees on all 189 792 257 floats in [2^-20, 6.5].
 * OpenCV restatements: cvRound (round half to even), cv::fastAtan2 (mathfuncs_core, 3.4.x).
 */
#include "eorb_oracle.h"
#include <math.h>
#include <string.h>

static inline uint64_t asu64(double d) { uint64_t u; memcpy(&u, &d, 8); return u; }
static inline double   asf64(uint64_t u) { double d; memcpy(&d, &u, 8); return d; }
static inline uint32_t asu32(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

int orc_cvround(double v) { return (int)lrint(v); }   /* default rounding mode: nearest-even */

/* tab[i] = bits(2^(i/32)) - (i << 47) */
static const uint64_t k_exp2_tab[32] = {
    0x3ff0000000000000ull, 0x3fefd9b0d3158574ull, 0x3fefb5586cf9890full, 0x3fef9301d0125b51ull,
    0x3fef72b83c7d517bull, 0x3fef54873168b9aaull, 0x3fef387a6e756238ull, 0x3fef1e9df51fdee1ull,
    0x3fef06fe0a31b715ull, 0x3feef1a7373aa9cbull, 0x3feedea64c123422ull, 0x3feece086061892dull,
    0x3feebfdad5362a27ull, 0x3feeb42b569d4f82ull, 0x3feeab07dd485429ull, 0x3feea47eb03a5585ull,
    0x3feea09e667f3bcdull, 0x3fee9f75e8ec5f74ull, 0x3feea11473eb0187ull, 0x3feea589994cce13ull,
    0x3feeace5422aa0dbull, 0x3feeb737b0cdc5e5ull, 0x3feec49182a3f090ull, 0x3feed503b23e255dull,
    0x3feee89f995ad3adull, 0x3feeff76f2fb5e47ull, 0x3fef199bdd85529cull, 0x3fef3720dcef9069ull,
    0x3fef5818dcfba487ull, 0x3fef7c97337b9b5full, 0x3fefa4afa2a490daull, 0x3fefd0765b6e4540ull,
};

float orc_expf(float x)
{
    const double N = 32.0;
    const double InvLn2N = 0x1.71547652b82fep+0 * N;
    const double Shift = 0x1.8p+52;
    const double C0 = 0x1.c6af84b912394p-5 / N / N / N;
    const double C1 = 0x1.ebfce50fac4f3p-3 / N / N;
    const double C2 = 0x1.62e42ff0c52d6p-1 / N;
    if (x != x) return x;
    if (x > 0x1.62e42ep6f) return INFINITY;          /* overflow (never reached on this path) */
    if (x < -0x1.9fe368p6f) return 0.0f;             /* underflow to zero */
    double xd = (double)x;
    double z = InvLn2N * xd;
    double kd = z + Shift;
    uint64_t ki = asu64(kd);
    kd -= Shift;
    double r = z - kd;
    uint64_t t = k_exp2_tab[ki % 32];
    t += ki << (52 - 5);
    double s = asf64(t);
    z = C0 * r + C1;
    double r2 = r * r;
    double y = C2 * r + 1.0;
    y = z * r2 + y;
    y = y * s;
    return (float)y;
}

/* sincosf.h / s_sincosf_data.c */
static const double k_hpi_inv = 0x1.45F306DC9C883p+23;   /* 2/pi * 2^24 */
static const double k_hpi     = 0x1.921FB54442D18p0;
static const double k_c0 = 0x1p0, k_c1 = -0x1.ffffffd0c621cp-2, k_c2 = 0x1.55553e1068f19p-5,
                    k_c3 = -0x1.6c087e89a359dp-10, k_c4 = 0x1.99343027bf8c3p-16;
static const double k_s1 = -0x1.555545995a603p-3, k_s2 = 0x1.1107605230bc4p-7,
                    k_s3 = -0x1.994eb3774cf24p-13;
static const double k_sign[4] = { 1.0, -1.0, -1.0, 1.0 };

/* neg selects the second coefficient table (cosine coefficients negated) */
static double sincos_poly(double x, double x2, int neg, int n)
{
    if ((n & 1) == 0) {
        double x3 = x * x2;
        double s1 = k_s2 + x2 * k_s3;
        double x7 = x3 * x2;
        double s = x + x3 * k_s1;
        return s + x7 * s1;
    } else {
        double sg = neg ? -1.0 : 1.0;           /* exact sign flips */
        double x4 = x2 * x2;
        double c2 = sg * k_c3 + x2 * (sg * k_c4);
        double c1 = sg * k_c0 + x2 * (sg * k_c1);
        double x6 = x4 * x2;
        double c = c1 + x4 * (sg * k_c2);
        return c + x6 * c2;
    }
}

static double reduce_fast(double x, int* np)
{
    double r = x * k_hpi_inv;
    int n = ((int32_t)r + 0x800000) >> 24;
    *np = n;
    return x - (double)n * k_hpi;
}

float orc_sinf(float y)
{
    double x = (double)y;
    uint32_t top = (asu32(y) >> 20) & 0x7ff;
    if (top < 0x3f4) {                       /* abstop12(y) < abstop12(pi/4) */
        double s = x * x;
        if (top < 0x398) return y;           /* |y| < 2^-12 */
        return (float)sincos_poly(x, s, 0, 0);
    }
    /* |y| < 120 fast path (oracle domain: [0, 2*pi]) */
    int n;
    x = reduce_fast(x, &n);
    double s = k_sign[n & 3];
    return (float)sincos_poly(x * s, x * x, (n & 2) != 0, n);
}

float orc_cosf(float y)
{
    double x = (double)y;
    uint32_t top = (asu32(y) >> 20) & 0x7ff;
    if (top < 0x3f4) {
        double s = x * x;
        if (top < 0x398) return 1.0f;
        return (float)sincos_poly(x, s, 0, 1);
    }
    int n;
    x = reduce_fast(x, &n);
    double s = k_sign[n & 3];
    return (float)sincos_poly(x * s, x * x, (n & 2) != 0, n ^ 1);
}

/* cv::fastAtan2 (OpenCV 3.4 modules/core/src/mathfuncs_core.simd.hpp atan_f32), SURVEY App.B H10 */
float orc_fast_atan2(float y, float x)
{
    const float scale = (float)(180.0 / 3.1415926535897932384626433832795);
    const float p1 = 0.9997878412794807f * scale;
    const float p3 = -0.3258083974640975f * scale;
    const float p5 = 0.1555786518463281f * scale;
    const float p7 = -0.04432655554792128f * scale;
    const float eps = (float)2.2204460492503131e-16;   /* (float)DBL_EPSILON */
    float ax = fabsf(x), ay = fabsf(y);
    float a, c, c2;
    if (ax >= ay) {
        c = ay / (ax + eps);
        c2 = c * c;
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    } else {
        c = ax / (ay + eps);
        c2 = c * c;
        a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

uint64_t orc_math_hash(int which, uint32_t lo_bits, uint32_t hi_bits)
{
    uint64_t h = 0;
    for (uint64_t u = lo_bits; u <= hi_bits; u++) {
        uint32_t ub = (uint32_t)u; float x; memcpy(&x, &ub, 4);
        float y = which == 0 ? orc_expf(-x) : (which == 1 ? orc_sinf(x) : orc_cosf(x));
        uint32_t yb; memcpy(&yb, &y, 4);
        h += (((uint64_t)ub * 0x9E3779B97F4A7C15ull) ^ (uint64_t)yb) * 0xC2B2AE3D27D4EB4Full;
    }
    return h;
}

/* ---- double sin/cos (fdlibm k_sin / k_cos + medium-range pi/2 reduction) ------------------------------------------- */
static double k_sin(double x, double y, int iy)
{
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04,
                 S4 = 2.75573137070700676789e-06, S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    double z = x * x;
    double v = z * x;
    double r = S2 + z * (S3 + z * (S4 + z * (S5 + z * S6)));
    if (iy == 0) return x + v * (S1 + z * r);
    return x - ((z * (0.5 * y - v * r) - y) - v * S1);
}
static double k_cos(double x, double y)
{
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05,
                 C4 = -2.75573143513906633035e-07, C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    double z = x * x;
    double r = z * (C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6)))));
    double ax = fabs(x);
    if (ax < 0.3) return 1.0 - (0.5 * z - (z * r - x * y));
    double qx = (ax > 0.78125) ? 0.28125 : 0.25 * ax;      /* fdlibm takes x/4 with the low word cleared */
    {
        uint64_t u = asu64(qx); u &= 0xffffffff00000000ull; qx = (ax > 0.78125) ? 0.28125 : asf64(u);
    }
    double hz = 0.5 * z - qx;
    double a = 1.0 - qx;
    return a - (hz - (z * r - x * y));
}
static int rem_pio2_medium(double x, double* y0, double* y1)
{
    const double invpio2 = 6.36619772367581382433e-01, pio2_1 = 1.57079632673412561417e+00, pio2_1t = 6.07710050650619224932e-11;
    double t = fabs(x);
    int n = (int)(t * invpio2 + 0.5);
    double fn = (double)n;
    double r = t - fn * pio2_1;
    double w = fn * pio2_1t;
    double a = r - w;
    double b = (r - a) - w;
    if (x < 0) { *y0 = -a; *y1 = -b; return -n; }
    *y0 = a; *y1 = b; return n;
}
double orc_dsin(double x)
{
    if (fabs(x) <= 0.78539816339744830962) return k_sin(x, 0.0, 0);
    double y0, y1;
    int n = rem_pio2_medium(x, &y0, &y1);
    switch (n & 3) {
        case 0: return k_sin(y0, y1, 1);
        case 1: return k_cos(y0, y1);
        case 2: return -k_sin(y0, y1, 1);
        default: return -k_cos(y0, y1);
    }
}
double orc_dcos(double x)
{
    if (fabs(x) <= 0.78539816339744830962) return k_cos(x, 0.0);
    double y0, y1;
    int n = rem_pio2_medium(x, &y0, &y1);
    switch (n & 3) {
        case 0: return k_cos(y0, y1);
        case 1: return -k_sin(y0, y1, 1);
        case 2: return -k_cos(y0, y1);
        default: return k_sin(y0, y1, 1);
    }
}
