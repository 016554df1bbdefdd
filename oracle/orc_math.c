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
 *   tanf  : glibc >= 2.28 s_tanf.c (|x| <= pi/4: kernel; else the double reduction of sincosf.h, the reduced argument split into
 *           float hi + lo) + k_tanf.c (fdlibm float kernel with glibc's early return for |pi/4 - x| < 2^-13).
 *           Host glibc 2.35 agrees on all 2 150 419 662 floats in [-2.35, 2.35] (the domain KannalaBrandt8::unproject needs).
 *   atanf / atan2f : glibc s_atanf.c / e_atan2f.c (fdlibm float).  Host glibc 2.35 agrees on all 2 139 095 040 positive
 *           floats (atanf is odd) and on 398 439 896 pseudo-random / structured (y, x) pairs.
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

/* the fast path is glibc's for every |x| < 120, either sign (psi = atan2f(y, x) of KannalaBrandt8::project is in [-pi, pi]) */
float orc_sinf_any(float x) { return orc_sinf(x); }
float orc_cosf_any(float x) { return orc_cosf(x); }

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

/* ---- tanf / atanf / atan2f (KannalaBrandt8::project / unproject, src/CameraModels/KannalaBrandt8.cpp:87-190) ---- */
static inline float asf32(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static const float kT[13] = {
  3.3333334327e-01f, 1.3333334029e-01f, 5.3968254477e-02f, 2.1869488060e-02f, 8.8632395491e-03f, 3.5920790397e-03f,
  1.4562094584e-03f, 5.8804126456e-04f, 2.4646313977e-04f, 7.8179444245e-05f, 7.1407252108e-05f, -1.8558637748e-05f, 2.5907305826e-05f };
static float k_tanf(float x, float y, int iy)
{
    const float pio4 = 7.8539812565e-01f, pio4lo = 3.7748947079e-08f;
    float z, r, v, w, s;
    int32_t hx = (int32_t)asu32(x), ix = hx & 0x7fffffff;
    if (ix < 0x39000000) {                                   /* |x| < 2^-13 */
        if ((int)x == 0) {
            if ((ix | (iy + 1)) == 0) return 1.0f / fabsf(x);
            else return (iy == 1) ? x : -1.0f / x;
        }
    }
    if (ix >= 0x3f2ca140) {                                  /* |x| >= 0.6744 */
        if (hx < 0) { x = -x; y = -y; }
        z = pio4 - x;
        w = pio4lo - y;
        x = z + w; y = 0.0f;
        if (fabsf(x) < 0x1p-13f) return (float)((1 - ((hx >> 30) & 2)) * iy) * (1.0f - (float)(2 * iy) * x);
    }
    z = x * x;
    w = z * z;
    r = kT[1] + w * (kT[3] + w * (kT[5] + w * (kT[7] + w * (kT[9] + w * kT[11]))));
    v = z * (kT[2] + w * (kT[4] + w * (kT[6] + w * (kT[8] + w * (kT[10] + w * kT[12])))));
    s = z * x;
    r = y + z * (s * (r + v) + y);
    r += kT[0] * s;
    w = x + r;
    if (ix >= 0x3f2ca140) {
        v = (float)iy;
        return (float)(1 - ((hx >> 30) & 2)) * (v - 2.0f * (x - (w * w / (w + v) - r)));
    }
    if (iy == 1) return w;
    {   /* -1 / (x + r), accurately */
        float a, t;
        z = asf32(asu32(w) & 0xfffff000u);
        v = r - (z - x);
        t = a = -1.0f / w;
        t = asf32(asu32(t) & 0xfffff000u);
        s = 1.0f + t * z;
        return t + a * (s + t * v);
    }
}
float orc_tanf(float x)                                      /* |x| < 120 */
{
    int32_t hx = (int32_t)asu32(x), ix = hx & 0x7fffffff;
    if (ix <= 0x3f490fda) return k_tanf(x, 0.0f, 1);
    int n;
    const double xr = reduce_fast((double)x, &n);
    const float y0 = (float)xr;
    const float y1 = (float)(xr - (double)y0);
    return k_tanf(y0, y1, 1 - ((n & 1) << 1));
}
float orc_atanf(float x)
{
    static const float atanhi[4] = {4.6364760399e-01f, 7.8539812565e-01f, 9.8279368877e-01f, 1.5707962513e+00f};
    static const float atanlo[4] = {5.0121582440e-09f, 3.7748947079e-08f, 3.4473217170e-08f, 7.5497894159e-08f};
    static const float aT[11] = {3.3333334327e-01f, -2.0000000298e-01f, 1.4285714924e-01f, -1.1111110449e-01f, 9.0908870101e-02f,
      -7.6918758452e-02f, 6.6610731184e-02f, -5.8335702866e-02f, 4.9768779427e-02f, -3.6531571299e-02f, 1.6285819933e-02f};
    float w, s1, s2, z;
    int32_t hx = (int32_t)asu32(x), ix = hx & 0x7fffffff, id;
    if (ix >= 0x4c000000) {                                  /* |x| >= 2^25 */
        if (ix > 0x7f800000) return x + x;
        return (hx > 0) ? atanhi[3] + atanlo[3] : -atanhi[3] - atanlo[3];
    }
    if (ix < 0x3ee00000) {                                   /* |x| < 0.4375 */
        if (ix < 0x31000000) return x;
        id = -1;
    } else {
        x = fabsf(x);
        if (ix < 0x3f980000) {
            if (ix < 0x3f300000) { id = 0; x = (2.0f * x - 1.0f) / (2.0f + x); }
            else { id = 1; x = (x - 1.0f) / (x + 1.0f); }
        } else {
            if (ix < 0x401c0000) { id = 2; x = (x - 1.5f) / (1.0f + 1.5f * x); }
            else { id = 3; x = -1.0f / x; }
        }
    }
    z = x * x;
    w = z * z;
    s1 = z * (aT[0] + w * (aT[2] + w * (aT[4] + w * (aT[6] + w * (aT[8] + w * aT[10])))));
    s2 = w * (aT[1] + w * (aT[3] + w * (aT[5] + w * (aT[7] + w * aT[9]))));
    if (id < 0) return x - x * (s1 + s2);
    z = atanhi[id] - ((x * (s1 + s2) - atanlo[id]) - x);
    return (hx < 0) ? -z : z;
}
float orc_atan2f(float y, float x)
{
    const float pi_o_4 = 7.8539818525e-01f, pi_o_2 = 1.5707963705e+00f, pi = 3.1415927410e+00f, pi_lo = -8.7422776573e-08f, tiny = 1.0e-30f;
    float z;
    int32_t hx = (int32_t)asu32(x), ix = hx & 0x7fffffff, hy = (int32_t)asu32(y), iy = hy & 0x7fffffff, k, m;
    if (ix > 0x7f800000 || iy > 0x7f800000) return x + y;
    if (hx == 0x3f800000) return orc_atanf(y);
    m = ((hy >> 31) & 1) | ((hx >> 30) & 2);
    if (iy == 0) { switch (m) { case 0: case 1: return y; case 2: return pi + tiny; default: return -pi - tiny; } }
    if (ix == 0) return (hy < 0) ? -pi_o_2 - tiny : pi_o_2 + tiny;
    if (ix == 0x7f800000) {
        if (iy == 0x7f800000) { switch (m) { case 0: return pi_o_4 + tiny; case 1: return -pi_o_4 - tiny;
                                            case 2: return 3.0f * pi_o_4 + tiny; default: return -3.0f * pi_o_4 - tiny; } }
        else { switch (m) { case 0: return 0.0f; case 1: return -0.0f; case 2: return pi + tiny; default: return -pi - tiny; } }
    }
    if (iy == 0x7f800000) return (hy < 0) ? -pi_o_2 - tiny : pi_o_2 + tiny;
    k = (iy - ix) >> 23;
    if (k > 60) z = pi_o_2 + 0.5f * pi_lo;
    else if (hx < 0 && k < -60) z = 0.0f;
    else z = orc_atanf(fabsf(y / x));
    switch (m) {
        case 0: return z;
        case 1: return asf32(asu32(z) ^ 0x80000000u);
        case 2: return pi - (z - pi_lo);
        default: return (z - pi_lo) - pi;
    }
}

/* the (y, x) pair number i of the atan2f self-check (device and oracle generate the same pairs) */
void orc_atan2_pair(uint64_t i, float* y, float* x)
{
    uint64_t s = (i + 1) * 0x9E3779B97F4A7C15ull;
    s ^= s >> 29; s *= 0xBF58476D1CE4E5B9ull; s ^= s >> 32;
    if (i & 1) { *y = (float)((int32_t)(s & 0xffff) - 32768) / 97.0f; *x = (float)((int32_t)((s >> 20) & 0xffff) - 32768) / 89.0f; }
    else { *y = asf32((uint32_t)s); *x = asf32((uint32_t)(s >> 32)); }
}
uint64_t orc_atan2_hash(uint64_t first, uint64_t count)
{
    uint64_t h = 0;
    for (uint64_t i = first; i < first + count; i++) {
        float y, x; orc_atan2_pair(i, &y, &x);
        if (y != y || x != x) continue;
        const uint32_t rb = asu32(orc_atan2f(y, x));
        h += ((i * 0x9E3779B97F4A7C15ull) ^ (uint64_t)rb) * 0xC2B2AE3D27D4EB4Full;
    }
    return h;
}

uint64_t orc_math_hash(int which, uint32_t lo_bits, uint32_t hi_bits)
{
    uint64_t h = 0;
    for (uint64_t u = lo_bits; u <= hi_bits; u++) {
        uint32_t ub = (uint32_t)u; float x; memcpy(&x, &ub, 4);
        float y = which == 0 ? orc_expf(-x) : (which == 1 ? orc_sinf(x) : (which == 2 ? orc_cosf(x) : (which == 3 ? orc_tanf(x) : orc_atanf(x))));
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
