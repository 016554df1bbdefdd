// dev_math.h -- device-side scalar math with a fixed IEEE-754 operation order (gfx950).
//
// The whole library is compiled with -ffp-contract=off: every * + - / below is a single
// correctly-rounded IEEE operation (f32 denormals on, IEEE f32 division), so results are
// reproducible bit for bit against any other IEEE machine evaluating the same sequence.
// expf / sinf / cosf follow the double-precision evaluation schemes glibc (>= 2.28) uses for its
// single-precision functions, which is what the reference's std::exp/cos/sin calls resolve to
// (src/Event/EventConversion.cc:59-65, src/ORBextractor.cc:117-118); fastAtan2 follows
// OpenCV 3.4's cv::fastAtan2 (call site src/ORBextractor.cc:103); cvRound = round-half-even.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace eorb {

__device__ __constant__ static const uint64_t kExp2Tab[32] = {
    0x3ff0000000000000ull, 0x3fefd9b0d3158574ull, 0x3fefb5586cf9890full, 0x3fef9301d0125b51ull,
    0x3fef72b83c7d517bull, 0x3fef54873168b9aaull, 0x3fef387a6e756238ull, 0x3fef1e9df51fdee1ull,
    0x3fef06fe0a31b715ull, 0x3feef1a7373aa9cbull, 0x3feedea64c123422ull, 0x3feece086061892dull,
    0x3feebfdad5362a27ull, 0x3feeb42b569d4f82ull, 0x3feeab07dd485429ull, 0x3feea47eb03a5585ull,
    0x3feea09e667f3bcdull, 0x3fee9f75e8ec5f74ull, 0x3feea11473eb0187ull, 0x3feea589994cce13ull,
    0x3feeace5422aa0dbull, 0x3feeb737b0cdc5e5ull, 0x3feec49182a3f090ull, 0x3feed503b23e255dull,
    0x3feee89f995ad3adull, 0x3feeff76f2fb5e47ull, 0x3fef199bdd85529cull, 0x3fef3720dcef9069ull,
    0x3fef5818dcfba487ull, 0x3fef7c97337b9b5full, 0x3fefa4afa2a490daull, 0x3fefd0765b6e4540ull,
};

// exp(x) for x <= 0 (the Gaussian stamp only needs the non-positive half-line).
// tab: the 32-entry table above (constant memory) or an LDS copy of it.
// CHECK = false: the caller guarantees x > -104 (no underflow test on the hot path).
template <bool CHECK = true>
__device__ __forceinline__ float dev_expf_nonpos(float x, const uint64_t* tab)
{
    const double N = 32.0;
    const double InvLn2N = 0x1.71547652b82fep+0 * N;
    const double Shift = 0x1.8p+52;
    const double C0 = 0x1.c6af84b912394p-5 / N / N / N;
    const double C1 = 0x1.ebfce50fac4f3p-3 / N / N;
    const double C2 = 0x1.62e42ff0c52d6p-1 / N;
    if (CHECK && x < -0x1.9fe368p6f) return 0.0f;
    double xd = (double)x;
    double z = InvLn2N * xd;
    double kd = z + Shift;
    uint64_t ki = (uint64_t)__double_as_longlong(kd);
    kd = kd - Shift;
    double r = z - kd;
    uint64_t t = tab[ki & 31];
    t += ki << 47;
    double s = __longlong_as_double((long long)t);
    double zz = C0 * r + C1;
    double r2 = r * r;
    double y = C2 * r + 1.0;
    y = zz * r2 + y;
    y = y * s;
    return (float)y;
}

// sin and cos of an angle in [0, 2*pi] (radians, f32), glibc sincosf fast path.
__device__ __forceinline__ void dev_sincosf(float yf, float* sn, float* cs)
{
    const double hpi_inv = 0x1.45F306DC9C883p+23;
    const double hpi = 0x1.921FB54442D18p0;
    const double c0 = 0x1p0, c1 = -0x1.ffffffd0c621cp-2, c2 = 0x1.55553e1068f19p-5,
                 c3 = -0x1.6c087e89a359dp-10, c4 = 0x1.99343027bf8c3p-16;
    const double s1 = -0x1.555545995a603p-3, s2 = 0x1.1107605230bc4p-7, s3 = -0x1.994eb3774cf24p-13;
    double x = (double)yf;
    const uint32_t top = (__float_as_uint(yf) >> 20) & 0x7ff;
    int n = 0;
    double sgn = 1.0;
    bool small = top < 0x3f4;
    if (!small) {
        double r = x * hpi_inv;
        n = ((int32_t)r + 0x800000) >> 24;
        x = x - (double)n * hpi;
        sgn = (n & 1) != ((n >> 1) & 1) ? -1.0 : 1.0;      // sign table {1,-1,-1,1}
    }
    const double x2 = x * x;
    const double cg = (n & 2) ? -1.0 : 1.0;                   // second coefficient table: cos coeffs negated
    // sine-branch polynomial on (x*sgn) and cosine-branch polynomial (uses x2 only)
    const double xs = x * sgn;
    double x3 = xs * x2;
    double ps1 = s2 + x2 * s3;
    double x7 = x3 * x2;
    double ps = xs + x3 * s1;
    const double sin_poly = ps + x7 * ps1;
    double x4 = x2 * x2;
    double pc2 = cg * c3 + x2 * (cg * c4);
    double pc1 = cg * c0 + x2 * (cg * c1);
    double x6 = x4 * x2;
    double pc = pc1 + x4 * (cg * c2);
    const double cos_poly = pc + x6 * pc2;
    float s_out, c_out;
    if (small) {
        s_out = (top < 0x398) ? yf : (float)sin_poly;
        c_out = (top < 0x398) ? 1.0f : (float)cos_poly;
    } else {
        // sinf: poly index n; cosf: poly index n^1  (odd index = cosine branch)
        s_out = (float)((n & 1) ? cos_poly : sin_poly);
        c_out = (float)((n & 1) ? sin_poly : cos_poly);
    }
    *sn = s_out; *cs = c_out;
}

// cv::fastAtan2(y, x) in degrees, [0, 360]
__device__ __forceinline__ float dev_fast_atan2(float y, float x)
{
    const float scale = (float)(180.0 / 3.1415926535897932384626433832795);
    const float p1 = 0.9997878412794807f * scale;
    const float p3 = -0.3258083974640975f * scale;
    const float p5 = 0.1555786518463281f * scale;
    const float p7 = -0.04432655554792128f * scale;
    const float eps = (float)2.2204460492503131e-16;
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

// double-precision sin/cos for the per-event Eigen::AngleAxisd -> rotation matrix of ev2mci_gg_f
// (src/Event/EventConversion.cc:317-320): fdlibm k_sin / k_cos with the two-term pi/2 reduction, |x| < 100.
__device__ __forceinline__ double dev_ksin(double x, double y, int iy)
{
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04,
                 S4 = 2.75573137070700676789e-06, S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    const double z = x * x;
    const double v = z * x;
    const double r = S2 + z * (S3 + z * (S4 + z * (S5 + z * S6)));
    if (iy == 0) return x + v * (S1 + z * r);
    return x - ((z * (0.5 * y - v * r) - y) - v * S1);
}
__device__ __forceinline__ double dev_kcos(double x, double y)
{
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05,
                 C4 = -2.75573143513906633035e-07, C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    const double z = x * x;
    const double r = z * (C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6)))));
    const double ax = fabs(x);
    if (ax < 0.3) return 1.0 - (0.5 * z - (z * r - x * y));
    double qx = 0.28125;
    if (!(ax > 0.78125)) {
        const double q = 0.25 * ax;
        qx = __longlong_as_double(__double_as_longlong(q) & (long long)0xffffffff00000000ull);
    }
    const double hz = 0.5 * z - qx;
    const double a = 1.0 - qx;
    return a - (hz - (z * r - x * y));
}
__device__ __forceinline__ void dev_dsincos(double x, double* sn, double* cs)
{
    if (fabs(x) <= 0.78539816339744830962) { *sn = dev_ksin(x, 0.0, 0); *cs = dev_kcos(x, 0.0); return; }
    const double invpio2 = 6.36619772367581382433e-01, pio2_1 = 1.57079632673412561417e+00, pio2_1t = 6.07710050650619224932e-11;
    const double t = fabs(x);
    int n = (int)(t * invpio2 + 0.5);
    const double fn = (double)n;
    const double r = t - fn * pio2_1;
    const double w = fn * pio2_1t;
    double a = r - w;
    double b = (r - a) - w;
    if (x < 0) { a = -a; b = -b; n = -n; }
    const double ks = dev_ksin(a, b, 1), kc = dev_kcos(a, b);
    switch (n & 3) {
        case 0: *sn = ks; *cs = kc; break;
        case 1: *sn = kc; *cs = -ks; break;
        case 2: *sn = -ks; *cs = -kc; break;
        default: *sn = -kc; *cs = ks; break;
    }
}

// ---- tanf / atanf / atan2f: glibc's float algorithms (s_tanf.c + k_tanf.c with the sincosf.h double reduction; s_atanf.c;
// e_atan2f.c), the operation sequences of oracle/orc_math.c written independently here; needed by KannalaBrandt8::unproject /
// project (src/CameraModels/KannalaBrandt8.cpp:87-190).  Checked against the oracle over their whole domains by
// eorb_selfcheck_math (which = 3, 4, 5), the oracle against the host glibc exhaustively (oracle/check_libm.c).
__device__ __forceinline__ float dev_ktanf(float x, float y, int iy)
{
    const float T0 = 3.3333334327e-01f, T1 = 1.3333334029e-01f, T2 = 5.3968254477e-02f, T3 = 2.1869488060e-02f, T4 = 8.8632395491e-03f,
                T5 = 3.5920790397e-03f, T6 = 1.4562094584e-03f, T7 = 5.8804126456e-04f, T8 = 2.4646313977e-04f, T9 = 7.8179444245e-05f,
                T10 = 7.1407252108e-05f, T11 = -1.8558637748e-05f, T12 = 2.5907305826e-05f;
    const float pio4 = 7.8539812565e-01f, pio4lo = 3.7748947079e-08f;
    const int hx = (int)__float_as_uint(x), ix = hx & 0x7fffffff;
    if (ix < 0x39000000) {
        if ((int)x == 0) {
            if ((ix | (iy + 1)) == 0) return 1.0f / fabsf(x);
            return (iy == 1) ? x : -1.0f / x;
        }
    }
    if (ix >= 0x3f2ca140) {
        if (hx < 0) { x = -x; y = -y; }
        const float z0 = pio4 - x;
        const float w0 = pio4lo - y;
        x = z0 + w0; y = 0.0f;
        if (fabsf(x) < 0x1p-13f) return (float)((1 - ((hx >> 30) & 2)) * iy) * (1.0f - (float)(2 * iy) * x);
    }
    float z = x * x;
    float w = z * z;
    float r = T1 + w * (T3 + w * (T5 + w * (T7 + w * (T9 + w * T11))));
    float v = z * (T2 + w * (T4 + w * (T6 + w * (T8 + w * (T10 + w * T12)))));
    float s = z * x;
    r = y + z * (s * (r + v) + y);
    r += T0 * s;
    w = x + r;
    if (ix >= 0x3f2ca140) {
        v = (float)iy;
        return (float)(1 - ((hx >> 30) & 2)) * (v - 2.0f * (x - (w * w / (w + v) - r)));
    }
    if (iy == 1) return w;
    z = __uint_as_float(__float_as_uint(w) & 0xfffff000u);
    v = r - (z - x);
    const float a = -1.0f / w;
    const float t = __uint_as_float(__float_as_uint(a) & 0xfffff000u);
    s = 1.0f + t * z;
    return t + a * (s + t * v);
}
__device__ __forceinline__ float dev_tanf(float x)             // |x| < 120
{
    const int ix = (int)(__float_as_uint(x) & 0x7fffffffu);
    if (ix <= 0x3f490fda) return dev_ktanf(x, 0.0f, 1);
    const double hpi_inv = 0x1.45F306DC9C883p+23, hpi = 0x1.921FB54442D18p0;
    const double xd = (double)x;
    const double rr = xd * hpi_inv;
    const int n = ((int)rr + 0x800000) >> 24;
    const double xr = xd - (double)n * hpi;
    const float y0 = (float)xr;
    const float y1 = (float)(xr - (double)y0);
    return dev_ktanf(y0, y1, 1 - ((n & 1) << 1));
}
__device__ __forceinline__ float dev_atanf(float x)
{
    const float aT0 = 3.3333334327e-01f, aT1 = -2.0000000298e-01f, aT2 = 1.4285714924e-01f, aT3 = -1.1111110449e-01f,
                aT4 = 9.0908870101e-02f, aT5 = -7.6918758452e-02f, aT6 = 6.6610731184e-02f, aT7 = -5.8335702866e-02f,
                aT8 = 4.9768779427e-02f, aT9 = -3.6531571299e-02f, aT10 = 1.6285819933e-02f;
    const int hx = (int)__float_as_uint(x), ix = hx & 0x7fffffff;
    int id;
    float hi = 0.f, lo = 0.f;
    if (ix >= 0x4c000000) {
        if (ix > 0x7f800000) return x + x;
        return (hx > 0) ? 1.5707962513e+00f + 7.5497894159e-08f : -1.5707962513e+00f - 7.5497894159e-08f;
    }
    if (ix < 0x3ee00000) {
        if (ix < 0x31000000) return x;
        id = -1;
    } else {
        x = fabsf(x);
        if (ix < 0x3f980000) {
            if (ix < 0x3f300000) { id = 0; hi = 4.6364760399e-01f; lo = 5.0121582440e-09f; x = (2.0f * x - 1.0f) / (2.0f + x); }
            else { id = 1; hi = 7.8539812565e-01f; lo = 3.7748947079e-08f; x = (x - 1.0f) / (x + 1.0f); }
        } else {
            if (ix < 0x401c0000) { id = 2; hi = 9.8279368877e-01f; lo = 3.4473217170e-08f; x = (x - 1.5f) / (1.0f + 1.5f * x); }
            else { id = 3; hi = 1.5707962513e+00f; lo = 7.5497894159e-08f; x = -1.0f / x; }
        }
    }
    const float z = x * x;
    const float w = z * z;
    const float s1 = z * (aT0 + w * (aT2 + w * (aT4 + w * (aT6 + w * (aT8 + w * aT10)))));
    const float s2 = w * (aT1 + w * (aT3 + w * (aT5 + w * (aT7 + w * aT9))));
    if (id < 0) return x - x * (s1 + s2);
    const float r = hi - ((x * (s1 + s2) - lo) - x);
    return (hx < 0) ? -r : r;
}
__device__ __forceinline__ float dev_atan2f(float y, float x)
{
    const float pi_o_4 = 7.8539818525e-01f, pi_o_2 = 1.5707963705e+00f, pi = 3.1415927410e+00f, pi_lo = -8.7422776573e-08f, tiny = 1.0e-30f;
    const int hx = (int)__float_as_uint(x), ix = hx & 0x7fffffff, hy = (int)__float_as_uint(y), iy = hy & 0x7fffffff;
    if (ix > 0x7f800000 || iy > 0x7f800000) return x + y;
    if (hx == 0x3f800000) return dev_atanf(y);
    const int m = ((hy >> 31) & 1) | ((hx >> 30) & 2);
    if (iy == 0) { if (m < 2) return y; return (m == 2) ? pi + tiny : -pi - tiny; }
    if (ix == 0) return (hy < 0) ? -pi_o_2 - tiny : pi_o_2 + tiny;
    if (ix == 0x7f800000) {
        if (iy == 0x7f800000) { return (m == 0) ? pi_o_4 + tiny : (m == 1) ? -pi_o_4 - tiny : (m == 2) ? 3.0f * pi_o_4 + tiny : -3.0f * pi_o_4 - tiny; }
        return (m == 0) ? 0.0f : (m == 1) ? -0.0f : (m == 2) ? pi + tiny : -pi - tiny;
    }
    if (iy == 0x7f800000) return (hy < 0) ? -pi_o_2 - tiny : pi_o_2 + tiny;
    const int k = (iy - ix) >> 23;
    float z;
    if (k > 60) z = pi_o_2 + 0.5f * pi_lo;
    else if (hx < 0 && k < -60) z = 0.0f;
    else z = dev_atanf(fabsf(y / x));
    if (m == 0) return z;
    if (m == 1) return __uint_as_float(__float_as_uint(z) ^ 0x80000000u);
    if (m == 2) return pi - (z - pi_lo);
    return (z - pi_lo) - pi;
}

__device__ __forceinline__ int dev_cvround(float v) { return __float2int_rn(v); }

// popcount of a 256-bit XOR: ORBmatcher::DescriptorDistance (src/ORBmatcher.cc:2360-2378)
__device__ __forceinline__ int dev_hamming256(const uint64_t a[4], const uint64_t b[4])
{
    return __popcll(a[0] ^ b[0]) + __popcll(a[1] ^ b[1]) + __popcll(a[2] ^ b[2]) + __popcll(a[3] ^ b[3]);
}

__device__ __forceinline__ int reflect101(int p, int len)
{
    if (len == 1) return 0;
    while (p < 0 || p >= len) p = (p < 0) ? -p : 2 * len - 2 - p;
    return p;
}

}  // namespace eorb
