/*
 * orc_events.c -- ORACLE (test infrastructure only): event -> image accumulation.
 * Restates src/Event/EventConversion.cc of the reference, sequentially, one IEEE op per
 * written op (-ffp-contract=off).  powf(v,2) is v*v (what GCC emits for a literal exponent 2).
 */
#include "eorb_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* MyCalibrator::isInImage, src/Utils/MyCalibrator.cpp:36-39 */
static inline int in_image(int x, int y, int W, int H) { return x >= 0 && x < W && y >= 0 && y < H; }

/* Mat::convertTo(CV_8UC1, alpha, beta): saturate_cast<uchar>(cvRound(src*alpha + beta)) in f32
 * (SURVEY App.B H2).  normalizeImage: EventConversion.cc:67-72 */
void orc_normalize_u8(const float* src, size_t npix, float maxVal, float minVal, uint8_t* dst)
{
    float alpha = 255.f / (maxVal - minVal);
    float beta = -minVal * alpha;
    for (size_t i = 0; i < npix; i++) {
        float m = src[i] * alpha;
        float v = m + beta;
        int iv = orc_cvround((double)v);
        dst[i] = (uint8_t)(iv < 0 ? 0 : iv > 255 ? 255 : iv);
    }
}

/* EventConversion.cc:173-212 */
int orc_ev2im(const orc_event* ev, size_t n, int W, int H, int pol, int normalized,
              float* out_f32, uint8_t* out_u8, float* minmax)
{
    float maxVal = -1000000.0f, minVal = 0.0f;
    memset(out_f32, 0, sizeof(float) * (size_t)W * H);
    for (size_t i = 0; i < n; i++) {
        float polSign = (pol && !ev[i].p) ? -1.0f : 1.0f;          /* resolvePolarity :26-30 */
        int pX = (int)roundf(ev[i].x);                              /* roundFloatCoord :46-49 */
        int pY = (int)roundf(ev[i].y);
        if (!in_image(pX, pY, W, H)) continue;
        float newVal = out_f32[(size_t)pY * W + pX] + (polSign * 0.001f);
        out_f32[(size_t)pY * W + pX] = newVal;
        if (newVal > maxVal) maxVal = newVal;                       /* resolveMinMaxVals :32-39 */
        if (newVal < minVal) minVal = newVal;
    }
    if (minmax) { minmax[0] = minVal; minmax[1] = maxVal; }
    if (normalized && maxVal > minVal) {
        if (out_u8) orc_normalize_u8(out_f32, (size_t)W * H, maxVal, minVal, out_u8);
        return 1;
    }
    return 0;
}

/* exp_XY2f :59-65 */
static inline float exp_xy2f(float x, float y, float sig2)
{
    float xx = x * x;
    float yy = y * y;
    float dd = xx + yy;
    dd = dd / (2.0f * sig2);
    const float two_pi = 2.0f * (float)3.1415926535897932384626433832795;
    float val = orc_expf(-dd) / (two_pi * sig2);
    return val;
}

/* EventConversion.cc:215-269 */
int orc_ev2im_gauss(const orc_event* ev, size_t n, int W, int H, float sigma, int pol,
                    int normalized, float* out_f32, uint8_t* out_u8, float* minmax)
{
    float maxVal = -1000000.0f, minVal = 0.0f;
    float sig2 = sigma * sigma;
    int h = (int)ceil((double)sigma * 3.0);
    memset(out_f32, 0, sizeof(float) * (size_t)W * H);
    for (size_t k = 0; k < n; k++) {
        /* breakFloatCoords :51-57 */
        int xi = (int)floor((double)ev[k].x);
        float xr = ev[k].x - (float)xi;
        int yi = (int)floor((double)ev[k].y);
        float yr = ev[k].y - (float)yi;
        float polSign = (pol && !ev[k].p) ? -1.0f : 1.0f;
        for (int i = -h; i <= h; i++) {
            for (int j = -h; j <= h; j++) {
                int xn = xi + i, yn = yi + j;
                if (!in_image(xn, yn, W, H)) continue;
                float val = exp_xy2f((float)i - xr, (float)j - yr, sig2);
                float newVal = out_f32[(size_t)yn * W + xn] + polSign * val;
                out_f32[(size_t)yn * W + xn] = newVal;
                if (newVal > maxVal) maxVal = newVal;
                if (newVal < minVal) minVal = newVal;
            }
        }
    }
    if (minmax) { minmax[0] = minVal; minmax[1] = maxVal; }
    if (normalized) {
        if (out_u8) orc_normalize_u8(out_f32, (size_t)W * H, maxVal, minVal, out_u8);
        return 1;
    }
    return 0;
}

/* ---- motion-compensated accumulation: EventConversion.cc:280-531 ------------------------------------------------- */
/* KannalaBrandt8::unproject (src/CameraModels/KannalaBrandt8.cpp:163-190): Newton iterations on theta in float, std::tan(float) */
static void kb8_unproject(const orc_camera* c, float x, float y, float* X, float* Y)
{
    const float pwx = (x - c->cx) / c->fx, pwy = (y - c->cy) / c->fy;
    float scale = 1.f;
    float theta_d = sqrtf(pwx * pwx + pwy * pwy);
    theta_d = fminf(fmaxf((float)(-3.1415926535897932384626433832795 / 2.f), theta_d), (float)(3.1415926535897932384626433832795 / 2.f));
    if ((double)theta_d > 1e-8) {
        float theta = theta_d;
        for (int j = 0; j < 10; j++) {
            const float theta2 = theta * theta, theta4 = theta2 * theta2, theta6 = theta4 * theta2, theta8 = theta4 * theta4;
            const float k0_theta2 = c->k[0] * theta2, k1_theta4 = c->k[1] * theta4;
            const float k2_theta6 = c->k[2] * theta6, k3_theta8 = c->k[3] * theta8;
            const float theta_fix = (theta * (1 + k0_theta2 + k1_theta4 + k2_theta6 + k3_theta8) - theta_d) /
                                    (1 + 3 * k0_theta2 + 5 * k1_theta4 + 7 * k2_theta6 + 9 * k3_theta8);
            theta = theta - theta_fix;
            if (fabsf(theta_fix) < c->precision) break;
        }
        scale = orc_tanf(theta) / theta_d;
    }
    *X = pwx * scale; *Y = pwy * scale;
}
/* KannalaBrandt8::project(cv::Point3f) (:87-103); cos / sin of the float angle resolve to the float overloads (<math.h> of
 * libstdc++ is reached through opencv2/opencv.hpp -> flann/lsh_table.h: unpinned, as for the other libm calls) */
static void kb8_project_f(const orc_camera* c, float X, float Y, float Z, float* u, float* v)
{
    const float x2_plus_y2 = X * X + Y * Y;
    const float theta = orc_atan2f(sqrtf(x2_plus_y2), Z);
    const float psi = orc_atan2f(Y, X);
    const float theta2 = theta * theta, theta3 = theta * theta2, theta5 = theta3 * theta2, theta7 = theta5 * theta2, theta9 = theta7 * theta2;
    const float r = theta + c->k[0] * theta3 + c->k[1] * theta5 + c->k[2] * theta7 + c->k[3] * theta9;
    *u = c->fx * r * orc_cosf_any(psi) + c->cx;
    *v = c->fy * r * orc_sinf_any(psi) + c->cy;
}
/* KannalaBrandt8::project(Eigen::Vector3d) (:111-133): atan2f / sqrtf on the float-converted arguments, the polynomial and
 * cos / sin in double */
static void kb8_project_d(const orc_camera* c, const double P[3], double* u, double* v)
{
    const double x2_plus_y2 = P[0] * P[0] + P[1] * P[1];
    const double theta = (double)orc_atan2f(sqrtf((float)x2_plus_y2), (float)P[2]);
    const double psi = (double)orc_atan2f((float)P[1], (float)P[0]);
    const double theta2 = theta * theta, theta3 = theta * theta2, theta5 = theta3 * theta2, theta7 = theta5 * theta2, theta9 = theta7 * theta2;
    const double r = theta + (double)c->k[0] * theta3 + (double)c->k[1] * theta5 + (double)c->k[2] * theta7 + (double)c->k[3] * theta9;
    *u = (double)c->fx * r * orc_dcos(psi) + (double)c->cx;
    *v = (double)c->fy * r * orc_dsin(psi) + (double)c->cy;
}

void orc_mci_warp_se3(const orc_event* ev, size_t n, const orc_pinhole* cam, double angle, const double axis[3],
                      const double tt[3], float medDepth, const float* depth_per_event, float* uv)
{
    const orc_camera c = {0, cam->fx, cam->fy, cam->cx, cam->cy, {0.f, 0.f, 0.f, 0.f}, 0.f};
    orc_mci_warp_se3_cam(ev, n, &c, angle, axis, tt, medDepth, depth_per_event, uv);
}

void orc_mci_warp_se3_cam(const orc_event* ev, size_t n, const orc_camera* cam, double angle, const double axis[3],
                          const double tt[3], float medDepth, const float* depth_per_event, float* uv)
{
    if (n == 0) return;
    const double t1 = ev[n - 1].ts;
    const double DT = t1 - ev[0].ts;
    const double invDT = 1.0 / DT;
    for (size_t k = 0; k < n; k++) {
        const double etRate = (t1 - ev[k].ts) * invDT;
        float X, Y;
        if (cam->model == 1) kb8_unproject(cam, ev[k].x, ev[k].y, &X, &Y);
        else { X = (ev[k].x - cam->cx) / cam->fx; Y = (ev[k].y - cam->cy) / cam->fy; }       /* Pinhole::unproject (Pinhole.cpp:59-62) */
        const double P[3] = { (double)X, (double)Y, (double)1.f };
        /* Eigen::AngleAxisd(omega.angle()*etRate, omega.axis()).toRotationMatrix() */
        const double a = angle * etRate;
        const double sn = orc_dsin(a), c = orc_dcos(a);
        const double sax = sn * axis[0], say = sn * axis[1], saz = sn * axis[2];
        const double c1x = (1.0 - c) * axis[0], c1y = (1.0 - c) * axis[1], c1z = (1.0 - c) * axis[2];
        double R[3][3];
        double tmp;
        tmp = c1x * axis[1]; R[0][1] = tmp - saz; R[1][0] = tmp + saz;
        tmp = c1x * axis[2]; R[0][2] = tmp + say; R[2][0] = tmp - say;
        tmp = c1y * axis[2]; R[1][2] = tmp - sax; R[2][1] = tmp + sax;
        R[0][0] = c1x * axis[0] + c; R[1][1] = c1y * axis[1] + c; R[2][2] = c1z * axis[2] + c;
        const double d = (double)(depth_per_event ? depth_per_event[k] : medDepth);
        double np[3];
        for (int i = 0; i < 3; i++) {
            /* (medDepth * newR) * P3D + newT (:324): Eigen 3.3 (libeigen3-dev, build_eorb_slam.sh:54) evaluates the scaled
             * matrix into a temporary, then the fixed-size coefficient-based product row(i).cwiseProduct(P3D).sum(), whose
             * completely unrolled 3-term redux (redux_novec_unroller, halves 1 + 2) associates as a0 + (a1 + a2) */
            const double a0 = (d * R[i][0]) * P[0];
            const double a1 = (d * R[i][1]) * P[1];
            const double a2 = (d * R[i][2]) * P[2];
            const double acc = a0 + (a1 + a2);
            np[i] = acc + tt[i] * etRate;
        }
        double u, v;
        if (cam->model == 1) kb8_project_d(cam, np, &u, &v);
        else {                                         /* Pinhole::project(Eigen::Vector3d) :41-47 */
            u = (double)cam->fx * np[0] / np[2] + (double)cam->cx;
            v = (double)cam->fy * np[1] / np[2] + (double)cam->cy;
        }
        uv[2 * k] = (float)u; uv[2 * k + 1] = (float)v;
    }
}

void orc_mci_warp_se2(const orc_event* ev, size_t n, const orc_pinhole* cam, const float* params2D, int nparams, float* uv)
{
    const orc_camera c = {0, cam->fx, cam->fy, cam->cx, cam->cy, {0.f, 0.f, 0.f, 0.f}, 0.f};
    orc_mci_warp_se2_cam(ev, n, &c, params2D, nparams, uv);
}

void orc_mci_warp_se2_cam(const orc_event* ev, size_t n, const orc_camera* cam, const float* params2D, int nparams, float* uv)
{
    if (n == 0) return;
    const double t1 = ev[n - 1].ts;
    const float DT = (float)(t1 - ev[0].ts);
    const float invDT = 1.f / DT;
    const float omega0 = params2D[0] * invDT, vx0 = params2D[1] * invDT, vy0 = params2D[2] * invDT;
    float sc = 1.f;
    if (nparams > 3) sc = params2D[3];
    const float scDiff = 1.f - sc;
    for (size_t k = 0; k < n; k++) {
        const float tk = (float)(t1 - ev[k].ts);
        float X, Y;
        if (cam->model == 1) kb8_unproject(cam, ev[k].x, ev[k].y, &X, &Y);
        else { X = (ev[k].x - cam->cx) / cam->fx; Y = (ev[k].y - cam->cy) / cam->fy; }
        const float Z = 1.f;
        const float theta_k = tk * omega0;
        const float currSc = scDiff * (1 - tk * invDT) + sc;
        const float cs = orc_cosf(theta_k), sn = orc_sinf(theta_k);
        const float xp = currSc * (X * cs - Y * sn) + vx0 * tk;
        const float yp = currSc * (X * sn + Y * cs) + vy0 * tk;
        if (cam->model == 1) kb8_project_f(cam, xp, yp, Z, &uv[2 * k], &uv[2 * k + 1]);
        else {                                         /* Pinhole::project(cv::Point3f) :30-33 */
            uv[2 * k] = cam->fx * xp / Z + cam->cx;
            uv[2 * k + 1] = cam->fy * yp / Z + cam->cy;
        }
    }
}

static int mci_splat(const orc_event* ev, size_t n, const float* uv, int W, int H, float sigma, int pol, int normalized,
                     float* out_f32, uint8_t* out_u8, float* minmax)
{
    orc_event* w = (orc_event*)malloc(sizeof(orc_event) * (n ? n : 1));
    for (size_t k = 0; k < n; k++) { w[k] = ev[k]; w[k].x = uv[2 * k]; w[k].y = uv[2 * k + 1]; }
    int r = orc_ev2im_gauss(w, n, W, H, sigma, pol, normalized, out_f32, out_u8, minmax);
    free(w);
    return r;
}

int orc_ev2mci_se3(const orc_event* ev, size_t n, const orc_pinhole* cam, double angle, const double axis[3], const double tt[3],
                   float medDepth, const float* depth_per_event, int W, int H, float sigma, int pol, int normalized,
                   float* out_f32, uint8_t* out_u8, float* minmax)
{
    const orc_camera c = {0, cam->fx, cam->fy, cam->cx, cam->cy, {0.f, 0.f, 0.f, 0.f}, 0.f};
    return orc_ev2mci_se3_cam(ev, n, &c, angle, axis, tt, medDepth, depth_per_event, W, H, sigma, pol, normalized, out_f32, out_u8, minmax);
}

int orc_ev2mci_se3_cam(const orc_event* ev, size_t n, const orc_camera* cam, double angle, const double axis[3], const double tt[3],
                       float medDepth, const float* depth_per_event, int W, int H, float sigma, int pol, int normalized,
                       float* out_f32, uint8_t* out_u8, float* minmax)
{
    if (n == 0) {                       /* "no events": returns the zero CV_32FC1 image (:292-295) */
        memset(out_f32, 0, sizeof(float) * (size_t)W * H);
        if (minmax) { minmax[0] = 0.f; minmax[1] = -1000000.0f; }
        return 0;
    }
    float* uv = (float*)malloc(sizeof(float) * 2 * n);
    orc_mci_warp_se3_cam(ev, n, cam, angle, axis, tt, medDepth, depth_per_event, uv);
    int r = mci_splat(ev, n, uv, W, H, sigma, pol, normalized, out_f32, out_u8, minmax);
    free(uv);
    return r;
}

int orc_ev2mci_se2(const orc_event* ev, size_t n, const orc_pinhole* cam, const float* params2D, int nparams,
                   int W, int H, float sigma, int pol, int normalized, float* out_f32, uint8_t* out_u8, float* minmax)
{
    const orc_camera c = {0, cam->fx, cam->fy, cam->cx, cam->cy, {0.f, 0.f, 0.f, 0.f}, 0.f};
    return orc_ev2mci_se2_cam(ev, n, &c, params2D, nparams, W, H, sigma, pol, normalized, out_f32, out_u8, minmax);
}

int orc_ev2mci_se2_cam(const orc_event* ev, size_t n, const orc_camera* cam, const float* params2D, int nparams,
                       int W, int H, float sigma, int pol, int normalized, float* out_f32, uint8_t* out_u8, float* minmax)
{
    if (n == 0) {
        memset(out_f32, 0, sizeof(float) * (size_t)W * H);
        if (minmax) { minmax[0] = 0.f; minmax[1] = -1000000.0f; }
        return 0;
    }
    float* uv = (float*)malloc(sizeof(float) * 2 * n);
    orc_mci_warp_se2_cam(ev, n, cam, params2D, nparams, uv);
    int r = mci_splat(ev, n, uv, W, H, sigma, pol, normalized, out_f32, out_u8, minmax);
    free(uv);
    return r;
}

/* EvImConverter::measureImageFocus :74-111 ; cv::meanStdDev on a CV_32FC1 patch: sum and sum of squares accumulated in
 * double in raster order, stddev = sqrt(max(sqsum/N - mean^2, 0)) */
float orc_measure_image_focus(const float* img, int W, int H)
{
    const int patch = 30;
    float localStd = 0.f;
    int cnt = 0;
    for (int i = 0; i < H; i += patch) {
        for (int j = 0; j < W; j += patch) {
            const int maxRow = i + patch < H ? i + patch : H, maxCol = j + patch < W ? j + patch : W;
            double s = 0, sq = 0;
            for (int y = i; y < maxRow; y++)
                for (int x = j; x < maxCol; x++) { const double v = (double)img[(size_t)y * W + x]; s += v; sq += v * v; }
            const double N = (double)(maxRow - i) * (double)(maxCol - j);
            const double scale = 1.0 / N;
            const double mean = s * scale;
            double var = sq * scale - mean * mean;
            if (var < 0) var = 0;
            const float currStd = (float)sqrt(var);
            localStd += currStd;
            cnt++;
        }
    }
    return localStd / (float)cnt;
}

/* cv::normalize(src, dst, 255, 0, NORM_MINMAX, CV_8UC1): scale = 255 * (1/(max-min)) in double (0 when max-min <=
 * DBL_EPSILON), shift = -min*scale; convertTo works with float(scale), float(shift) */
void orc_cv_normalize_minmax_u8(const float* img, size_t npix, uint8_t* dst)
{
    double smin = img[0], smax = img[0];
    for (size_t i = 1; i < npix; i++) { if (img[i] < smin) smin = img[i]; if (img[i] > smax) smax = img[i]; }
    const double scale = (255.0 - 0.0) * (smax - smin > 2.2204460492503131e-16 ? 1. / (smax - smin) : 0);
    const double shift = 0.0 - smin * scale;
    const float fs = (float)scale, fh = (float)shift;
    for (size_t i = 0; i < npix; i++) {
        const float m = img[i] * fs;
        const float v = m + fh;
        int iv = orc_cvround((double)v);
        dst[i] = (uint8_t)(iv < 0 ? 0 : iv > 255 ? 255 : iv);
    }
}

/* EventDataStore::getEventChunkRectified :264-305 (parse excluded) */
size_t orc_undistort_events(const orc_raw_event* raw, size_t n, const float* mapX, const float* mapY, int LW, int LH,
                            int W, int H, int checkInImage, double tsFactor, orc_event* out)
{
    size_t k = 0;
    for (size_t i = 0; i < n; i++) {
        const int x = (int)raw[i].x, y = (int)raw[i].y;                 /* static_cast<int>(srcPt.x) :172-173 */
        if (x < 0 || x >= LW || y < 0 || y >= LH) continue;             /* assert :175 */
        const float ux = mapX[(size_t)y * LW + x], uy = mapY[(size_t)y * LW + x];
        if (checkInImage && !((ux >= 0 && ux < (float)W) && (uy >= 0 && uy < (float)H))) continue;   /* :295-296 */
        memset(&out[k], 0, sizeof(orc_event));
        out[k].ts = raw[i].t / tsFactor; out[k].x = ux; out[k].y = uy; out[k].p = raw[i].p ? 1 : 0;
        k++;
    }
    return k;
}

/* EventLoader.cpp:80-92 "stream >> ts >> x >> y >> p" over the lines of a text buffer */
long orc_parse_events_text(const char* text, size_t nbytes, orc_raw_event* out, size_t cap)
{
    size_t pos = 0; long line = 0; size_t n = 0;
    char buf[256];
    while (pos < nbytes) {
        size_t end = pos;
        while (end < nbytes && text[end] != '\n') end++;
        size_t len = end - pos;
        if (len > 0 && text[pos + len - 1] == '\r') len--;
        size_t i = 0;
        while (i < len && (text[pos + i] == ' ' || text[pos + i] == '\t')) i++;
        if (i < len && text[pos + i] != '#') {                           /* isComment / blank */
            if (len >= sizeof(buf)) return -line - 1;
            memcpy(buf, text + pos, len); buf[len] = 0;
            for (size_t k = 0; k < len; k++) {                            /* grammar: digits, '.', blanks only */
                const char ch = buf[k];
                if (!((ch >= '0' && ch <= '9') || ch == '.' || ch == ' ' || ch == '\t')) return -line - 1;
            }
            char* q = buf; char* e;
            {   /* ts restricted to the exactly convertible case: <= 19 significant digits, mantissa < 2^53, <= 22 fraction digits */
                const char* d = buf + i; unsigned long long mant = 0; int sig = 0, frac = 0, point = 0, big = 0;
                for (; (*d >= '0' && *d <= '9') || (*d == '.' && !point); d++) {
                    if (*d == '.') { point = 1; continue; }
                    if (mant != 0 || *d != '0') { if (sig < 19) { mant = mant * 10 + (unsigned)(*d - '0'); sig++; } else big = 1; }
                    if (point) frac++;
                }
                if (big || mant >= (1ull << 53) || frac > 22) return -line - 1;
            }
            const double ts = strtod(q, &e); if (e == q) return -line - 1; q = e;
            const float x = strtof(q, &e); if (e == q) return -line - 1; q = e;
            const float y = strtof(q, &e); if (e == q) return -line - 1; q = e;
            const long pp = strtol(q, &e, 10); if (e == q) return -line - 1; q = e;
            while (*q == ' ' || *q == '\t') q++;
            if (*q != 0 || (pp != 0 && pp != 1)) return -line - 1;
            if (x != (float)(int)x || y != (float)(int)y || x < 0 || x > 65535 || y < 0 || y > 65535) return -line - 1;
            if (n >= cap) return -line - 1;
            memset(&out[n], 0, sizeof(orc_raw_event));
            out[n].x = (uint16_t)(int)x; out[n].y = (uint16_t)(int)y; out[n].p = (uint32_t)pp; out[n].t = ts;
            n++;
        }
        pos = end + 1; line++;
    }
    return (long)n;
}
