/*
 * orc_events.c -- ORACLE (test infrastructure only): event -> image accumulation.
 * Restates src/Event/EventConversion.cc of the reference, sequentially, one IEEE op per
 * written op (-ffp-contract=off).  powf(v,2) is v*v (what GCC emits for a literal exponent 2).
 */
#include "eorb_oracle.h"
#include <math.h>
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
