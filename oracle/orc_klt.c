/* Oracle restatement of cv::calcOpticalFlowPyrLK as ELK_Tracker::trackCurrImage calls it (src/Event/KLT_Tracker.cpp:49-98,
 * parameters Examples/Event/EvETHZ.yaml:205-208).  TEST INFRASTRUCTURE ONLY (see oracle/README.md).
 *
 * The algorithm lives in OpenCV 3.4.1 (modules/video/src/lkpyramid.cpp, modules/imgproc/src/pyramids.cpp), a dependency that
 * is absent from the reference tree: this file restates its published scalar code path -- buildOpticalFlowPyramid (pyrDown
 * 5x5 [1 4 6 4 1], BORDER_REFLECT_101 padding by winSize), calcSharrDeriv (3-10-3 Scharr, int16 pairs, zero padding),
 * LKTrackerInvoker (14-bit fixed-point bilinear patches, float normal equations accumulated in raster order).  PARITY UNPINNED:
 * nothing in the reference pins it, and an SSE2 build of OpenCV accumulates the same sums in four-lane partial sums.
 */
#include "eorb_oracle.h"
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

static int reflect101(int p, int len)
{   /* cv::borderInterpolate(p, len, BORDER_REFLECT_101) */
    if ((unsigned)p < (unsigned)len) return p;
    if (len == 1) return 0;
    do { if (p < 0) p = -p; else p = 2 * len - p - 2; } while ((unsigned)p >= (unsigned)len);
    return p;
}

typedef struct { int w, h, stride; uint8_t* img; int16_t* der; } lk_level;   /* both padded by `win` on every side */

static void level_alloc(lk_level* L, int w, int h, int win)
{
    L->w = w; L->h = h; L->stride = w + 2 * win;
    L->img = (uint8_t*)calloc((size_t)L->stride * (h + 2 * win), 1);
    L->der = (int16_t*)calloc((size_t)L->stride * (h + 2 * win) * 2, sizeof(int16_t));
}
static void level_free(lk_level* L) { free(L->img); free(L->der); }
#define PIX(L, x, y) ((L)->img[(size_t)((y) + win) * (L)->stride + (x) + win])

static void pad_reflect(lk_level* L, int win)
{   /* copyMakeBorder(level, temp, win, win, win, win, BORDER_REFLECT_101 | BORDER_ISOLATED) */
    for (int y = -win; y < L->h + win; y++)
        for (int x = -win; x < L->w + win; x++)
            if (x < 0 || x >= L->w || y < 0 || y >= L->h) PIX(L, x, y) = PIX(L, reflect101(x, L->w), reflect101(y, L->h));
}

static void pyr_down(const lk_level* S, lk_level* D, int win)
{   /* cv::pyrDown 8u: horizontal 1 4 6 4 1 into ints, vertical 1 4 6 4 1, (sum + 128) >> 8 */
    int* rows = (int*)malloc(sizeof(int) * 5 * D->w);
    for (int y = 0; y < D->h; y++) {
        for (int k = 0; k < 5; k++) {
            const int sy = reflect101(2 * y - 2 + k, S->h);
            for (int x = 0; x < D->w; x++) {
                const int x0 = reflect101(2 * x - 2, S->w), x1 = reflect101(2 * x - 1, S->w), x2 = reflect101(2 * x, S->w),
                          x3 = reflect101(2 * x + 1, S->w), x4 = reflect101(2 * x + 2, S->w);
                rows[k * D->w + x] = PIX(S, x2, sy) * 6 + (PIX(S, x1, sy) + PIX(S, x3, sy)) * 4 + PIX(S, x0, sy) + PIX(S, x4, sy);
            }
        }
        for (int x = 0; x < D->w; x++) {
            const int v = rows[x] + rows[4 * D->w + x] + (rows[D->w + x] + rows[3 * D->w + x]) * 4 + rows[2 * D->w + x] * 6;
            PIX(D, x, y) = (uint8_t)((v + 128) >> 8);
        }
    }
    free(rows);
}

static void scharr_deriv(lk_level* L, int win)
{   /* calcSharrDeriv: vertical [3 10 3] / [-1 0 1] then horizontal, borders replicate the inner neighbour (reflect-101) */
    const int w = L->w, h = L->h;
    int16_t* t0 = (int16_t*)malloc(sizeof(int16_t) * (w + 2)); int16_t* t1 = (int16_t*)malloc(sizeof(int16_t) * (w + 2));
    for (int y = 0; y < h; y++) {
        const int y0 = y > 0 ? y - 1 : (h > 1 ? 1 : 0), y2 = y < h - 1 ? y + 1 : (h > 1 ? h - 2 : 0);
        for (int x = 0; x < w; x++) {
            t0[x + 1] = (int16_t)((PIX(L, x, y0) + PIX(L, x, y2)) * 3 + PIX(L, x, y) * 10);
            t1[x + 1] = (int16_t)(PIX(L, x, y2) - PIX(L, x, y0));
        }
        const int xl = w > 1 ? 1 : 0, xr = w > 1 ? w - 2 : 0;
        t0[0] = t0[xl + 1]; t0[w + 1] = t0[xr + 1]; t1[0] = t1[xl + 1]; t1[w + 1] = t1[xr + 1];
        int16_t* d = L->der + ((size_t)(y + win) * L->stride + win) * 2;
        for (int x = 0; x < w; x++) {
            d[2 * x] = (int16_t)(t0[x + 2] - t0[x]);
            d[2 * x + 1] = (int16_t)((t1[x + 2] + t1[x]) * 3 + t1[x + 1] * 10);
        }
    }
    free(t0); free(t1);
}

#define DESCALE(x, n) (((x) + (1 << ((n) - 1))) >> (n))

/* cv::calcOpticalFlowPyrLK(prev, next, prevPts, nextPts, status, err, Size(win, win), maxLevel,
 *                          TermCriteria(COUNT + EPS, maxCount, epsilon), flags, minEigThreshold).
 * flags: 4 = OPTFLOW_USE_INITIAL_FLOW, 8 = OPTFLOW_LK_GET_MIN_EIGENVALS.  next_pts in/out (n x 2). */
void orc_calc_optical_flow_pyr_lk(const uint8_t* prev, const uint8_t* next, int W, int H, int stride, const float* prev_pts,
                                  float* next_pts, int n, int win, int maxLevel, int maxCount, double epsilon, int flags,
                                  float minEigThreshold, uint8_t* status, float* err)
{
    if (maxCount < 0) maxCount = 0; if (maxCount > 100) maxCount = 100;          /* lkpyramid.cpp: criteria clamps */
    if (epsilon < 0) epsilon = 0; if (epsilon > 10) epsilon = 10;
    epsilon *= epsilon;
    for (int i = 0; i < n; i++) { status[i] = 1; err[i] = 0; }
    /* buildOpticalFlowPyramid for both images */
    lk_level P[16], N[16];
    int levels = 0;
    {
        int w = W, h = H;
        for (int lv = 0; lv <= maxLevel && lv < 16; lv++) {
            level_alloc(&P[lv], w, h, win); level_alloc(&N[lv], w, h, win);
            if (lv == 0) {
                for (int y = 0; y < H; y++) { memcpy(&P[0].img[(size_t)(y + win) * P[0].stride + win], prev + (size_t)y * stride, W);
                                              memcpy(&N[0].img[(size_t)(y + win) * N[0].stride + win], next + (size_t)y * stride, W); }
            } else { pyr_down(&P[lv - 1], &P[lv], win); pyr_down(&N[lv - 1], &N[lv], win); }
            pad_reflect(&P[lv], win); pad_reflect(&N[lv], win);
            levels = lv + 1;
            w = (w + 1) / 2; h = (h + 1) / 2;
            if (w <= win || h <= win) break;
        }
        maxLevel = levels - 1;
    }
    for (int lv = 0; lv < levels; lv++) scharr_deriv(&P[lv], win);
    const float halfWin = (win - 1) * 0.5f;
    short* IWin = (short*)malloc(sizeof(short) * win * win); short* dIWin = (short*)malloc(sizeof(short) * win * win * 2);
    for (int level = maxLevel; level >= 0; level--) {
        const lk_level* I = &P[level]; const lk_level* J = &N[level];
        const int st = I->stride;
        for (int pt = 0; pt < n; pt++) {
            float prevx = prev_pts[2 * pt] * (float)(1. / (1 << level)), prevy = prev_pts[2 * pt + 1] * (float)(1. / (1 << level));
            float nextx, nexty;
            if (level == maxLevel) {
                if (flags & 4) { nextx = next_pts[2 * pt] * (float)(1. / (1 << level)); nexty = next_pts[2 * pt + 1] * (float)(1. / (1 << level)); }
                else { nextx = prevx; nexty = prevy; }
            } else { nextx = next_pts[2 * pt] * 2.f; nexty = next_pts[2 * pt + 1] * 2.f; }
            next_pts[2 * pt] = nextx; next_pts[2 * pt + 1] = nexty;
            prevx -= halfWin; prevy -= halfWin;
            const int ipx = (int)floorf(prevx), ipy = (int)floorf(prevy);
            if (ipx < -win || ipx >= I->w || ipy < -win || ipy >= I->h) {
                if (level == 0) { status[pt] = 0; err[pt] = 0; }
                continue;
            }
            float a = prevx - ipx, b = prevy - ipy;
            const int W_BITS = 14, W_BITS1 = 14;
            const float FLT_SCALE = 1.f / (1 << 20);
            int iw00 = orc_cvround((1.f - a) * (1.f - b) * (1 << W_BITS));
            int iw01 = orc_cvround(a * (1.f - b) * (1 << W_BITS));
            int iw10 = orc_cvround((1.f - a) * b * (1 << W_BITS));
            int iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;
            float iA11 = 0, iA12 = 0, iA22 = 0;
            for (int y = 0; y < win; y++) {
                const uint8_t* src = I->img + (size_t)(y + ipy + win) * st + ipx + win;
                const int16_t* dsrc = I->der + ((size_t)(y + ipy + win) * st + ipx + win) * 2;
                for (int x = 0; x < win; x++, dsrc += 2) {
                    const int ival = DESCALE(src[x] * iw00 + src[x + 1] * iw01 + src[x + st] * iw10 + src[x + st + 1] * iw11, W_BITS1 - 5);
                    const int ixval = DESCALE(dsrc[0] * iw00 + dsrc[2] * iw01 + dsrc[2 * st] * iw10 + dsrc[2 * st + 2] * iw11, W_BITS1);
                    const int iyval = DESCALE(dsrc[1] * iw00 + dsrc[3] * iw01 + dsrc[2 * st + 1] * iw10 + dsrc[2 * st + 3] * iw11, W_BITS1);
                    IWin[y * win + x] = (short)ival; dIWin[(y * win + x) * 2] = (short)ixval; dIWin[(y * win + x) * 2 + 1] = (short)iyval;
                    iA11 += (float)(ixval * ixval); iA12 += (float)(ixval * iyval); iA22 += (float)(iyval * iyval);
                }
            }
            const float A11 = iA11 * FLT_SCALE, A12 = iA12 * FLT_SCALE, A22 = iA22 * FLT_SCALE;
            float D = A11 * A22 - A12 * A12;
            const float minEig = (A22 + A11 - sqrtf((A11 - A22) * (A11 - A22) + 4.f * A12 * A12)) / (2 * win * win);
            if (flags & 8) err[pt] = minEig;
            if (minEig < minEigThreshold || D < FLT_EPSILON) {
                if (level == 0) status[pt] = 0;
                continue;
            }
            D = 1.f / D;
            nextx -= halfWin; nexty -= halfWin;
            float pdx = 0, pdy = 0;
            for (int j = 0; j < maxCount; j++) {
                const int inx = (int)floorf(nextx), iny = (int)floorf(nexty);
                if (inx < -win || inx >= J->w || iny < -win || iny >= J->h) {
                    if (level == 0) status[pt] = 0;
                    break;
                }
                a = nextx - inx; b = nexty - iny;
                iw00 = orc_cvround((1.f - a) * (1.f - b) * (1 << W_BITS));
                iw01 = orc_cvround(a * (1.f - b) * (1 << W_BITS));
                iw10 = orc_cvround((1.f - a) * b * (1 << W_BITS));
                iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;
                float ib1 = 0, ib2 = 0;
                for (int y = 0; y < win; y++) {
                    const uint8_t* Jp = J->img + (size_t)(y + iny + win) * J->stride + inx + win;
                    for (int x = 0; x < win; x++) {
                        const int diff = DESCALE(Jp[x] * iw00 + Jp[x + 1] * iw01 + Jp[x + J->stride] * iw10 + Jp[x + J->stride + 1] * iw11,
                                                 W_BITS1 - 5) - IWin[y * win + x];
                        ib1 += (float)(diff * dIWin[(y * win + x) * 2]); ib2 += (float)(diff * dIWin[(y * win + x) * 2 + 1]);
                    }
                }
                const float b1 = ib1 * FLT_SCALE, b2 = ib2 * FLT_SCALE;
                const float dx = (float)((A12 * b2 - A22 * b1) * D), dy = (float)((A12 * b1 - A11 * b2) * D);
                nextx += dx; nexty += dy;
                next_pts[2 * pt] = nextx + halfWin; next_pts[2 * pt + 1] = nexty + halfWin;
                if ((double)dx * dx + (double)dy * dy <= epsilon) break;
                if (j > 0 && fabs(dx + pdx) < 0.01 && fabs(dy + pdy) < 0.01) {
                    next_pts[2 * pt] -= dx * 0.5f; next_pts[2 * pt + 1] -= dy * 0.5f;
                    break;
                }
                pdx = dx; pdy = dy;
            }
            if (status[pt] && level == 0 && !(flags & 8)) {
                const float npx = next_pts[2 * pt] - halfWin, npy = next_pts[2 * pt + 1] - halfWin;
                const int inx = (int)floorf(npx), iny = (int)floorf(npy);
                if (inx < -win || inx >= J->w || iny < -win || iny >= J->h) { status[pt] = 0; continue; }
                const float aa = npx - inx, bb = npy - iny;
                iw00 = orc_cvround((1.f - aa) * (1.f - bb) * (1 << W_BITS));
                iw01 = orc_cvround(aa * (1.f - bb) * (1 << W_BITS));
                iw10 = orc_cvround((1.f - aa) * bb * (1 << W_BITS));
                iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;
                float errval = 0.f;
                for (int y = 0; y < win; y++) {
                    const uint8_t* Jp = J->img + (size_t)(y + iny + win) * J->stride + inx + win;
                    for (int x = 0; x < win; x++) {
                        const int diff = DESCALE(Jp[x] * iw00 + Jp[x + 1] * iw01 + Jp[x + J->stride] * iw10 + Jp[x + J->stride + 1] * iw11,
                                                 W_BITS1 - 5) - IWin[y * win + x];
                        errval += fabsf((float)diff);
                    }
                }
                err[pt] = errval * 1.f / (32 * win * win);
            }
        }
    }
    free(IWin); free(dIWin);
    for (int lv = 0; lv < levels; lv++) { level_free(&P[lv]); level_free(&N[lv]); }
}
