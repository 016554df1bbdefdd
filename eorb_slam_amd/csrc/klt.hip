// klt.hip -- pyramidal Lucas-Kanade tracking on gfx950 for ELK_Tracker (src/Event/KLT_Tracker.cpp:49-98 of the reference):
// cv::calcOpticalFlowPyrLK(refFrame, currImage, refPoints, kpts, status, err, Size(win, win), maxLevel, criteria, flags).
// The algorithm is OpenCV 3.4.1's (modules/video/src/lkpyramid.cpp, scalar code path; modules/imgproc/src/pyramids.cpp), restated:
//   klt_level0 / klt_pyrdown / klt_pad   buildOpticalFlowPyramid: 5x5 [1 4 6 4 1] pyrDown, every level padded by the window size
//                                        with BORDER_REFLECT_101
//   klt_scharr                           calcSharrDeriv of the reference-frame levels (int16 dx, dy; zero padding)
//   klt_track                            LKTrackerInvoker: one wavefront per point walks the levels coarse to fine; 14-bit
//                                        fixed-point bilinear patches computed lane-parallel into LDS, the float normal
//                                        equations summed from there in raster order by single lanes (the order is part of
//                                        the result)
#include "eorb_ctx.h"
#include "dev_math.h"
#include <algorithm>
#include <float.h>
#include <math.h>

namespace eorb {

constexpr int kKltMaxLevels = 8;
struct KltLevel { int w, h, stride; size_t img_off, der_off; };     // stride = w + 2 win; offsets of the padded buffers
struct KltPyr { int levels, win; KltLevel lv[kKltMaxLevels]; };

__device__ __forceinline__ int klt_reflect(int p, int len)
{   // cv::borderInterpolate(p, len, BORDER_REFLECT_101)
    if ((unsigned)p < (unsigned)len) return p;
    if (len == 1) return 0;
    do { if (p < 0) p = -p; else p = 2 * len - p - 2; } while ((unsigned)p >= (unsigned)len);
    return p;
}

// level 0: the image copied into the centre of its padded buffer, padding filled by reflection in the same pass
__global__ void klt_level0_kernel(const uint8_t* __restrict__ src, int sstride, KltLevel L, int win, uint8_t* __restrict__ pyr)
{
    const int X = blockIdx.x * blockDim.x + threadIdx.x, Y = blockIdx.y;
    if (X >= L.stride) return;
    pyr[L.img_off + (size_t)Y * L.stride + X] = src[(size_t)klt_reflect(Y - win, L.h) * sstride + klt_reflect(X - win, L.w)];
}

// cv::pyrDown 8u: horizontal 1 4 6 4 1 in ints, vertical 1 4 6 4 1, (sum + 128) >> 8; source indices by reflect-101
__global__ void klt_pyrdown_kernel(KltLevel S, KltLevel D, int win, uint8_t* __restrict__ pyr)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= D.w) return;
    const uint8_t* s = pyr + S.img_off + (size_t)win * S.stride + win;
    int xs[5];
#pragma unroll
    for (int k = 0; k < 5; k++) xs[k] = klt_reflect(2 * x - 2 + k, S.w);
    int rows[5];
#pragma unroll
    for (int k = 0; k < 5; k++) {
        const uint8_t* r = s + (size_t)klt_reflect(2 * y - 2 + k, S.h) * S.stride;
        rows[k] = r[xs[2]] * 6 + (r[xs[1]] + r[xs[3]]) * 4 + r[xs[0]] + r[xs[4]];
    }
    const int v = rows[0] + rows[4] + (rows[1] + rows[3]) * 4 + rows[2] * 6;
    pyr[D.img_off + (size_t)(y + win) * D.stride + x + win] = (uint8_t)((v + 128) >> 8);
}

// copyMakeBorder(level, padded, win, win, win, win, BORDER_REFLECT_101 | BORDER_ISOLATED) for levels >= 1
__global__ void klt_pad_kernel(KltLevel L, int win, uint8_t* __restrict__ pyr)
{
    const int X = blockIdx.x * blockDim.x + threadIdx.x, Y = blockIdx.y;
    if (X >= L.stride) return;
    const int x = X - win, y = Y - win;
    if (x >= 0 && x < L.w && y >= 0 && y < L.h) return;
    uint8_t* b = pyr + L.img_off;
    b[(size_t)Y * L.stride + X] = b[(size_t)(klt_reflect(y, L.h) + win) * L.stride + klt_reflect(x, L.w) + win];
}

// calcSharrDeriv: vertical [3 10 3] / [-1 0 1], then horizontal; image borders replicate the inner neighbour
__global__ void klt_scharr_kernel(KltLevel L, int win, const uint8_t* __restrict__ pyr, int16_t* __restrict__ der)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= L.w) return;
    const uint8_t* s = pyr + L.img_off + (size_t)win * L.stride + win;
    const int w = L.w, h = L.h;
    const int y0 = y > 0 ? y - 1 : (h > 1 ? 1 : 0), y2 = y < h - 1 ? y + 1 : (h > 1 ? h - 2 : 0);
    auto T0 = [&](int xx) { return (int)(int16_t)((s[(size_t)y0 * L.stride + xx] + s[(size_t)y2 * L.stride + xx]) * 3 + s[(size_t)y * L.stride + xx] * 10); };
    auto T1 = [&](int xx) { return (int)(int16_t)(s[(size_t)y2 * L.stride + xx] - s[(size_t)y0 * L.stride + xx]); };
    const int xm = x > 0 ? x - 1 : (w > 1 ? 1 : 0), xp = x < w - 1 ? x + 1 : (w > 1 ? w - 2 : 0);
    int16_t* d = der + (L.der_off + (size_t)(y + win) * L.stride + x + win) * 2;
    d[0] = (int16_t)(T0(xp) - T0(xm));
    d[1] = (int16_t)((T1(xp) + T1(xm)) * 3 + T1(x) * 10);
}

#define KLT_DESCALE(x, n) (((x) + (1 << ((n) - 1))) >> (n))

struct KltArgs {
    KltPyr P;                          // reference-frame pyramid (img + derivatives) and, at +next_off, the current frame's
    size_t next_off;
    const uint8_t* pyr; const int16_t* der;
    const float* prev_pts; float* next_pts; int n;
    int maxCount; double epsilon; int flags; float minEig;
    uint8_t* status; float* err;
};

constexpr int kKltMaxWin = 31;         // window side handled by the LDS patch buffers (the reference uses 23)

// float sum of v[0..n) in index order, by ONE lane (the order is part of the result); four values per LDS read
__device__ __forceinline__ float klt_ordered_sum(const float* v, int n)
{
    // sixteen values in flight ahead of their adds (one wavefront per point and mostly one per CU: nothing else hides the LDS latency)
    float s = 0.f;
    int k = 0;
    for (; k + 16 <= n; k += 16) {
        const float4 q0 = *(const float4*)(v + k), q1 = *(const float4*)(v + k + 4), q2 = *(const float4*)(v + k + 8), q3 = *(const float4*)(v + k + 12);
        s = s + q0.x; s = s + q0.y; s = s + q0.z; s = s + q0.w;
        s = s + q1.x; s = s + q1.y; s = s + q1.z; s = s + q1.w;
        s = s + q2.x; s = s + q2.y; s = s + q2.z; s = s + q2.w;
        s = s + q3.x; s = s + q3.y; s = s + q3.z; s = s + q3.w;
    }
    for (; k + 4 <= n; k += 4) {
        const float4 q = *(const float4*)(v + k);
        s = s + q.x; s = s + q.y; s = s + q.z; s = s + q.w;
    }
    for (; k < n; k++) s = s + v[k];
    return s;
}

// One wavefront per point.  Lanes share the window's pixels for the fixed-point patch arithmetic (independent per pixel); the
// float accumulations of the normal equations run in raster order on single lanes (lane 0: A11 / b1 / err, lane 1: A12 / b2,
// lane 2: A22) from products staged in LDS.
__global__ __launch_bounds__(64) void klt_track_kernel(KltArgs A)
{
    __shared__ short sI[kKltMaxWin * kKltMaxWin], sDx[kKltMaxWin * kKltMaxWin], sDy[kKltMaxWin * kKltMaxWin];
    __shared__ __attribute__((aligned(16))) float sP0[kKltMaxWin * kKltMaxWin + 3], sP1[kKltMaxWin * kKltMaxWin + 3],
        sP2[kKltMaxWin * kKltMaxWin + 3];
    const int pt = blockIdx.x, lane = threadIdx.x;
    const int win = A.P.win, WW = win * win;
    const float halfWin = (win - 1) * 0.5f;
    const int maxLevel = A.P.levels - 1;
    bool st = true; float errv = 0.f;
    float outx = A.next_pts[2 * pt], outy = A.next_pts[2 * pt + 1];
    const float ppx = A.prev_pts[2 * pt], ppy = A.prev_pts[2 * pt + 1];
    const int W_BITS = 14, W_BITS1 = 14;
    const float FLT_SCALE = 1.f / (1 << 20);
    for (int level = maxLevel; level >= 0; level--) {
        const KltLevel L = A.P.lv[level];
        const int stp = L.stride;
        const uint8_t* I = A.pyr + L.img_off + (size_t)win * stp + win;
        const uint8_t* J = A.pyr + A.next_off + L.img_off + (size_t)win * stp + win;
        const int16_t* dI = A.der + (L.der_off + (size_t)win * stp + win) * 2;
        const float sc = (float)(1. / (1 << level));
        float prevx = ppx * sc, prevy = ppy * sc;
        float nextx, nexty;
        if (level == maxLevel) {
            if (A.flags & 4) { nextx = outx * sc; nexty = outy * sc; }
            else { nextx = prevx; nexty = prevy; }
        } else { nextx = outx * 2.f; nexty = outy * 2.f; }
        outx = nextx; outy = nexty;
        prevx -= halfWin; prevy -= halfWin;
        const int ipx = (int)floorf(prevx), ipy = (int)floorf(prevy);
        if (ipx < -win || ipx >= L.w || ipy < -win || ipy >= L.h) {
            if (level == 0) { st = false; errv = 0.f; }
            continue;
        }
        float a = prevx - (float)ipx, b = prevy - (float)ipy;
        int iw00 = dev_cvround((1.f - a) * (1.f - b) * (1 << W_BITS));
        int iw01 = dev_cvround(a * (1.f - b) * (1 << W_BITS));
        int iw10 = dev_cvround((1.f - a) * b * (1 << W_BITS));
        int iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;
        __syncthreads();                                          // the previous level's readers are done with the buffers
        for (int k = lane; k < WW; k += 64) {
            const int y = k / win, x = k - y * win;
            const uint8_t* src = I + (ptrdiff_t)(y + ipy) * stp + ipx + x;
            const int16_t* ds = dI + ((ptrdiff_t)(y + ipy) * stp + ipx + x) * 2;
            const int ival = KLT_DESCALE(src[0] * iw00 + src[1] * iw01 + src[stp] * iw10 + src[stp + 1] * iw11, W_BITS1 - 5);
            const int ixval = KLT_DESCALE(ds[0] * iw00 + ds[2] * iw01 + ds[2 * stp] * iw10 + ds[2 * stp + 2] * iw11, W_BITS1);
            const int iyval = KLT_DESCALE(ds[1] * iw00 + ds[3] * iw01 + ds[2 * stp + 1] * iw10 + ds[2 * stp + 3] * iw11, W_BITS1);
            sI[k] = (short)ival; sDx[k] = (short)ixval; sDy[k] = (short)iyval;
            sP0[k] = (float)(ixval * ixval); sP1[k] = (float)(ixval * iyval); sP2[k] = (float)(iyval * iyval);
        }
        __syncthreads();
        // the three raster-order chains side by side on lanes 0..2 (one branch per lane would run them one after the other)
        float acc = 0.f;
        if (lane < 3) acc = klt_ordered_sum(lane == 0 ? sP0 : (lane == 1 ? sP1 : sP2), WW);
        const float A11 = __shfl(acc, 0, 64) * FLT_SCALE, A12 = __shfl(acc, 1, 64) * FLT_SCALE, A22 = __shfl(acc, 2, 64) * FLT_SCALE;
        float D = A11 * A22 - A12 * A12;
        const float dif = A11 - A22;
        const float minEig = (A22 + A11 - sqrtf(dif * dif + 4.f * A12 * A12)) / (float)(2 * win * win);
        if (A.flags & 8) errv = minEig;
        if (minEig < A.minEig || D < FLT_EPSILON) {
            if (level == 0) st = false;
            continue;
        }
        D = 1.f / D;
        nextx -= halfWin; nexty -= halfWin;
        float pdx = 0.f, pdy = 0.f;
        // the window's pixels of the next image stay in registers from one iteration to the next: a step moves the window by a
        // fraction of a pixel, so its integer corner -- and with it every pixel the interpolation reads -- is the same most of the time,
        // and only the four weights change (the reload was a global round trip in each of up to ten iterations per level)
        constexpr int T = (kKltMaxWin * kKltMaxWin + 63) / 64;
        uint32_t j00[T], j01[T], j10[T], j11[T];
        int held_x = 0x7fffffff, held_y = 0x7fffffff;
        for (int j = 0; j < A.maxCount; j++) {
            const int inx = (int)floorf(nextx), iny = (int)floorf(nexty);
            if (inx < -win || inx >= L.w || iny < -win || iny >= L.h) {
                if (level == 0) st = false;
                break;
            }
            a = nextx - (float)inx; b = nexty - (float)iny;
            iw00 = dev_cvround((1.f - a) * (1.f - b) * (1 << W_BITS));
            iw01 = dev_cvround(a * (1.f - b) * (1 << W_BITS));
            iw10 = dev_cvround((1.f - a) * b * (1 << W_BITS));
            iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;
            __syncthreads();                                      // last iteration's sums have been read
            {
                // all of the window's pixel reads in flight before the first is used (the window's size is a run-time value: left to
                // itself the loop waits for every trip's four loads, 9 round trips per iteration for a 23 x 23 window)
                if (inx != held_x || iny != held_y) {
                    held_x = inx; held_y = iny;
#pragma unroll
                    for (int t = 0; t < T; t++) {
                        const int k = lane + 64 * t;
                        if (k < WW) {
                            const int y = k / win, x = k - y * win;
                            const uint8_t* Jp = J + (ptrdiff_t)(y + iny) * stp + inx + x;
                            j00[t] = Jp[0]; j01[t] = Jp[1]; j10[t] = Jp[stp]; j11[t] = Jp[stp + 1];
                        }
                    }
                }
#pragma unroll
                for (int t = 0; t < T; t++) {
                    const int k = lane + 64 * t;
                    if (k < WW) {
                        const int diff = KLT_DESCALE((int)j00[t] * iw00 + (int)j01[t] * iw01 + (int)j10[t] * iw10 + (int)j11[t] * iw11, W_BITS1 - 5) - sI[k];
                        sP0[k] = (float)(diff * sDx[k]); sP1[k] = (float)(diff * sDy[k]);
                    }
                }
            }
            __syncthreads();
            float sb = 0.f;
            if (lane < 2) sb = klt_ordered_sum(lane == 0 ? sP0 : sP1, WW);
            const float b1 = __shfl(sb, 0, 64) * FLT_SCALE, b2 = __shfl(sb, 1, 64) * FLT_SCALE;
            const float dx = (A12 * b2 - A22 * b1) * D, dy = (A12 * b1 - A11 * b2) * D;
            nextx += dx; nexty += dy;
            outx = nextx + halfWin; outy = nexty + halfWin;
            if ((double)dx * (double)dx + (double)dy * (double)dy <= A.epsilon) break;
            if (j > 0 && (double)fabsf(dx + pdx) < 0.01 && (double)fabsf(dy + pdy) < 0.01) {
                outx -= dx * 0.5f; outy -= dy * 0.5f;
                break;
            }
            pdx = dx; pdy = dy;
        }
        if (st && level == 0 && !(A.flags & 8)) {
            const float npx = outx - halfWin, npy = outy - halfWin;
            const int inx = (int)floorf(npx), iny = (int)floorf(npy);
            if (inx < -win || inx >= L.w || iny < -win || iny >= L.h) { st = false; continue; }
            const float aa = npx - (float)inx, bb = npy - (float)iny;
            iw00 = dev_cvround((1.f - aa) * (1.f - bb) * (1 << W_BITS));
            iw01 = dev_cvround(aa * (1.f - bb) * (1 << W_BITS));
            iw10 = dev_cvround((1.f - aa) * bb * (1 << W_BITS));
            iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;
            __syncthreads();
            for (int k = lane; k < WW; k += 64) {
                const int y = k / win, x = k - y * win;
                const uint8_t* Jp = J + (ptrdiff_t)(y + iny) * stp + inx + x;
                const int diff = KLT_DESCALE(Jp[0] * iw00 + Jp[1] * iw01 + Jp[stp] * iw10 + Jp[stp + 1] * iw11, W_BITS1 - 5) - sI[k];
                sP0[k] = fabsf((float)diff);
            }
            __syncthreads();
            float e = 0.f;
            if (lane == 0) e = klt_ordered_sum(sP0, WW);
            e = __shfl(e, 0, 64);
            errv = e * 1.f / (float)(32 * win * win);
        }
    }
    if (lane == 0) {
        A.next_pts[2 * pt] = outx; A.next_pts[2 * pt + 1] = outy;
        A.status[pt] = st ? 1 : 0; A.err[pt] = errv;
    }
}

// d_prev / d_next: W x H u8 images (row stride `stride`); d_prev_pts, d_next_pts: n x 2 floats; everything device resident
int klt_track_dev(eorb_ctx* c, const uint8_t* d_prev, const uint8_t* d_next, int W, int H, int stride, const float* d_prev_pts,
                  float* d_next_pts, int n, int win, int maxLevel, int maxCount, double epsilon, int flags, float minEig,
                  uint8_t* d_status, float* d_err, unsigned long long ref_key)
{
    if (win > kKltMaxWin) return set_err(c, EORB_E_CAPACITY, "klt: window %d exceeds the %d-pixel LDS patch", win, kKltMaxWin);
    KltPyr P{}; P.win = win;
    size_t off = 0;
    {
        int w = W, h = H;
        for (int lv = 0; lv <= maxLevel && lv < kKltMaxLevels; lv++) {
            KltLevel& L = P.lv[lv];
            L.w = w; L.h = h; L.stride = w + 2 * win; L.img_off = off; L.der_off = off;
            off += (((size_t)L.stride * (h + 2 * win)) + 63) & ~(size_t)63;
            P.levels = lv + 1;
            w = (w + 1) / 2; h = (h + 1) / 2;
            if (w <= win || h <= win) break;                      // buildOpticalFlowPyramid stops here
        }
    }
    const size_t one = off;                                       // bytes of one padded pyramid
    int rc;
    if ((rc = ensure(c, c->klt_pyr, 2 * one))) return rc;
    if ((rc = ensure(c, c->klt_der, sizeof(int16_t) * 2 * one))) return rc;
    uint8_t* pyr = (uint8_t*)c->klt_pyr.p; int16_t* der = (int16_t*)c->klt_der.p;
    ProfScope ps(c, "klt_track");
    // the reference frame of an LK tracker does not change between its frames (ELK_Tracker::mRefFrame, KLT_Tracker.cpp:22-46): with a
    // key, its pyramid and derivatives are built by the first call and reused by the following ones
    const unsigned long long geo = ((unsigned long long)W << 40) | ((unsigned long long)H << 20) | ((unsigned long long)win << 8) | (unsigned long long)P.levels;
    const bool reuse = ref_key && c->klt_ref_key == ref_key && c->klt_ref_geo == geo;
    c->klt_ref_key = ref_key; c->klt_ref_geo = geo;
    if (!reuse) EORB_HIP(c, hipMemsetAsync(der, 0, sizeof(int16_t) * 2 * one, c->stream));           // derivative padding = BORDER_CONSTANT 0
    for (int f = reuse ? 1 : 0; f < 2; f++) {
        uint8_t* base = pyr + (size_t)f * one;
        const uint8_t* img = f ? d_next : d_prev;
        for (int lv = 0; lv < P.levels; lv++) {
            const KltLevel L = P.lv[lv];
            const dim3 gp((L.stride + 63) / 64, L.h + 2 * win), gi((L.w + 63) / 64, L.h);
            if (lv == 0) klt_level0_kernel<<<gp, 64, 0, c->stream>>>(img, stride, L, win, base);
            else {
                klt_pyrdown_kernel<<<gi, 64, 0, c->stream>>>(P.lv[lv - 1], L, win, base);
                klt_pad_kernel<<<gp, 64, 0, c->stream>>>(L, win, base);
            }
            if (f == 0) klt_scharr_kernel<<<gi, 64, 0, c->stream>>>(L, win, base, der);
        }
    }
    if (n > 0) {
        KltArgs A{P, one, pyr, der, d_prev_pts, d_next_pts, n, std::min(std::max(maxCount, 0), 100), 0.0, flags, minEig, d_status, d_err};
        double eps = std::min(std::max(epsilon, 0.0), 10.0);
        A.epsilon = eps * eps;
        klt_track_kernel<<<n, 64, 0, c->stream>>>(A);
    }
    EORB_LAUNCH_CHECK(c, "klt kernels");
    return EORB_OK;
}

}  // namespace eorb
